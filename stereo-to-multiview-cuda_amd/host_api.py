"""Host-flavour stage API: numpy in, numpy out, every call crosses host<->device like the reference's
`stage(host ptrs...)` wrappers that image_io.cpp:171-292 calls.  Names, argument order and semantics follow
the reference headers (d_*.h); the arrays replace the raw pointers:

  images       uint8  [H][W][3]  BGR
  cost volume  float32 [D][H][W] (the C ABI receives it as the reference's table of D plane pointers)
  cross arms   uint8  [4][H][W]  UP, DOWN, LEFT, RIGHT
  disparity    float32 [H][W]
All work happens in libstm_hip.so (HIP kernels); nothing here computes.
"""
import ctypes as C

import numpy as np

from ._lib import f32p, f32pp, lib, u8p, u8pp


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _plane_table_f32(vol):
    """float** over the planes of a contiguous [D][H][W] array."""
    D = vol.shape[0]
    return (f32p * D)(*[vol[d].ctypes.data_as(f32p) for d in range(D)])


def _plane_table_u8(vol):
    n = vol.shape[0]
    return (u8p * n)(*[vol[k].ctypes.data_as(u8p) for k in range(n)])


def ci_adcensus(img_l, img_r, ad_coeff, census_coeff, num_disp, zero_disp):
    """d_ci_adcensus.h:23-25.  Returns (cost_l, cost_r), each [D][H][W]."""
    H, W, E = img_l.shape
    img_l, pl = _u8(img_l)
    img_r, pr = _u8(img_r)
    cl = np.zeros((num_disp, H, W), np.float32)
    cr = np.zeros((num_disp, H, W), np.float32)
    lib().stm_ci_adcensus(pl, pr, C.cast(_plane_table_f32(cl), f32pp), C.cast(_plane_table_f32(cr), f32pp),
                          ad_coeff, census_coeff, num_disp, zero_disp, H, W, E)
    return cl, cr


def ca_cross(img, cost, ucd, lcd, usd, lsd):
    """d_ca_cross.h:19-21.  Returns (cross[4][H][W], acost[D][H][W]); `cost` is left untouched."""
    H, W, E = img.shape
    img, pi = _u8(img)
    cost, _ = _f32(cost)
    D = cost.shape[0]
    cross = np.zeros((4, H, W), np.uint8)
    acost = np.zeros_like(cost)
    lib().stm_ca_cross(pi, C.cast(_plane_table_u8(cross), u8pp), C.cast(_plane_table_f32(cost), f32pp),
                       C.cast(_plane_table_f32(acost), f32pp), ucd, lcd, usd, lsd, D, H, W, E)
    return cross, acost


def dc_wta(cost, zero_disp):
    """d_dc_wta.h:16-18."""
    cost, _ = _f32(cost)
    D, H, W = cost.shape
    disp = np.zeros((H, W), np.float32)
    lib().stm_dc_wta(C.cast(_plane_table_f32(cost), f32pp), disp.ctypes.data_as(f32p), D, zero_disp, H, W)
    return disp


def dc_hslo(cost, img_l, img_r, T, H1, H2, zero_disp):
    """d_dc_hslo.h:18-22 (parity unpinned: the reference is a stub)."""
    cost, _ = _f32(cost)
    D, H, W = cost.shape
    img_l, pl = _u8(img_l)
    img_r, pr = _u8(img_r)
    disp = np.zeros((H, W), np.float32)
    lib().stm_dc_hslo(C.cast(_plane_table_f32(cost), f32pp), disp.ctypes.data_as(f32p), pl, pr, T, H1, H2,
                      D, zero_disp, H, W, img_l.shape[2])
    return disp


def dr_dcc(disp_l, disp_r):
    """d_dr_dcc.h:17-19.  Returns (outliers_l, outliers_r) in {0,1,2}."""
    disp_l, pl = _f32(disp_l)
    disp_r, pr = _f32(disp_r)
    H, W = disp_l.shape
    ol = np.zeros((H, W), np.uint8)
    orr = np.zeros((H, W), np.uint8)
    lib().stm_dr_dcc(ol.ctypes.data_as(u8p), orr.ctypes.data_as(u8p), pl, pr, H, W)
    return ol, orr


def dr_irv(disp, outliers, cross, thresh_s, thresh_h, num_disp, zero_disp, usd, iterations):
    """d_dr_irv.h:15-19 (in place in the reference; copies are returned here)."""
    disp = np.array(disp, dtype=np.float32, order="C", copy=True)
    outliers = np.array(outliers, dtype=np.uint8, order="C", copy=True)
    cross, _ = _u8(cross)
    H, W = disp.shape
    lib().stm_dr_irv(disp.ctypes.data_as(f32p), outliers.ctypes.data_as(u8p), C.cast(_plane_table_u8(cross), u8pp),
                     thresh_s, thresh_h, H, W, num_disp, zero_disp, usd, iterations)
    return disp, outliers


def filter_bilateral_1(img, radius, sigma_color, sigma_spatial, num_disp):
    """d_filter_bilateral.h:17-20."""
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    H, W = img.shape
    lib().stm_filter_bilateral_1(img.ctypes.data_as(f32p), radius, sigma_color, sigma_spatial, H, W, num_disp)
    return img


def filter_gaussian_1(img, radius, sigma_spatial):
    """d_filter_gaussian.h:20-22 (grow-only: out = max(in, blur))."""
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    H, W = img.shape
    lib().stm_filter_gaussian_1(img.ctypes.data_as(f32p), radius, sigma_spatial, H, W)
    return img


def filter_median(img):
    """d_filter.h:11-12."""
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    H, W = img.shape
    lib().stm_filter_median(img.ctypes.data_as(f32p), H, W)
    return img


def filter_bleed_1(img, radius):
    """d_filter.h:26-28."""
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    H, W = img.shape
    lib().stm_filter_bleed_1(img.ctypes.data_as(u8p), radius, H, W)
    return img


def dibr_occl(disp_l, disp_r):
    """d_dibr_occl.h:31-33.  Returns (occl_l, occl_r) hit maps."""
    disp_l, pl = _f32(disp_l)
    disp_r, pr = _f32(disp_r)
    H, W = disp_l.shape
    ol = np.zeros((H, W), np.uint8)
    orr = np.zeros((H, W), np.uint8)
    lib().stm_dibr_occl(ol.ctypes.data_as(u8p), orr.ctypes.data_as(u8p), pl, pr, H, W)
    return ol, orr


def dibr_occl_to_mask(occl_l, occl_r):
    """d_dibr_occl.h:18-20."""
    occl_l, pl = _u8(occl_l)
    occl_r, pr = _u8(occl_r)
    H, W = occl_l.shape
    ml = np.zeros((H, W), np.float32)
    mr = np.zeros((H, W), np.float32)
    lib().stm_dibr_occl_to_mask(ml.ctypes.data_as(f32p), mr.ctypes.data_as(f32p), pl, pr, H, W)
    return ml, mr


def dibr_dbm(img_l, img_r, disp_l, disp_r, occl_l, occl_r, mask_l, mask_r, shift):
    """d_dibr_bwarp.h:29-34 (host flavour: mask blur gaussian(7,10))."""
    H, W, E = img_l.shape
    img_l, pil = _u8(img_l)
    img_r, pir = _u8(img_r)
    disp_l, pdl = _f32(disp_l)
    disp_r, pdr = _f32(disp_r)
    occl_l, pol = _u8(occl_l)
    occl_r, por = _u8(occl_r)
    mask_l, pml = _f32(mask_l)
    mask_r, pmr = _f32(mask_r)
    out = np.zeros((H, W, E), np.uint8)
    lib().stm_dibr_dbm(out.ctypes.data_as(u8p), pil, pir, pdl, pdr, pol, por, pml, pmr, shift, H, W, E)
    return out


def dibr_dfm(img_l, img_r, disp_l, disp_r, shift):
    """d_dibr_fwarp.h:17-20 (deterministic; parity unpinned)."""
    H, W, E = img_l.shape
    img_l, pil = _u8(img_l)
    img_r, pir = _u8(img_r)
    disp_l, pdl = _f32(disp_l)
    disp_r, pdr = _f32(disp_r)
    out = np.zeros((H, W, E), np.uint8)
    lib().stm_dibr_dfm(out.ctypes.data_as(u8p), pil, pir, pdl, pdr, shift, H, W, E)
    return out


def mux_multiview(views, angle, out_rows, out_cols):
    """d_mux_multiview.h:39-41.  views[0] = right image ... views[N-1] = left image."""
    views = [np.ascontiguousarray(v, dtype=np.uint8) for v in views]
    N = len(views)
    H, W, E = views[0].shape
    tab = (u8p * N)(*[v.ctypes.data_as(u8p) for v in views])
    out = np.zeros((out_rows, out_cols, E), np.uint8)
    lib().stm_mux_multiview(C.cast(tab, u8pp), out.ctypes.data_as(u8p), N, angle, H, W, out_rows, out_cols, E)
    return out


def adcensus_stm(img_sbs, num_cols, out_rows, out_cols, num_views, angle, num_disp, zero_disp,
                 ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h):
    """d_io.h:32-40: host SBS frame in, (disp_l, disp_r, interlaced) out."""
    img_sbs, ps = _u8(img_sbs)
    H, Wsbs, E = img_sbs.shape
    dl = np.zeros((H, num_cols), np.float32)
    dr = np.zeros((H, num_cols), np.float32)
    out = np.zeros((out_rows, out_cols, E), np.uint8)
    lib().stm_adcensus_stm(ps, dl.ctypes.data_as(f32p), dr.ctypes.data_as(f32p), out.ctypes.data_as(u8p),
                           H, Wsbs, num_cols, out_rows, out_cols, E, num_views, angle, num_disp, zero_disp,
                           ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h)
    return dl, dr, out


def adcensus_stm_2(img_sbs, num_cols, out_rows, out_cols, disp_rows, disp_cols, disp_scale, num_views, angle, num_disp,
                   zero_disp, ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h):
    """d_io.h:42-52: disparity at reduced resolution (disp_rows x disp_cols), views at full resolution."""
    img_sbs, ps = _u8(img_sbs)
    H, Wsbs, E = img_sbs.shape
    dl = np.zeros((H, num_cols), np.float32)
    dr = np.zeros((H, num_cols), np.float32)
    out = np.zeros((out_rows, out_cols, E), np.uint8)
    lib().stm_adcensus_stm_2(ps, dl.ctypes.data_as(f32p), dr.ctypes.data_as(f32p), out.ctypes.data_as(u8p),
                             H, Wsbs, num_cols, out_rows, out_cols, disp_rows, disp_cols, E, disp_scale, num_views, angle,
                             num_disp, zero_disp, ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h)
    return dl, dr, out


def tx_scale(img, out_rows, out_cols):
    """d_tx_scale.h:17-18 (bilinear resize)."""
    img, pi = _u8(img)
    H, W, E = img.shape
    out = np.zeros((out_rows, out_cols, E), np.uint8)
    lib().stm_d_tx_scale(pi, out.ctypes.data_as(u8p), H, W, out_rows, out_cols, E)
    return out


def bmp_read(path):
    """stm_bmp_read: the C++ twin of bmp_io.read_bmp (replaces cv::imread, image_io.cpp:95-96)."""
    h, w = C.c_int(0), C.c_int(0)
    p = lib().stm_bmp_read(path.encode(), C.byref(h), C.byref(w))
    if not p:
        raise IOError("stm_bmp_read failed for %s" % path)
    try:
        arr = np.ctypeslib.as_array(C.cast(p, u8p), shape=(h.value, w.value, 3)).copy()
    finally:
        lib().stm_bmp_free(p)
    return arr


def bmp_write(path, img):
    img, pi = _u8(img)
    if lib().stm_bmp_write(path.encode(), pi, img.shape[0], img.shape[1]) != 0:
        raise IOError("stm_bmp_write failed for %s" % path)
