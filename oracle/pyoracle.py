"""ctypes/numpy front end of the CPU oracle (oracle/stm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Every function takes / returns
numpy arrays in the layouts the reference uses at its host boundary
(SURVEY.md section 8b): BGR u8 [H][W][3], cost volumes float32 [D][H][W],
cross arms u8 [4][H][W] in the order UP, DOWN, LEFT, RIGHT.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "libstm_oracle.so")
    src = os.path.join(_HERE, "stm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


def build_fast():
    """The TIMED build of the same source for bench.py's cpu_baseline leg: -O3 -march=native, still
    -ffp-contract=off and no fast-math, so results are those of the checker build.  Compiled on the machine that runs it
    (-march=native must not travel), into the temp directory."""
    import hashlib
    import tempfile
    src = os.path.join(_HERE, "stm_oracle.c")
    tag = hashlib.sha1(open(src, "rb").read()).hexdigest()[:12]
    so = os.path.join(tempfile.gettempdir(), "libstm_oracle_fast_%s_%d.so" % (tag, os.getuid()))
    if not os.path.exists(so):
        tmp = so + ".%d.tmp" % os.getpid()
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                               "-fvisibility=hidden", "-fopenmp", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


def select(so):
    """Switch this module to another build of the oracle library (bench.py: build_fast())."""
    global _LIB
    _LIB = C.CDLL(so)
    _LIB.orc_mux_y_interval.restype = C.c_float
    return _LIB


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("STM_ORACLE_SO")  # e.g. the sanitizer build of `make -C oracle asan`
        if not so:
            so = os.path.join(_HERE, "libstm_oracle.so")
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
        _LIB.orc_mux_y_interval.restype = C.c_float
    return _LIB


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def set_ref_quirks(on):
    """Non-default: the reference's shared-tile strays at d = 0 (SURVEY A-Q7) in ci_adcensus; widths that are multiples of 160."""
    lib().orc_set_ref_quirks(int(bool(on)))


def set_irv_paper_ratio(on):
    """Non-default: region voting accepts on count / S instead of the reference's bin index / S (SURVEY A-Q17 iv)."""
    lib().orc_set_irv_paper_ratio(int(bool(on)))


def num_threads():
    return int(lib().orc_num_threads())


def usable_cpus():
    """CPUs this process may really use: the smaller of the scheduler affinity and the cgroup CPU quota (a container with
    a 16-CPU share on a 256-thread host runs 256 OpenMP threads far slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def limit_threads_to_usable_cpus():
    """Timed runs (bench.py's cpu_baseline, tools/c1_cpu_timing.py): one OpenMP thread per usable CPU.  Returns the count."""
    n = min(num_threads(), usable_cpus())
    lib().orc_set_num_threads(n)
    return num_threads()


def grey(img):
    H, W, E = img.shape
    img, pi = _u8(img)
    out = np.empty((H, W), np.uint8)
    lib().orc_grey(pi, out.ctypes.data_as(u8p), H, W, E)
    return out


def census(g):
    H, W = g.shape
    g, pg = _u8(g)
    out = np.empty((H, W), np.uint64)
    lib().orc_census(pg, out.ctypes.data_as(u64p), H, W)
    return out


def hamdist64(a, b):
    return int(lib().orc_hamdist64(C.c_uint64(int(a)), C.c_uint64(int(b))))


def hamdist64_closed(a, b):
    return int(lib().orc_hamdist64_closed(C.c_uint64(int(a)), C.c_uint64(int(b))))


def rho_luts(ad_coeff, census_coeff):
    la = np.empty(766, np.float32)
    lc = np.empty(65, np.float32)
    lib().orc_rho_luts(C.c_float(ad_coeff), C.c_float(census_coeff), la.ctypes.data_as(f32p), lc.ctypes.data_as(f32p))
    return la, lc


def ci_adcensus(img_l, img_r, ad_coeff, census_coeff, D, zd):
    H, W, E = img_l.shape
    img_l, pl = _u8(img_l)
    img_r, pr = _u8(img_r)
    cl = np.empty((D, H, W), np.float32)
    cr = np.empty((D, H, W), np.float32)
    lib().orc_ci_adcensus_slab(pl, pr, cl.ctypes.data_as(f32p), cr.ctypes.data_as(f32p),
                               C.c_float(ad_coeff), C.c_float(census_coeff), D, zd, H, W, E)
    return cl, cr


def cross_arms(img, ucd, lcd, usd, lsd):
    H, W, E = img.shape
    img, pi = _u8(img)
    out = np.empty((4, H, W), np.uint8)
    lib().orc_cross_arms(pi, out.ctypes.data_as(u8p), C.c_float(ucd), C.c_float(lcd), usd, lsd, H, W, E)
    return out


def agg_hpass(vol, cross):
    D, H, W = vol.shape
    vol, pv = _f32(vol)
    cross, pc = _u8(cross)
    out = np.empty_like(vol)
    lib().orc_agg_hpass(pv, out.ctypes.data_as(f32p), pc, D, H, W)
    return out


def agg_vpass(vol, cross):
    D, H, W = vol.shape
    vol, pv = _f32(vol)
    cross, pc = _u8(cross)
    out = np.empty_like(vol)
    lib().orc_agg_vpass(pv, out.ctypes.data_as(f32p), pc, D, H, W)
    return out


def ca_cross(img, cost, ucd, lcd, usd, lsd):
    """returns (cross[4][H][W], acost[D][H][W]); cost is left untouched (host flavour)."""
    H, W, E = img.shape
    D = cost.shape[0]
    img, pi = _u8(img)
    cost, pc = _f32(cost)
    cross = np.empty((4, H, W), np.uint8)
    acost = np.empty_like(cost)
    tmp = np.empty_like(cost)
    lib().orc_ca_cross_slab(pi, cross.ctypes.data_as(u8p), pc, acost.ctypes.data_as(f32p), tmp.ctypes.data_as(f32p),
                            C.c_float(ucd), C.c_float(lcd), usd, lsd, D, H, W, E)
    return cross, acost


def dc_wta(cost, zd):
    D, H, W = cost.shape
    cost, pc = _f32(cost)
    disp = np.empty((H, W), np.float32)
    lib().orc_dc_wta_slab(pc, disp.ctypes.data_as(f32p), D, zd, H, W)
    return disp


def dc_hslo(cost, img_l, img_r, T, H1, H2, zd, return_cost=False):
    D, H, W = cost.shape
    E = img_l.shape[2]
    cost, pc = _f32(cost)
    img_l, pl = _u8(img_l)
    img_r, pr = _u8(img_r)
    disp = np.empty((H, W), np.float32)
    co = np.empty_like(cost) if return_cost else None
    lib().orc_dc_hslo_slab(pc, disp.ctypes.data_as(f32p), co.ctypes.data_as(f32p) if return_cost else None,
                           pl, pr, C.c_float(T), C.c_float(H1), C.c_float(H2), D, zd, H, W, E)
    return (disp, co) if return_cost else disp


def dr_dcc(disp_l, disp_r):
    H, W = disp_l.shape
    disp_l, pl = _f32(disp_l)
    disp_r, pr = _f32(disp_r)
    ol = np.empty((H, W), np.uint8)
    orr = np.empty((H, W), np.uint8)
    lib().orc_dr_dcc(ol.ctypes.data_as(u8p), orr.ctypes.data_as(u8p), pl, pr, H, W)
    return ol, orr


def dr_irv(disp, outl, cross, thresh_s, thresh_h, D, zd, usd, iterations, device_flavour=True):
    H, W = disp.shape
    disp = np.array(disp, dtype=np.float32, order="C", copy=True)
    outl = np.array(outl, dtype=np.uint8, order="C", copy=True)
    cross, pc = _u8(cross)
    lib().orc_dr_irv(disp.ctypes.data_as(f32p), outl.ctypes.data_as(u8p), pc, thresh_s, C.c_float(thresh_h),
                     H, W, D, zd, usd, iterations, 1 if device_flavour else 0)
    return disp, outl


def gaussian_kernel_2d(radius, sigma):
    k = np.empty((2 * radius + 1, 2 * radius + 1), np.float32)
    lib().orc_gaussian_kernel_2d(k.ctypes.data_as(f32p), radius, C.c_float(sigma))
    return k


def gaussian_kernel_1d(size, sigma):
    k = np.empty(size, np.float32)
    lib().orc_gaussian_kernel_1d(k.ctypes.data_as(f32p), size, C.c_float(sigma))
    return k


def filter_bilateral_1(img, radius, sigma_color, sigma_spatial, D):
    H, W = img.shape
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    lib().orc_filter_bilateral_1(img.ctypes.data_as(f32p), radius, C.c_float(sigma_color), C.c_float(sigma_spatial), H, W, D)
    return img


def filter_gaussian_1(img, radius, sigma):
    H, W = img.shape
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    lib().orc_filter_gaussian_1(img.ctypes.data_as(f32p), radius, C.c_float(sigma), H, W)
    return img


def filter_median(img):
    H, W = img.shape
    img = np.array(img, dtype=np.float32, order="C", copy=True)
    lib().orc_filter_median(img.ctypes.data_as(f32p), H, W)
    return img


def filter_bleed_1(img, radius):
    H, W = img.shape
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    lib().orc_filter_bleed_1(img.ctypes.data_as(u8p), radius, H, W)
    return img


def dibr_occl(disp_l, disp_r):
    H, W = disp_l.shape
    disp_l, pl = _f32(disp_l)
    disp_r, pr = _f32(disp_r)
    ol = np.empty((H, W), np.uint8)
    orr = np.empty((H, W), np.uint8)
    lib().orc_dibr_occl(ol.ctypes.data_as(u8p), orr.ctypes.data_as(u8p), pl, pr, H, W)
    return ol, orr


def dibr_occl_to_mask(occl_l, occl_r):
    H, W = occl_l.shape
    occl_l, pl = _u8(occl_l)
    occl_r, pr = _u8(occl_r)
    ml = np.empty((H, W), np.float32)
    mr = np.empty((H, W), np.float32)
    lib().orc_dibr_occl_to_mask(ml.ctypes.data_as(f32p), mr.ctypes.data_as(f32p), pl, pr, H, W)
    return ml, mr


def dibr_dbm(img_l, img_r, disp_l, disp_r, mask_l, mask_r, shift, g_radius=10, g_sigma=15.0):
    H, W, E = img_l.shape
    img_l, pil = _u8(img_l)
    img_r, pir = _u8(img_r)
    disp_l, pdl = _f32(disp_l)
    disp_r, pdr = _f32(disp_r)
    mask_l, pml = _f32(mask_l)
    mask_r, pmr = _f32(mask_r)
    out = np.empty((H, W, E), np.uint8)
    lib().orc_dibr_dbm(out.ctypes.data_as(u8p), pil, pir, pdl, pdr, pml, pmr, C.c_float(shift), H, W, E,
                       g_radius, C.c_float(g_sigma))
    return out


def dibr_dfm(img_l, img_r, disp_l, disp_r, shift):
    H, W, E = img_l.shape
    img_l, pil = _u8(img_l)
    img_r, pir = _u8(img_r)
    disp_l, pdl = _f32(disp_l)
    disp_r, pdr = _f32(disp_r)
    out = np.empty((H, W, E), np.uint8)
    lib().orc_dibr_dfm(out.ctypes.data_as(u8p), pil, pir, pdl, pdr, C.c_float(shift), H, W, E)
    return out


def mux_y_interval(num_views, angle, elem_sz=3):
    return float(lib().orc_mux_y_interval(num_views, C.c_float(angle), elem_sz))


def mux_multiview(views, angle, Hout, Wout, variant=2):
    views = [np.ascontiguousarray(v, dtype=np.uint8) for v in views]
    N = len(views)
    Hin, Win, E = views[0].shape
    arr = (u8p * N)(*[v.ctypes.data_as(u8p) for v in views])
    out = np.zeros((Hout, Wout, E), np.uint8)
    lib().orc_mux_multiview(arr, out.ctypes.data_as(u8p), N, C.c_float(angle), Hin, Win, Hout, Wout, E, variant)
    return out


def demux_sbs(sbs, W):
    H, Wsbs, E = sbs.shape
    sbs, ps = _u8(sbs)
    l = np.zeros((H, W, E), np.uint8)
    r = np.zeros((H, W, E), np.uint8)
    lib().orc_demux_sbs(l.ctypes.data_as(u8p), r.ctypes.data_as(u8p), ps, H, Wsbs, W, E)
    return l, r


def adcensus_stm(sbs, Hout, Wout, N, angle, D, zd, ad_coeff, census_coeff, ucd, lcd, usd, lsd, thresh_s, thresh_h,
                 stop_after_wta=False, want_views=False, hslo=False):
    """Whole-frame pipeline (d_io.cu:7-238).  Returns a dict of outputs."""
    H, Wsbs, E = sbs.shape
    W = Wsbs // 2
    sbs, ps = _u8(sbs)
    dl = np.zeros((H, W), np.float32)
    dr = np.zeros((H, W), np.float32)
    wl = np.zeros((H, W), np.float32)
    wr = np.zeros((H, W), np.float32)
    inter = np.zeros((Hout, Wout, E), np.uint8)
    views = np.zeros((N, H, W, E), np.uint8) if want_views else None
    lib().orc_adcensus_stm(ps, dl.ctypes.data_as(f32p), dr.ctypes.data_as(f32p), inter.ctypes.data_as(u8p),
                           H, Wsbs, W, Hout, Wout, E, N, C.c_float(angle), D, zd,
                           C.c_float(ad_coeff), C.c_float(census_coeff), C.c_float(ucd), C.c_float(lcd), usd, lsd,
                           thresh_s, C.c_float(thresh_h), wl.ctypes.data_as(f32p), wr.ctypes.data_as(f32p),
                           views.ctypes.data_as(u8p) if want_views else None, (1 if stop_after_wta else 0) | (2 if hslo else 0))
    return {"disp_l": dl, "disp_r": dr, "wta_l": wl, "wta_r": wr, "interlaced": inter, "views": views}


def tx_scale_bilinear(img, out_rows, out_cols):
    H, W, E = img.shape
    img, pi = _u8(img)
    out = np.zeros((out_rows, out_cols, E), np.uint8)
    lib().orc_tx_scale_bilinear(pi, out.ctypes.data_as(u8p), H, W, out_rows, out_cols, E)
    return out


def tx_disp_scale(disp, out_rows, out_cols, scale):
    h, w = disp.shape
    disp, pd = _f32(disp)
    out = np.zeros((out_rows, out_cols), np.float32)
    lib().orc_tx_disp_scale(out.ctypes.data_as(f32p), pd, out_rows, out_cols, h, w, C.c_float(scale))
    return out


def adcensus_stm_2(sbs, Hout, Wout, h, w, disp_scale, N, angle, D, zd, ad_coeff, census_coeff, ucd, lcd, usd, lsd,
                   thresh_s, thresh_h):
    """Reduced-resolution frame pipeline (d_io.cu:240-508)."""
    H, Wsbs, E = sbs.shape
    W = Wsbs // 2
    sbs, ps = _u8(sbs)
    dl = np.zeros((H, W), np.float32)
    dr = np.zeros((H, W), np.float32)
    inter = np.zeros((Hout, Wout, E), np.uint8)
    lib().orc_adcensus_stm_2(ps, dl.ctypes.data_as(f32p), dr.ctypes.data_as(f32p), inter.ctypes.data_as(u8p),
                             H, Wsbs, W, Hout, Wout, h, w, E, C.c_float(disp_scale), N, C.c_float(angle), D, zd,
                             C.c_float(ad_coeff), C.c_float(census_coeff), C.c_float(ucd), C.c_float(lcd), usd, lsd,
                             thresh_s, C.c_float(thresh_h))
    return {"disp_l": dl, "disp_r": dr, "interlaced": inter}
