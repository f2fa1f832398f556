"""Device-flavour stage API: torch CUDA(=HIP) tensors in, work enqueued on torch's current stream,
nothing synchronised -- the analogue of the reference's `d_stage(device ptrs...)` wrappers that
adcensus_stm (d_io.cu:7-238) chains.  torch is plumbing only (device memory + streams); every kernel is
hand-written HIP inside libstm_hip.so.
"""
import ctypes as C

import torch

from ._lib import f32p, f32pp, lib


def _use_current_stream():
    lib().stm_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))


def _p(t):
    return C.c_void_p(t.data_ptr())


class FrameParams:
    """Parameters of one adcensus_stm call (d_io.h:32-40), defaults from SURVEY.md section 8d."""

    def __init__(self, num_disp=64, zero_disp=32, num_views=8, angle=18.43, ad_coeff=10.0, census_coeff=30.0,
                 ucd=6.0, lcd=20.0, usd=34, lsd=17, thresh_s=20, thresh_h=0.4, out_rows=None, out_cols=None):
        self.num_disp, self.zero_disp, self.num_views, self.angle = num_disp, zero_disp, num_views, angle
        self.ad_coeff, self.census_coeff = ad_coeff, census_coeff
        self.ucd, self.lcd, self.usd, self.lsd = ucd, lcd, usd, lsd
        self.thresh_s, self.thresh_h = thresh_s, thresh_h
        self.out_rows, self.out_cols = out_rows, out_cols


def d_adcensus_stm(sbs, disp_l, disp_r, interlaced, p, stages=3):
    """stm_d_adcensus_stm: sbs uint8 [H][2W][3] on the GPU; outputs are written in place.
    stages: 1 = cost+aggregation+WTA, 2 = + refinement, 3 = full frame (views + interlacing)."""
    assert sbs.is_cuda and sbs.dtype == torch.uint8 and sbs.is_contiguous()
    H, Wsbs, E = sbs.shape
    W = Wsbs // 2
    assert disp_l.shape == (H, W) and disp_r.shape == (H, W) and disp_l.dtype == torch.float32
    Ho, Wo = interlaced.shape[0], interlaced.shape[1]
    _use_current_stream()
    lib().stm_d_adcensus_stm(_p(sbs), _p(disp_l), _p(disp_r), _p(interlaced), H, Wsbs, W, Ho, Wo, E,
                             p.num_views, p.angle, p.num_disp, p.zero_disp, p.ad_coeff, p.census_coeff,
                             p.ucd, p.lcd, p.usd, p.lsd, p.thresh_s, p.thresh_h, stages)


def plane_table(slab):
    """Device table of plane pointers for a contiguous [D][H][W] float tensor."""
    D = slab.shape[0]
    stride = slab.stride(0) * slab.element_size()
    ptrs = torch.tensor([slab.data_ptr() + d * stride for d in range(D)], dtype=torch.int64)
    return ptrs.to(slab.device)


def d_ci_adcensus(img_l, img_r, slab, ad_coeff, census_coeff, num_disp, zero_disp):
    """stm_d_ci_adcensus: slab float32 [2][D][H][W]; returns (d_tab_l, d_tab_r) device pointer tables."""
    H, W, E = img_l.shape
    _use_current_stream()
    tab_l = torch.zeros(num_disp, dtype=torch.int64, device=slab.device)
    tab_r = torch.zeros(num_disp, dtype=torch.int64, device=slab.device)
    h_l = (f32p * num_disp)()
    h_r = (f32p * num_disp)()
    lib().stm_d_ci_adcensus(_p(img_l), _p(img_r), _p(tab_l), _p(tab_r), C.cast(h_l, f32pp), C.cast(h_r, f32pp),
                            _p(slab), ad_coeff, census_coeff, num_disp, zero_disp, H, W, E)
    torch.cuda.current_stream().synchronize()  # h_l/h_r are read by the async table upload
    return tab_l, tab_r


def d_ca_cross(img, cost_tab, scratch, cross, ucd, lcd, usd, lsd, num_disp):
    """stm_d_ca_cross: result lands in the planes cost_tab points at; cross uint8 [4][H][W] receives the arms."""
    H, W, E = img.shape
    _use_current_stream()
    acost_tab = torch.zeros(num_disp, dtype=torch.int64, device=img.device)
    h_a = (f32p * num_disp)()
    cross_tab = torch.tensor([cross[k].data_ptr() for k in range(4)], dtype=torch.int64).to(img.device)
    lib().stm_d_ca_cross(_p(img), _p(cost_tab), _p(acost_tab), C.cast(h_a, f32pp), _p(scratch), _p(cross_tab),
                         ucd, lcd, usd, lsd, num_disp, H, W, E)
    torch.cuda.current_stream().synchronize()
    return cross_tab


def d_dc_wta(cost_tab, disp, num_disp, zero_disp):
    H, W = disp.shape
    _use_current_stream()
    lib().stm_d_dc_wta(_p(cost_tab), _p(disp), num_disp, zero_disp, H, W)


def prof_enable(on=True):
    """True / 1: HIP events around every named kernel; 2: around the aggregation kernels only; False / 0: off."""
    lib().stm_prof_enable(int(on))


def prof_reset():
    lib().stm_prof_reset()


def prof_read(name):
    ms = C.c_float(0.0)
    n = lib().stm_prof_read(name.encode(), C.byref(ms))
    return n, float(ms.value)
