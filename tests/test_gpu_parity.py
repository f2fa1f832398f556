"""GPU parity tests: every stage is called through the C ABI (libstm_hip.so, host flavour unless stated) and
compared with the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): bit-exact for integer / byte / index outputs (census, arms, WTA indices,
outlier classes, hit maps, views, interlaced image); float stages within 1e-4 relative to max(|ref|, 1e-3).
The HIP path sums windows and filter taps in the reference's order with no contraction, so the float stages
are in fact asserted BIT-EXACT here, which is stronger than the stated tolerance.
"""
import os

import numpy as np
import pytest

from conftest import rand_pair

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star tolerance for float cost / filter stages


def rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-3)))


def assert_float_stage(got, want):
    assert rel_err(got, want) <= TOL
    assert np.array_equal(got, want), "within tolerance but not bit-exact: max rel err %g" % rel_err(got, want)


CASES = [
    # H, W, D, zd, usd, lsd
    (48, 64, 8, 5, 9, 4),
    (37, 53, 7, 2, 6, 3),       # ragged: odd sizes, D % 4 != 0
    (64, 300, 33, 16, 17, 8),   # W > one tile, D = 4k+1
    (130, 96, 16, 15, 34, 17),  # paper-default arm lengths, zd at the edge of the range
    (20, 24, 1, 0, 3, 1),       # single hypothesis
]


@pytest.fixture(scope="module")
def api(gpu_ready):
    from stm_amd import host_api
    return host_api


@pytest.mark.parametrize("H,W,D,zd,usd,lsd", CASES)
def test_cost_init(api, orc, H, W, D, zd, usd, lsd):
    L, R = rand_pair(H, W, 11 + H)
    cl, cr = api.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    ol, orr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    assert_float_stage(cl, ol)
    assert_float_stage(cr, orr)


@pytest.mark.parametrize("H,W,D,zd,usd,lsd", CASES)
def test_cross_aggregation_and_wta(api, orc, H, W, D, zd, usd, lsd):
    L, R = rand_pair(H, W, 23 + W)
    cost, _ = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    cost_in = cost.copy()
    cross, acost = api.ca_cross(L, cost, 6.0, 20.0, usd, lsd)
    ocross, oacost = orc.ca_cross(L, cost, 6.0, 20.0, usd, lsd)
    assert np.array_equal(cost, cost_in)  # host flavour leaves `cost` untouched (d_ca_cross.cu:419-422)
    assert np.array_equal(cross, ocross)
    assert_float_stage(acost, oacost)
    assert np.array_equal(api.dc_wta(acost, zd), orc.dc_wta(oacost, zd))


def test_aggregation_on_uniform_random_volume(api, orc):
    """The isolated-aggregation workload of SURVEY 8d: uniform [0,2) costs, arms from an image."""
    H, W, D = 72, 200, 12
    L, _ = rand_pair(H, W, 5)
    cost = (np.random.RandomState(9).random_sample((D, H, W)) * 2).astype(np.float32)
    cross, acost = api.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    ocross, oacost = orc.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    assert np.array_equal(cross, ocross) and np.array_equal(acost, oacost)


def _odd_volume(D, H, W, seed):
    """uniform [0,2) costs with isolated elements that are not ordinary numbers (a common "invalid cost" marker), a huge value, a
    denormal and negative values; each kind far enough from the others that every one of them meets windows of its own"""
    rs = np.random.RandomState(seed)
    cost = (rs.random_sample((D, H, W)) * 2).astype(np.float32)
    marks = [np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan), np.finfo(np.float32).max, np.float32(1e-40), np.float32(-3.5),
             -np.finfo(np.float32).max, np.float32(-1e-42)]
    spots = []
    for k, v in enumerate(marks):
        d, y, x = k % D, (7 + 11 * k) % H, (5 + 23 * k) % W
        cost[d, y, x] = v
        spots.append((d, y, x))
    cost[D - 1, H // 2, 1:4] = np.float32(1e-39)  # a run of denormals: a window made of nothing else
    return cost, spots


def _same_with_nans(a, b):
    """element equality where NaNs must sit at the same places (their payload bits are not compared)"""
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb]))


@pytest.mark.parametrize("H,W,D,usd,lsd", [(72, 200, 12, 34, 17), (50, 90, 5, 9, 4), (40, 64, 20, 60, 30)])
def test_ca_cross_nonfinite_and_denormal_planes(api, gpu_ready, orc, H, W, D, usd, lsd):
    """The per-stage ca_cross / d_ca_cross on a caller's volume that holds +-inf, NaN, FLT_MAX, denormals and negative values.
    A non-finite element may only reach the windows that contain it (d_ca_cross_sum.cu:284-289): the matrix-pipe kernels add
    masked elements as acc += 0 * b, so such a volume must take the vector-ALU kernels (stm_k_to_pq raises the flag)."""
    import torch
    from stm_amd import device_api as dev
    L, _ = rand_pair(H, W, 11 + H)
    cost, spots = _odd_volume(D, H, W, 3 + W)
    ocross, oacost = orc.ca_cross(L, cost, 6.0, 20.0, usd, lsd)
    # the oracle itself: a NaN / infinity stays inside the windows that contain it -- most of the volume is finite
    assert np.isfinite(oacost).mean() > 0.5 and not np.isfinite(oacost).all()
    # host flavour
    cross, acost = api.ca_cross(L, cost, 6.0, 20.0, usd, lsd)
    assert np.array_equal(cross, ocross)
    assert _same_with_nans(acost, oacost)
    # device flavour: the result replaces the input planes (A-Q11)
    dL = torch.from_numpy(L).cuda()
    slab = torch.from_numpy(cost.copy()).cuda()
    tab = torch.tensor([slab.data_ptr() + d * H * W * 4 for d in range(D)], dtype=torch.int64).cuda()
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device="cuda")
    dcross = torch.zeros(4, H, W, dtype=torch.uint8, device="cuda")
    dev.d_ca_cross(dL, tab, scratch, dcross, 6.0, 20.0, usd, lsd, D)
    assert np.array_equal(dcross.cpu().numpy(), ocross)
    assert _same_with_nans(slab.cpu().numpy(), oacost)
    # and an ordinary volume right after it is back on the matrix-pipe kernels with the same answer as ever
    plain = (np.random.RandomState(1).random_sample((D, H, W)) * 2).astype(np.float32)
    _, a2 = api.ca_cross(L, plain, 6.0, 20.0, usd, lsd)
    assert np.array_equal(a2, orc.ca_cross(L, plain, 6.0, 20.0, usd, lsd)[1])


def test_wta_ties_and_order(api, orc):
    c = np.ones((9, 6, 10), np.float32)
    c[7, 1, 1] = 0.25
    c[2, 1, 1] = 0.25
    c[8, 5, 9] = -3.0
    assert np.array_equal(api.dc_wta(c, 4), orc.dc_wta(c, 4))


def test_golden_crop_all_stages(api, orc, golden):
    """The committed crop of the reference's img/bud_2 + bud_3 pair, stage by stage against the golden vectors."""
    g = golden
    D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = [float(v) for v in g["params"]]
    D, zd, usd, lsd, ts, N = int(D), int(zd), int(usd), int(lsd), int(ts), int(N)
    L, R = g["L"], g["R"]
    cl, cr = api.ci_adcensus(L, R, ad, ce, D, zd)
    assert np.array_equal(cl, g["cost_l"]) and np.array_equal(cr, g["cost_r"])
    xl, al = api.ca_cross(L, cl, ucd, lcd, usd, lsd)
    xr, ar = api.ca_cross(R, cr, ucd, lcd, usd, lsd)
    assert np.array_equal(xl, g["cross_l"]) and np.array_equal(xr, g["cross_r"])
    assert np.array_equal(al, g["acost_l"]) and np.array_equal(ar, g["acost_r"])
    dl, dr = api.dc_wta(al, zd), api.dc_wta(ar, zd)
    assert np.array_equal(dl, g["wta_l"]) and np.array_equal(dr, g["wta_r"])
    ol, orr = api.dr_dcc(dl, dr)
    assert np.array_equal(ol, g["outl_l"]) and np.array_equal(orr, g["outl_r"])
    # device-flavour IRV semantics (5 x vote+apply) are exercised by the frame test; host flavour votes once
    il, iol = api.dr_irv(dl, ol, xl, ts, th, D, zd, usd, 1)
    wl, wol = orc.dr_irv(dl, ol, xl, ts, th, D, zd, usd, 1, device_flavour=False)
    assert np.array_equal(il, wl) and np.array_equal(iol, wol)
    bl = api.filter_bilateral_1(g["irv_l"], 7, 5.0, 10.0, D)
    assert_float_stage(bl, g["bil_l"])
    ocl, ocr = api.dibr_occl(g["bil_l"], g["bil_r"])
    ocl, ocr = api.filter_bleed_1(ocl, 1), api.filter_bleed_1(ocr, 1)
    assert np.array_equal(ocl, g["occl_l"]) and np.array_equal(ocr, g["occl_r"])
    ml, mr = api.dibr_occl_to_mask(ocl, ocr)
    wml, wmr = orc.dibr_occl_to_mask(g["occl_l"], g["occl_r"])
    assert np.array_equal(ml, wml) and np.array_equal(mr, wmr)
    assert np.array_equal(api.mux_multiview(list(g["views"]), angle, L.shape[0], L.shape[1]), g["mux"])
    assert np.array_equal(api.dc_hslo(g["cost_l"], L, R, 15.0, 1.0, 3.0, zd), g["hslo_l"])
    assert np.array_equal(api.dibr_dfm(L, R, g["bil_l"], g["bil_r"], 0.5), g["dfm"])


def test_golden_crop_whole_frame(api, golden):
    """adcensus_stm (d_io.cu:7-238) end to end: device-flavour constants (IRV x5, bilateral 7/5/10, gaussian 10/15)."""
    g = golden
    D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = [float(v) for v in g["params"]]
    L, R = g["L"], g["R"]
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    dl, dr, mux = api.adcensus_stm(sbs, L.shape[1], L.shape[0], L.shape[1], int(N), angle, int(D), int(zd), ad, ce,
                                   ucd, lcd, int(usd), int(lsd), int(ts), th)
    assert np.array_equal(dl, g["frame_disp_l"]) and np.array_equal(dr, g["frame_disp_r"])
    assert np.array_equal(mux, g["frame_mux"])


def test_fish_identical_pair(api, orc, golden):
    F, D, zd = golden["fish"], int(golden["params"][0]), int(golden["params"][1])
    cl, cr = api.ci_adcensus(F, F, 10.0, 30.0, D, zd)
    assert np.array_equal(cl, golden["fish_cost_l"]) and not cl[zd].any() and not cr[zd].any()
    _, a = api.ca_cross(F, cl, 6.0, 20.0, 9, 4)
    assert not a[zd].any()
    assert (api.dc_wta(a, zd) <= 0).all()


@pytest.mark.parametrize("H,W,D,zd,usd,lsd", CASES[:4])
def test_refinement_chain(api, orc, H, W, D, zd, usd, lsd):
    L, R = rand_pair(H, W, 31 + D)
    cl, cr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    xl, al = orc.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
    xr, ar = orc.ca_cross(R, cr, 6.0, 20.0, usd, lsd)
    dl, dr = orc.dc_wta(al, zd), orc.dc_wta(ar, zd)
    ol, orr = api.dr_dcc(dl, dr)
    wol, worr = orc.dr_dcc(dl, dr)
    assert np.array_equal(ol, wol) and np.array_equal(orr, worr)
    for it in (1, 3):
        il, iol = api.dr_irv(dl, wol, xl, 4, 0.1, D, zd, usd, it)
        wl, wo = orc.dr_irv(dl, wol, xl, 4, 0.1, D, zd, usd, it, device_flavour=False)
        assert np.array_equal(il, wl) and np.array_equal(iol, wo)
    for (r, sc, ss) in [(7, 5.0, 10.0), (7, 7.0, 7.0), (2, 1.5, 1.0)]:
        assert_float_stage(api.filter_bilateral_1(dl, r, sc, ss, D), orc.filter_bilateral_1(dl, r, sc, ss, D))


def test_bilateral_integer_map_kernel_checks_its_promise(api, orc, stm):
    """The frame pipeline's bilateral kernel takes the colour-LUT index from an integer copy of its tile (maps of whole numbers whose
    differences stay inside the LUT: WTA / region-voting output).  It checks that promise tile by tile and falls back to the
    general form, so ANY map gives the reference's result (d_filter_bilateral.cu:284-300): whole numbers inside the LUT (fast
    form), differences >= D (clamped LUT index), fractions, huge values, and a map where only some tiles break the promise.
    stm_set_agg_variant(500) routes the per-stage filter through that kernel."""
    rng = np.random.RandomState(11)
    H, W, D = 70, 150, 16
    maps = {"inside": rng.randint(-7, 8, (H, W)).astype(np.float32),
            "wide": rng.randint(-40, 41, (H, W)).astype(np.float32),
            "fractions": (rng.random_sample((H, W)) * 12 - 6).astype(np.float32),
            "huge": (rng.randint(-3, 4, (H, W)) * 3.0e6).astype(np.float32)}
    mixed = maps["inside"].copy()
    mixed[20:30, 60:90] += 0.25   # a few tiles with fractions
    mixed[50:60, 10:20] *= 9.0    # a few tiles wider than the LUT
    maps["mixed"] = mixed
    stm.lib().stm_set_agg_variant(500)
    try:
        for name, m in maps.items():
            got = api.filter_bilateral_1(m, 7, 5.0, 10.0, D)
            want = orc.filter_bilateral_1(m, 7, 5.0, 10.0, D)
            assert np.array_equal(got, want), name
    finally:
        stm.lib().stm_set_agg_variant(0)


def test_gaussian_and_bleed(api, orc):
    rng = np.random.RandomState(4)
    m = (rng.random_sample((45, 70)) > 0.8).astype(np.float32)
    for (r, s) in [(10, 15.0), (7, 10.0), (1, 0.5)]:
        assert_float_stage(api.filter_gaussian_1(m, r, s), orc.filter_gaussian_1(m, r, s))
    b = (rng.random_sample((33, 41)) > 0.7).astype(np.uint8)
    for r in (1, 2):
        assert np.array_equal(api.filter_bleed_1(b, r), orc.filter_bleed_1(b, r))


@pytest.mark.parametrize("H,W", [(1, 1), (2, 3), (45, 70), (33, 300)])
def test_median(api, orc, H, W):
    """filter_median (d_filter.h:11-12): int-truncating selection sort, flat-index sampling (no border rule)."""
    rng = np.random.RandomState(H * 1000 + W)
    whole = rng.randint(-40, 40, size=(H, W)).astype(np.float32)          # disparity-like: integer valued
    frac = (rng.random_sample((H, W)) * 60 - 30).astype(np.float32)       # bilateral output: fractional
    for img in (whole, frac):
        assert np.array_equal(api.filter_median(img), orc.filter_median(img))


@pytest.mark.parametrize("H,W", [(48, 64), (37, 53), (90, 310)])
def test_dibr_and_mux(api, orc, H, W):
    L, R = rand_pair(H, W, 77)
    rng = np.random.RandomState(H)
    dl = rng.randint(-9, 6, size=(H, W)).astype(np.float32) + rng.random_sample((H, W)).astype(np.float32) * 0.9
    dr = rng.randint(-9, 6, size=(H, W)).astype(np.float32) + rng.random_sample((H, W)).astype(np.float32) * 0.9
    ocl, ocr = api.dibr_occl(dl, dr)
    wl, wr = orc.dibr_occl(dl, dr)
    assert np.array_equal(ocl, wl) and np.array_equal(ocr, wr)
    ml, mr = orc.dibr_occl_to_mask(orc.filter_bleed_1(wl, 1), orc.filter_bleed_1(wr, 1))
    views = [R]
    for v in range(1, 7):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(8) - 1.0)))
        got = api.dibr_dbm(L, R, dl, dr, wl, wr, ml, mr, shift)       # host flavour: gaussian(7, 10)
        want = orc.dibr_dbm(L, R, dl, dr, ml, mr, shift, 7, 10.0)
        assert np.array_equal(got, want)
        views.append(want)
    views.append(L)
    for (Ho, Wo) in [(H, W), (H + 8 - H % 8, 2 * W), (H + 3, W - 5)]:   # kernel_2 and the general kernel (Ho % N != 0)
        variant = 2 if Ho % 8 == 0 else 1
        assert np.array_equal(api.mux_multiview(views, 18.43, Ho, Wo), orc.mux_multiview(views, 18.43, Ho, Wo, variant))
    assert np.array_equal(api.dibr_dfm(L, R, dl, dr, 0.4), orc.dibr_dfm(L, R, dl, dr, 0.4))


def test_hslo(api, orc):
    L, R = rand_pair(40, 56, 8)
    for (D, zd) in [(8, 4), (13, 3)]:
        c, _ = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
        assert np.array_equal(api.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd), orc.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd))


def test_device_flavour_stage_chain(gpu_ready, orc):
    """d_ci_adcensus -> d_ca_cross -> d_dc_wta with the reference's pointer-table contract:
    the cost slab is [2][D][H][W], d_ca_cross overwrites its input volume with the result (A-Q11)."""
    import torch
    from stm_amd import device_api as dev
    H, W, D, zd, usd, lsd = 40, 72, 10, 4, 9, 4
    L, R = rand_pair(H, W, 3)
    dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    slab = torch.zeros(2, D, H, W, dtype=torch.float32, device="cuda")
    tab_l, tab_r = dev.d_ci_adcensus(dL, dR, slab, 10.0, 30.0, D, zd)
    ocl, ocr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    assert np.array_equal(slab[0].cpu().numpy(), ocl) and np.array_equal(slab[1].cpu().numpy(), ocr)
    assert tab_l.cpu().tolist() == [slab.data_ptr() + d * H * W * 4 for d in range(D)]
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device="cuda")
    cross = torch.zeros(4, H, W, dtype=torch.uint8, device="cuda")
    dev.d_ca_cross(dL, tab_l, scratch, cross, 6.0, 20.0, usd, lsd, D)
    ox, oa = orc.ca_cross(L, ocl, 6.0, 20.0, usd, lsd)
    assert np.array_equal(cross.cpu().numpy(), ox)
    assert np.array_equal(slab[0].cpu().numpy(), oa)          # result landed in the input volume
    disp = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dev.d_dc_wta(tab_l, disp, D, zd)
    torch.cuda.synchronize()
    assert np.array_equal(disp.cpu().numpy(), orc.dc_wta(oa, zd))


@pytest.mark.parametrize("stages", [1, 2, 3])
def test_device_frame_pipeline_vs_oracle(gpu_ready, orc, stages):
    """stm_d_adcensus_stm on a synthetic frame bigger than one tile in every direction."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 150, 330, 24, 12
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=stages)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h, stop_after_wta=(stages == 1))
    if stages == 1:
        assert np.array_equal(dl.cpu().numpy(), want["wta_l"]) and np.array_equal(dr.cpu().numpy(), want["wta_r"])
    else:
        if stages == 3:
            assert np.array_equal(out.cpu().numpy(), want["interlaced"])
        assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])


def test_full_size_properties_1080p(gpu_ready):
    """BASELINE config 2 at full size (1920x1080, D=64): too big for the oracle in a unit test, so check
    size-independent properties: (1) the fused pipeline (last pass + WTA in LDS) equals the un-fused
    per-stage device API bit for bit; (2) an identical pair gives offset <= 0 everywhere (cost at
    d = zd aggregates to exactly 0, the global minimum; strict '>' keeps the first zero)."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 1080, 1920, 64, 32
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=1)
    # (1) un-fused chain through the per-stage device API
    dL = d_sbs[:, :W].contiguous()
    dR = d_sbs[:, W:].contiguous()
    slab = torch.zeros(2, D, H, W, dtype=torch.float32, device="cuda")
    tab_l, tab_r = dev.d_ci_adcensus(dL, dR, slab, p.ad_coeff, p.census_coeff, D, zd)
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device="cuda")
    cross = torch.zeros(4, H, W, dtype=torch.uint8, device="cuda")
    dev.d_ca_cross(dL, tab_l, scratch, cross, p.ucd, p.lcd, p.usd, p.lsd, D)
    d2 = torch.zeros_like(dl)
    dev.d_dc_wta(tab_l, d2, D, zd)
    torch.cuda.synchronize()
    assert torch.equal(dl, d2)
    assert float(dl.min()) >= -zd and float(dl.max()) <= D - 1 - zd
    # (2) identical pair
    sbs2 = np.ascontiguousarray(np.concatenate([sbs[:, :W], sbs[:, :W]], axis=1))
    dev.d_adcensus_stm(torch.from_numpy(sbs2).cuda(), dl, dr, out, p, stages=1)
    torch.cuda.synchronize()
    assert float(dl.max()) <= 0 and float(dr.max()) <= 0


def test_full_size_frame_equals_per_stage_device_chain(gpu_ready, stm):
    """1920x1080, D=64, full frame: the frame pipeline (fused split / L-R check / hit-mask / all-views kernels, two
    views per launch) must equal, bit for bit, the chain of per-stage device-flavour calls in d_io.cu's order, which
    runs the separate un-fused kernels.  No oracle involved: a size-independent consistency property."""
    import ctypes as C
    import torch
    from stm_amd import device_api as dev, synth
    lib = stm.lib()
    H, W, D, zd = 1080, 1920, 64, 32
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    N = p.num_views
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    torch.cuda.synchronize()

    P = lambda t: C.c_void_p(t.data_ptr())
    dL = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dR = torch.zeros_like(dL)
    lib.stm_d_demux_sbs(P(dL), P(dR), P(d_sbs), H, 2 * W, W, 3)                                  # d_io.cu:52-60
    slab = torch.zeros(2, D, H, W, dtype=torch.float32, device="cuda")
    tab_l, tab_r = dev.d_ci_adcensus(dL, dR, slab, p.ad_coeff, p.census_coeff, D, zd)             # :74-90
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device="cuda")
    cross_l = torch.zeros(4, H, W, dtype=torch.uint8, device="cuda")
    cross_r = torch.zeros_like(cross_l)
    dev.d_ca_cross(dL, tab_l, scratch, cross_l, p.ucd, p.lcd, p.usd, p.lsd, D)                    # :116-132
    dev.d_ca_cross(dR, tab_r, scratch, cross_r, p.ucd, p.lcd, p.usd, p.lsd, D)
    wl = torch.zeros_like(dl)
    wr = torch.zeros_like(dl)
    dev.d_dc_wta(tab_l, wl, D, zd)
    dev.d_dc_wta(tab_r, wr, D, zd)
    del slab, scratch
    ol = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    orr = torch.zeros_like(ol)
    lib.stm_d_dr_dcc(P(ol), P(orr), P(wl), P(wr), H, W)                                          # :138-143
    xt_l, xt_r = dev.plane_table(cross_l), dev.plane_table(cross_r)
    lib.stm_d_dr_irv(P(wl), P(ol), P(xt_l), p.thresh_s, p.thresh_h, H, W, D, zd, p.usd, 5)       # :147-148
    lib.stm_d_dr_irv(P(wr), P(orr), P(xt_r), p.thresh_s, p.thresh_h, H, W, D, zd, p.usd, 5)
    lib.stm_d_filter_bilateral_1(P(wl), 7, 5.0, 10.0, H, W, D)                                   # :150-151
    lib.stm_d_filter_bilateral_1(P(wr), 7, 5.0, 10.0, H, W, D)
    torch.cuda.synchronize()
    assert torch.equal(dl, wl) and torch.equal(dr, wr)

    occl_l = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    occl_r = torch.zeros_like(occl_l)
    lib.stm_d_dibr_occl(P(occl_l), P(occl_r), P(wl), P(wr), H, W)                                # :165
    lib.stm_d_filter_bleed_1(P(occl_l), 1, H, W)                                                 # :167-168
    lib.stm_d_filter_bleed_1(P(occl_r), 1, H, W)
    ml = torch.zeros_like(dl)
    mr = torch.zeros_like(dl)
    lib.stm_d_dibr_occl_to_mask(P(ml), P(mr), P(occl_l), P(occl_r), H, W)                        # :175-176
    views = [dR] + [torch.zeros_like(dL) for _ in range(N - 2)] + [dL]                           # :182-183
    for v in range(1, N - 1):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))           # :189
        lib.stm_d_dibr_dbm(P(views[v]), P(dL), P(dR), P(wl), P(wr), P(occl_l), P(occl_r), P(ml), P(mr), shift, H, W, 3)
    vt = torch.tensor([v.data_ptr() for v in views], dtype=torch.int64).cuda()
    out2 = torch.zeros_like(out)
    lib.stm_d_mux_multiview(P(vt), P(out2), N, p.angle, H, W, H, W, 3)                            # :203
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


# ----------------------------------------------------------------------------- edge cases
EDGE = [
    # H, W, D, zd, usd, lsd
    (1, 1, 1, 0, 3, 1),        # a single pixel
    (3, 5, 2, 1, 9, 4),        # arms longer than the image
    (2, 70, 5, 4, 34, 17),     # two rows, window wider than a wave
    (9, 17, 3, 7, 5, 2),       # zero_disp outside [0, D): every offset negative
    (70, 33, 6, 2, 40, 10),    # usd > 32: wide IRV rows; H just above one V-pass step
    (150, 40, 4, 1, 70, 20),   # usd = 70: IRV regions taller than 128 rows, ring of 2*70 rows
]


@pytest.mark.parametrize("H,W,D,zd,usd,lsd", EDGE)
def test_edge_cases_stage_chain(api, orc, H, W, D, zd, usd, lsd):
    L, R = rand_pair(max(H, 8), max(W, 8), 101 + H + W)
    L, R = np.ascontiguousarray(L[:H, :W]), np.ascontiguousarray(R[:H, :W])
    cl, cr = api.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    ocl, ocr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    assert np.array_equal(cl, ocl) and np.array_equal(cr, ocr)
    xl, al = api.ca_cross(L, ocl, 6.0, 20.0, usd, lsd)
    oxl, oal = orc.ca_cross(L, ocl, 6.0, 20.0, usd, lsd)
    assert np.array_equal(xl, oxl) and np.array_equal(al, oal)
    _, oar = orc.ca_cross(R, ocr, 6.0, 20.0, usd, lsd)
    dl, dr = orc.dc_wta(oal, zd), orc.dc_wta(oar, zd)
    assert np.array_equal(api.dc_wta(oal, zd), dl)
    ol, orr = api.dr_dcc(dl, dr)
    wol, worr = orc.dr_dcc(dl, dr)
    assert np.array_equal(ol, wol) and np.array_equal(orr, worr)
    il, iol = api.dr_irv(dl, wol, oxl, 1, 0.0, D, zd, usd, 2)
    wl, wo = orc.dr_irv(dl, wol, oxl, 1, 0.0, D, zd, usd, 2, device_flavour=False)
    assert np.array_equal(il, wl) and np.array_equal(iol, wo)
    assert np.array_equal(api.filter_bilateral_1(dl, 7, 5.0, 10.0, max(D, 2)), orc.filter_bilateral_1(dl, 7, 5.0, 10.0, max(D, 2)))


@pytest.mark.parametrize("H,W,D", [(6, 2100, 5), (4, 4200, 4)])
def test_wide_rows_use_the_bigger_row_tiles(api, orc, H, W, D):
    """W > 2048 and W > 4096 switch the H pass to 1024-thread blocks with 4 / 8 pixels per thread."""
    L, _ = rand_pair(8, W, 7)
    L = np.ascontiguousarray(L[:H])
    cost = (np.random.RandomState(W).random_sample((D, H, W)) * 2).astype(np.float32)
    x, a = api.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    ox, oa = orc.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    assert np.array_equal(x, ox) and np.array_equal(a, oa)


def test_baseline_config4_shape_d128(gpu_ready, orc):
    """BASELINE config 4 (D=128, 8-view DIBR + mux) on a reduced frame: IRV histogram > 65 bins (SURVEY A-L7)."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 96, 256, 128, 64
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


def test_real_image_statistics_irv_heavy(gpu_ready, orc, golden):
    """A frame with MANY outliers (random right view): exercises IRV lists, pruning and multi-iteration apply."""
    import torch
    from stm_amd import device_api as dev
    rng = np.random.RandomState(12)
    H, W, D, zd = 80, 192, 16, 8
    L = np.kron(rng.randint(0, 256, size=(H // 4, W // 4, 3)), np.ones((4, 4, 1))).astype(np.uint8)
    R = np.roll(L, -2, axis=1).copy()
    R[:, 40:90] = rng.randint(0, 256, size=(H, 50, 3))  # an unmatched band -> lots of L/R outliers
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8, thresh_s=5, thresh_h=0.1)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=2)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert (want["wta_l"] != want["disp_l"]).any()  # refinement really changed something
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])


def test_4k_d256_properties(gpu_ready):
    """BASELINE config 5 shape on one GPU (3840x2160, D=256): 25 GB of cost volumes through the fused pipeline.
    Size-independent checks: fused pipeline == un-fused per-stage device chain (bit for bit), disparity range,
    and an identical pair yields offsets <= 0 everywhere."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 2160, 3840, 256, 128
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=1)
    torch.cuda.synchronize()
    assert float(dl.min()) >= -zd and float(dl.max()) <= D - 1 - zd
    import stm_amd
    stm_amd.lib().stm_release_workspace()
    dL, dR = d_sbs[:, :W].contiguous(), d_sbs[:, W:].contiguous()
    slab = torch.zeros(2, D, H, W, dtype=torch.float32, device="cuda")
    tab_l, tab_r = dev.d_ci_adcensus(dL, dR, slab, p.ad_coeff, p.census_coeff, D, zd)
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device="cuda")
    cross = torch.zeros(4, H, W, dtype=torch.uint8, device="cuda")
    dev.d_ca_cross(dL, tab_l, scratch, cross, p.ucd, p.lcd, p.usd, p.lsd, D)
    d2 = torch.zeros_like(dl)
    dev.d_dc_wta(tab_l, d2, D, zd)
    torch.cuda.synchronize()
    assert torch.equal(dl, d2)
    del slab, scratch
    sbs2 = np.ascontiguousarray(np.concatenate([sbs[:, :W], sbs[:, :W]], axis=1))
    dev.d_adcensus_stm(torch.from_numpy(sbs2).cuda(), dl, dr, out, p, stages=1)
    torch.cuda.synchronize()
    assert float(dl.max()) <= 0 and float(dr.max()) <= 0
    stm_amd.lib().stm_release_workspace()


def test_error_mode_records_instead_of_exiting(gpu_ready, stm):
    """cuda_utils.h:12-21 semantics (exit) are the default; mode 1 records the message and returns."""
    from stm_amd import host_api
    lib = stm.lib()
    lib.stm_set_error_mode(1)
    try:
        L = np.zeros((2, 8200, 3), np.uint8)
        cost = np.zeros((1, 2, 8200), np.float32)
        lib.stm_set_agg_variant(10000)  # the vector-ALU kernels (the matrix-pipe path of the default build has no such limit)
        host_api.ca_cross(L, cost, 6.0, 20.0, 3, 1)  # num_cols > 8192 is rejected by the row-tile kernel
        lib.stm_set_agg_variant(0)
        assert b"8192" in lib.stm_last_error()
        cross, acost = host_api.ca_cross(L, cost, 6.0, 20.0, 3, 1)  # default path: works, and sums of zeros are zeros
        assert acost.shape == cost.shape and not acost.any() and int(cross[3][0, 0]) == 3
        views = np.zeros((2, 4, 4, 3), np.uint8)
        host_api.mux_multiview([views[0], views[1]], 80.0, 4, 4)  # round(N / tan(angle) / 3) == 0: `ty % 0` in d_mux_multiview.cu:55
        assert b"y_interval" in lib.stm_last_error()
        host_api.mux_multiview([views[0], views[1]], 0.0, 4, 4)     # tan(0): division by zero at :146 (SURVEY A-Q24)
        assert b"not finite" in lib.stm_last_error()
        host_api.mux_multiview([views[0]], 18.43, 4, 4)             # one view: views[1] is read (:62-66)
        assert b"num_views = 1" in lib.stm_last_error()
        host_api.dc_wta(np.zeros((0, 4, 4), np.float32), 0)         # no hypotheses
        assert b"num_disp = 0" in lib.stm_last_error()
    finally:
        lib.stm_set_agg_variant(0)
        lib.stm_set_error_mode(0)


# ----------------------------------------------------------------------------- reduced-resolution mode
@pytest.mark.parametrize("H,W,h,w", [(96, 160, 48, 80), (75, 131, 40, 70)])
def test_reduced_resolution_pipeline(api, orc, H, W, h, w):
    """adcensus_stm_2 (d_io.cu:240-508): bilinear down-scale, match at low resolution, disparity up-scale, render."""
    from stm_amd import synth
    D, zd = 12, 6
    sbs, _ = synth.sbs_frame(H, W, 2 * D, 2 * zd)
    img = np.ascontiguousarray(sbs[:, :W])
    assert np.array_equal(api.tx_scale(img, h, w), orc.tx_scale_bilinear(img, h, w))
    scale = float(w) / float(W)
    got = api.adcensus_stm_2(sbs, W, H, W, h, w, scale, 8, 18.43, D, zd, 10.0, 30.0, 6.0, 20.0, 9, 4, 10, 0.2)
    want = orc.adcensus_stm_2(sbs, H, W, h, w, scale, 8, 18.43, D, zd, 10.0, 30.0, 6.0, 20.0, 9, 4, 10, 0.2)
    assert np.array_equal(got[0], want["disp_l"]) and np.array_equal(got[1], want["disp_r"])
    assert np.array_equal(got[2], want["interlaced"])


# ----------------------------------------------------------------------------- frame sequences
def test_frame_stream_matches_per_frame_calls(gpu_ready, orc):
    """The pipelined sequence front end returns, in order, exactly what adcensus_stm returns frame by frame."""
    from stm_amd import device_api as dev, host_api, synth, video
    H, W, D, zd = 64, 96, 8, 4
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + k)[0] for k in range(5)]
    got = list(video.process_sequence(iter(frames), p))
    assert [g[0] for g in got] == [0, 1, 2, 3, 4]
    for k, f in enumerate(frames):
        dl, dr, out = host_api.adcensus_stm(f, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd,
                                            p.usd, p.lsd, p.thresh_s, p.thresh_h)
        assert np.array_equal(got[k][1], dl) and np.array_equal(got[k][2], dr) and np.array_equal(got[k][3], out)
    want = orc.adcensus_stm(frames[3], H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert np.array_equal(got[3][3], want["interlaced"]) and np.array_equal(got[3][1], want["disp_l"])


def test_frame_stream_zero_copy(gpu_ready, orc):
    """Frames written straight into the stream's pinned input buffer and results read through views of its pinned output
    buffers (no host copies) equal the copying calls, frame for frame; a view stays valid until its slot is reused."""
    from stm_amd import device_api as dev, host_api, synth, video
    H, W, D, zd = 40, 72, 8, 4
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 300 + k)[0] for k in range(6)]
    fs = video.FrameStream(H, W, p)
    got = []
    pending = 0
    for f in frames:
        if pending == 2:
            k, dl, dr, out = fs.collect_view()
            got.append((k, dl.copy(), dr.copy(), out.copy()))
            pending -= 1
        buf = fs.input_buffer()
        assert buf is not None and buf.shape == f.shape
        buf[...] = f
        assert fs.submit_inplace() >= 0
        pending += 1
    assert fs.input_buffer() is None  # both slots in flight
    while pending:
        k, dl, dr, out = fs.collect_view()
        got.append((k, dl.copy(), dr.copy(), out.copy()))
        pending -= 1
    fs.close()
    assert [g[0] for g in got] == list(range(6))
    for k, f in enumerate(frames):
        dl, dr, out = host_api.adcensus_stm(f, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd,
                                            p.usd, p.lsd, p.thresh_s, p.thresh_h)
        assert np.array_equal(got[k][1], dl) and np.array_equal(got[k][2], dr) and np.array_equal(got[k][3], out), k


def test_frame_stream_graph_replay_survives_other_calls(gpu_ready, orc):
    """From its third frame on a stream slot replays a captured hipGraph whose kernel arguments point into the stream's
    private workspace.  Other library calls on the same thread -- here a much larger image, which regrows the shared
    workspace between submissions -- must not disturb it: every frame still equals the per-frame call."""
    from stm_amd import device_api as dev, host_api, synth, video
    H, W, D, zd = 48, 80, 8, 4
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 100 + k)[0] for k in range(9)]
    fs = video.FrameStream(H, W, p)
    got = []
    big_l, big_r = rand_pair(300, 500, 77)
    for k, f in enumerate(frames):
        if k >= 2:
            got.append(fs.collect())
        fs.submit(f)
        if k in (3, 6):
            host_api.ci_adcensus(big_l, big_r, 10.0, 30.0, 24, 12)   # grows the thread's shared workspace
    got.append(fs.collect())
    got.append(fs.collect())
    fs.close()
    assert [g[0] for g in got] == list(range(9))
    for k, f in enumerate(frames):
        dl, dr, out = host_api.adcensus_stm(f, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd,
                                            p.usd, p.lsd, p.thresh_s, p.thresh_h)
        assert np.array_equal(got[k][1], dl) and np.array_equal(got[k][2], dr) and np.array_equal(got[k][3], out), k
    # no kernel had to clamp the region-voting list on the way (a clamp would be reported here, not swallowed)
    import stm_amd
    assert b"outlier list" not in stm_amd.lib().stm_last_error()


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_frame_stream_overlap_modes_in_a_child_process(gpu_ready, overlap):
    """STM_STREAM_OVERLAP=0 (both buffer slots on one compute stream and workspace, as in round 2) and =1 (a stream and workspace
    per slot) give the per-frame call's results.  The switch is read when the stream is created: a child process per mode."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "import numpy as np, stm_amd\n"
        "from stm_amd import device_api as dev, host_api, synth, video\n"
        "H, W, D, zd = 48, 80, 8, 4\n"
        "p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)\n"
        "frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 300 + k)[0] for k in range(6)]\n"
        "fs = video.FrameStream(H, W, p); got = []\n"
        "for k, f in enumerate(frames):\n"
        "    if k >= 2: got.append(fs.collect())\n"
        "    fs.submit(f)\n"
        "got.append(fs.collect()); got.append(fs.collect()); fs.close()\n"
        "for k, f in enumerate(frames):\n"
        "    dl, dr, out = host_api.adcensus_stm(f, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd, p.thresh_s, p.thresh_h)\n"
        "    assert got[k][0] == k and np.array_equal(got[k][1], dl) and np.array_equal(got[k][2], dr) and np.array_equal(got[k][3], out), k\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, STM_STREAM_OVERLAP=overlap)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_video_cli_roundtrip(gpu_ready, orc, tmp_path):
    import subprocess
    import sys
    from conftest import ROOT
    from stm_amd import bmp_io, synth
    H, W, D, zd = 48, 64, 8, 4
    for k in range(3):
        bmp_io.write_bmp(str(tmp_path / ("f%03d.bmp" % k)), synth.sbs_frame(H, W, D, zd, seed=synth.SEED + k)[0])
    out = tmp_path / "o"
    args = [sys.executable, os.path.join(ROOT, "tools", "stm_video.py"), str(tmp_path), "8", "18.43", str(W), str(H), str(D), str(zd),
            "10", "30", "6", "20", "9", "4", "20", "0.4", str(out)]
    subprocess.check_call(args)
    assert sorted(os.listdir(out)) == sorted(["%s_%05d.bmp" % (n, k) for n in ("interlaced", "disp_l", "disp_r") for k in range(3)])
    assert bmp_io.read_bmp(str(out / "interlaced_00001.bmp")).shape == (H, W, 3)
    # like the reference binary, the tool truncates the slant (adcensus_stm takes `int angle`, d_io.h:36): 18.43 -> 18
    f1 = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 1)[0]
    want = orc.adcensus_stm(f1, H, W, 8, 18.0, D, zd, 10.0, 30.0, 6.0, 20.0, 9, 4, 20, 0.4)
    assert np.array_equal(bmp_io.read_bmp(str(out / "interlaced_00001.bmp")), want["interlaced"])


# ----------------------------------------------------------------------------- HSLO (parity unpinned: oracle-defined)
@pytest.mark.parametrize("D,zd", [(64, 32), (100, 40), (200, 90), (5, 1)])
def test_hslo_wave_per_line(api, orc, D, zd):
    """One wave per scan line, lanes = hypotheses: 1, 2 and 4 values per lane (D <= 64, 128, 256) and a ragged D."""
    L, R = rand_pair(36, 70, 19 + D)
    rng = np.random.RandomState(D)
    c = (rng.random_sample((D, 36, 70)) * 3).astype(np.float32)
    assert np.array_equal(api.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd), orc.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd))


@pytest.mark.parametrize("H,W,D,zd", [(1, 1, 1, 0), (1, 9, 4, 0), (2, 3, 7, 6), (7, 2, 16, 8), (5, 5, 17, 0), (9, 13, 65, 64),
                                      (6, 11, 129, 3), (4, 6, 256, 128), (33, 31, 48, 47)])
def test_hslo_ragged_shapes(api, orc, H, W, D, zd):
    """Group (4-pixel) and chunk (16-hypothesis) remainders, single rows / columns, zero_disp at either end of the range:
    the lanes of absent hypotheses hold +inf, the last group is partly outside the image."""
    L, R = rand_pair(H, W, 1000 + 7 * H + W)
    rng = np.random.RandomState(D + W)
    c = (rng.random_sample((D, H, W)) * 3).astype(np.float32)
    assert np.array_equal(api.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd), orc.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd))


def test_hslo_steps_on_the_threshold(api, orc):
    """Colour steps of exactly T fall in neither '<' nor '>' (third penalty class, d_dc_hslo.cu:73-93): grey images whose
    neighbouring pixels differ by 0, 15 or 30 hit all nine (class D1, class D2) combinations."""
    H, W, D, zd = 24, 40, 12, 5
    rng = np.random.RandomState(5)
    def steps():
        g = 60 + 15 * rng.randint(0, 3, size=(H, W))
        return np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
    L, R = steps(), steps()
    c = (rng.random_sample((D, H, W)) * 3).astype(np.float32)
    assert np.array_equal(api.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd), orc.dc_hslo(c, L, R, 15.0, 1.0, 3.0, zd))


@pytest.mark.parametrize("variant", [0, 10000])
def test_device_frame_with_hslo_ragged_width(gpu_ready, orc, variant):
    """A frame whose width is not a multiple of four, on the matrix-pipe aggregation (PQ volumes handed straight to the
    scanline passes) and on the vector-ALU aggregation (quads volumes converted first)."""
    import torch
    import stm_amd
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 37, 203, 20, 7
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=11, lsd=5)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    stm_amd.lib().stm_set_agg_variant(variant)
    try:
        dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3 | 0x100)
        torch.cuda.synchronize()
    finally:
        stm_amd.lib().stm_set_agg_variant(0)
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h, hslo=True)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


@pytest.mark.parametrize("variant", [0, 10, 20, 200, 300, 1000, 2000, 10000, 1000000, 10000000, 100000000, 1000000000])
@pytest.mark.parametrize("shape", [(40, 150, 130, 64, 34, 17), (64, 331, 64, 32, 20, 10), (151, 97, 20, 8, 36, 18)])
def test_device_frame_agg_variants(gpu_ready, orc, variant, shape):
    """Every result-preserving kernel selection of stm_set_agg_variant (include/stm_hip.h) gives the oracle's frame: the row walks
    (0; D = 130 takes three chunk sets through them, with the per-pixel WTA carry in LDS), the block-per-segment horizontal kernels
    (10; 20 = for D > 64 only; 1000 / 2000 = the cost-computing pass), unfused view synthesis (200), region voting over the raster
    list of round 3 instead of over column runs (300), the vector-ALU aggregation
    (10000), the separate cost kernel (1000000), the vertical passes on the LDS-ring kernel of round 3 instead of the register-ring
    kernel (10000000; the third shape has usd = 36, the longest arms the register-ring kernels take, and a height of 151 rows: a
    ragged last tile and three tiles of run-out), the last horizontal pass + WTA on the LDS row walk instead of the register-ring
    kernel (100000000; the default takes the second and third shape: D = 64 and D = 20 < 64, rows of 21 and 7 tiles with a ragged
    last one, split over two waves and one), the window tables of the register-ring kernels from the stand-alone kernels instead of stm_k_cross_arms (1000000000)."""
    import torch
    import stm_amd
    from stm_amd import device_api as dev, synth
    H, W, D, zd, usd, lsd = shape
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=usd, lsd=lsd)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    stm_amd.lib().stm_set_agg_variant(variant)
    try:
        dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3)
        torch.cuda.synchronize()
    finally:
        stm_amd.lib().stm_set_agg_variant(0)
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


def test_device_frame_very_large_disparity_range(gpu_ready, orc):
    """num_disp = 400 (25 chunks, 7 chunk sets; the cost-computing row walk cannot stage 2 x 215 pixels of padding per thread and
    hands the first pass to the block-per-segment kernel; the other passes walk the row once per chunk set)."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 24, 150, 400, 200
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


@pytest.mark.parametrize("variant", [0, 400])
@pytest.mark.parametrize("stages", [1, 3])
def test_device_frame_pipeline_with_hslo(gpu_ready, orc, stages, variant):
    """BASELINE config 3 ordering: aggregation -> scanline optimisation -> WTA -> DCC/IRV/bilateral (stages | 0x100).
    Variant 0: both horizontal directions of a row in one walk from both ends (D <= 64); 400: one launch per direction."""
    import torch
    import stm_amd
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 90, 201, 24, 12
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    stm_amd.lib().stm_set_agg_variant(variant)
    try:
        dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=stages | 0x100)
        torch.cuda.synchronize()
    finally:
        stm_amd.lib().stm_set_agg_variant(0)
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h, stop_after_wta=(stages == 1), hslo=True)
    key = "wta" if stages == 1 else "disp"
    assert np.array_equal(dl.cpu().numpy(), want[key + "_l"]) and np.array_equal(dr.cpu().numpy(), want[key + "_r"])
    if stages == 3:
        assert np.array_equal(out.cpu().numpy(), want["interlaced"])


def test_image_cli_matches_stage_chain(gpu_ready, orc, golden, tmp_path):
    """tools/stm_image.py = image_io.cpp without the window: its interlaced BMP equals the oracle's stage chain with the
    still-image driver's constants (IRV x1 host flavour, bilateral 7/7/7, dibr_dbm gaussian 7/10)."""
    import subprocess
    import sys
    from conftest import ROOT
    from stm_amd import bmp_io
    L, R = golden["L"], golden["R"]
    H, W, _ = L.shape
    bmp_io.write_bmp(str(tmp_path / "l.bmp"), L)
    bmp_io.write_bmp(str(tmp_path / "r.bmp"), R)
    D, zd, usd, lsd, N = 8, 5, 9, 4, 8
    out = tmp_path / "o"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "stm_image.py"), str(tmp_path / "l.bmp"), str(tmp_path / "r.bmp"),
                           "10", "30", str(D), str(zd), "6", "20", str(usd), str(lsd), str(N), "18.43", str(W), str(H), "20", "0.4", str(out)])
    cl, cr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    xl, al = orc.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
    xr, ar = orc.ca_cross(R, cr, 6.0, 20.0, usd, lsd)
    dl, dr = orc.dc_wta(al, zd), orc.dc_wta(ar, zd)
    ol, orr = orc.dr_dcc(dl, dr)
    dl, ol = orc.dr_irv(dl, ol, xl, 20, 0.4, D, zd, usd, 1, device_flavour=False)
    dr, orr = orc.dr_irv(dr, orr, xr, 20, 0.4, D, zd, usd, 1, device_flavour=False)
    dl, dr = orc.filter_bilateral_1(dl, 7, 7.0, 7.0, D), orc.filter_bilateral_1(dr, 7, 7.0, 7.0, D)
    occl_l, occl_r = orc.dibr_occl(dl, dr)
    ml, mr = orc.dibr_occl_to_mask(orc.filter_bleed_1(occl_l, 1), orc.filter_bleed_1(occl_r, 1))
    views = [R]
    for v in range(1, N - 1):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))
        views.append(orc.dibr_dbm(L, R, dl, dr, ml, mr, shift, 7, 10.0))
    views.append(L)
    want = orc.mux_multiview(views, 18.43, H, W, 2)
    assert np.array_equal(bmp_io.read_bmp(str(out / "interlaced.bmp")), want)
    assert np.array_equal(bmp_io.read_bmp(str(out / "view_3.bmp")), views[3])


def test_cxx_host_program_against_the_dropin_header(gpu_ready, orc, golden, tmp_path):
    """examples/stage_chain.cpp uses the reference's own C++ names (stm_dropin.hpp), is built with plain g++ and
    linked to libstm_hip.so: the header-swap integration of INTEGRATION.md, executed.  Same chain as image_io.cpp."""
    import subprocess
    from conftest import ROOT
    from stm_amd import bmp_io
    pkg = os.path.join(ROOT, "stereo-to-multiview-cuda_amd")
    exe = str(tmp_path / "stage_chain")
    subprocess.check_call(["g++", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "stage_chain.cpp"), "-L", pkg, "-lstm_hip",
                           "-Wl,-rpath," + pkg, "-o", exe])
    L, R = golden["L"], golden["R"]
    H, W, _ = L.shape
    bmp_io.write_bmp(str(tmp_path / "l.bmp"), L)
    bmp_io.write_bmp(str(tmp_path / "r.bmp"), R)
    D, zd, usd, lsd, N = 8, 5, 9, 4, 8
    subprocess.check_call([exe, str(tmp_path / "l.bmp"), str(tmp_path / "r.bmp"), str(D), str(zd), str(usd), str(lsd), str(N),
                           str(tmp_path)])
    cl, cr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    xl, al = orc.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
    xr, ar = orc.ca_cross(R, cr, 6.0, 20.0, usd, lsd)
    dl, dr = orc.dc_wta(al, zd), orc.dc_wta(ar, zd)
    assert np.array_equal(np.fromfile(str(tmp_path / "wta_l.f32"), np.float32).reshape(H, W), dl)
    ol, orr = orc.dr_dcc(dl, dr)
    dl, ol = orc.dr_irv(dl, ol, xl, 20, 0.4, D, zd, usd, 1, device_flavour=False)
    dr, orr = orc.dr_irv(dr, orr, xr, 20, 0.4, D, zd, usd, 1, device_flavour=False)
    dl, dr = orc.filter_bilateral_1(dl, 7, 7.0, 7.0, D), orc.filter_bilateral_1(dr, 7, 7.0, 7.0, D)
    assert np.array_equal(np.fromfile(str(tmp_path / "disp_l.f32"), np.float32).reshape(H, W), dl)
    assert np.array_equal(np.fromfile(str(tmp_path / "disp_r.f32"), np.float32).reshape(H, W), dr)
    occl_l, occl_r = orc.dibr_occl(dl, dr)
    ml, mr = orc.dibr_occl_to_mask(orc.filter_bleed_1(occl_l, 1), orc.filter_bleed_1(occl_r, 1))
    views = [R]
    for v in range(1, N - 1):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))
        views.append(orc.dibr_dbm(L, R, dl, dr, ml, mr, shift, 7, 10.0))
    views.append(L)
    assert np.array_equal(bmp_io.read_bmp(str(tmp_path / "interlaced.bmp")), orc.mux_multiview(views, 18.43, H, W, 2))


def test_two_host_threads_share_the_gpu(api, orc, stm):
    """Every host thread has its own workspace and current stream: two threads running different stage chains at the
    same time (ctypes drops the GIL during the calls) get the results of the serial runs."""
    import threading
    jobs = [(40, 56, 9, 4, 9, 4, 11), (33, 71, 6, 2, 12, 5, 12)]
    inputs = [rand_pair(H, W, seed) for (H, W, D, zd, usd, lsd, seed) in jobs]
    results = [None, None]

    def work(i):
        H, W, D, zd, usd, lsd, _ = jobs[i]
        L, R = inputs[i]
        for _ in range(6):
            cl, cr = api.ci_adcensus(L, R, 10.0, 30.0, D, zd)
            x, a = api.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
            d = api.dc_wta(a, zd)
            d = api.filter_bilateral_1(d, 7, 5.0, 10.0, max(D, 2))
        results[i] = (cl, x, a, d)
        stm.lib().stm_release_workspace()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i, (H, W, D, zd, usd, lsd, _) in enumerate(jobs):
        L, R = inputs[i]
        cl, _ = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
        x, a = orc.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
        d = orc.filter_bilateral_1(orc.dc_wta(a, zd), 7, 5.0, 10.0, max(D, 2))
        got = results[i]
        assert got is not None
        assert np.array_equal(got[0], cl) and np.array_equal(got[1], x) and np.array_equal(got[2], a) and np.array_equal(got[3], d)


def test_side_by_side_frame_with_an_odd_column(gpu_ready, orc):
    """num_cols_sbs = 2 * num_cols + 1: the splitter ignores the spare column (d_demux_common.cu:16-31)."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 37, 45, 10, 5
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    sbs = np.ascontiguousarray(np.concatenate([sbs, np.full((H, 1, 3), 200, np.uint8)], axis=1))
    assert sbs.shape[1] == 2 * W + 1
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd,
                            p.thresh_s, p.thresh_h)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


def _fuzz_cases(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        H = int(rng.randint(1, 90))
        W = int(rng.randint(2, 140))
        D = int(rng.randint(1, 41))
        zd = int(rng.randint(0, D)) if rng.rand() < 0.85 else int(rng.randint(-3, D + 4))
        usd = int(rng.randint(1, 41))
        lsd = int(rng.randint(1, usd + 1))
        views = int(rng.randint(2, 10))
        angle = float(rng.choice([18.43, -18.43, 45.0, 7.0, 60.0, 30.0]))
        if round(views / np.tan(np.radians(abs(angle))) / 3.0) < 1:   # row period 0: rejected input, tested elsewhere
            angle = 18.43
        cases.append(dict(H=H, W=W, D=D, zd=zd, usd=usd, lsd=lsd, ucd=float(rng.choice([0.0, 4.0, 6.0, 15.5, 60.0])),
                          lcd=float(rng.choice([0.0, 10.0, 20.0, 33.25, 255.0])), views=views, angle=angle,
                          ad=float(rng.choice([1.0, 10.0, 25.0])), cen=float(rng.choice([5.0, 30.0, 64.0])),
                          ts=int(rng.randint(0, 40)), th=float(rng.choice([0.0, 0.2, 0.4, 0.9, 1.5])),
                          noise=bool(rng.rand() < 0.4), seed=int(rng.randint(1, 1 << 30))))
    return cases


@pytest.mark.parametrize("c", _fuzz_cases(28, 20261004), ids=lambda c: "%dx%d_D%d" % (c["H"], c["W"], c["D"]))
def test_random_frames_and_parameters(gpu_ready, orc, c):
    """Seeded sweep over shapes and every parameter of adcensus_stm (d_io.h:32-40): ragged sizes, 2..9 views,
    zero_disp outside [0, D), zero / huge colour thresholds, pure-noise pairs (every pixel an outlier)."""
    import torch
    from stm_amd import device_api as dev, synth
    H, W, D, zd = c["H"], c["W"], c["D"], c["zd"]
    if c["noise"]:
        L, R = rand_pair(max(H, 8), max(W, 8), c["seed"])
        sbs = np.ascontiguousarray(np.concatenate([L[:H, :W], R[:H, :W]], axis=1))
    else:
        sbs, _ = synth.sbs_frame(H, W, D, min(max(zd, 0), D - 1), seed=c["seed"])
    assert sbs.shape == (H, 2 * W, 3)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, num_views=c["views"], angle=c["angle"], ad_coeff=c["ad"],
                        census_coeff=c["cen"], ucd=c["ucd"], lcd=c["lcd"], usd=c["usd"], lsd=c["lsd"],
                        thresh_s=c["ts"], thresh_h=c["th"])
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd,
                            p.lsd, p.thresh_s, p.thresh_h)
    assert np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
    assert np.array_equal(out.cpu().numpy(), want["interlaced"])


def test_ref_quirks_mode_of_the_per_stage_cost_init(gpu_ready, stm, orc):
    """stm_set_ref_quirks(1): ci_adcensus reproduces the reference's shared-tile strays at d = 0 (SURVEY A-Q7) -- on the
    reference's own pair at the size its launch geometry is valid for (640 = 4 x 160 columns, D = 32, zd = 16: census and AD
    strays) and with D - zd > zd (census strays only).  Non-default; oracle mode orc.set_ref_quirks."""
    import os
    from conftest import GOLDEN
    from stm_amd import bmp_io, host_api
    L, R = bmp_io.read_bmp(os.path.join(GOLDEN, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(GOLDEN, "bud_3.bmp"))
    lib = stm.lib()
    for (rows, D, zd) in ((slice(0, 384), 32, 16), (slice(100, 140), 64, 10)):
        Lc, Rc = np.ascontiguousarray(L[rows]), np.ascontiguousarray(R[rows])
        clean = host_api.ci_adcensus(Lc, Rc, 10.0, 30.0, D, zd)
        lib.stm_set_ref_quirks(1)
        orc.set_ref_quirks(1)
        try:
            got = host_api.ci_adcensus(Lc, Rc, 10.0, 30.0, D, zd)
            want = orc.ci_adcensus(Lc, Rc, 10.0, 30.0, D, zd)
        finally:
            lib.stm_set_ref_quirks(0)
            orc.set_ref_quirks(0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert not np.array_equal(got[0][0], clean[0][0]) and np.array_equal(got[0][1:], clean[0][1:])


def test_paper_ratio_mode_of_region_voting(gpu_ready, stm, orc):
    """stm_set_irv_paper_ratio(1): dr_irv accepts on count / S instead of the reference's bin index / S (SURVEY A-Q17 iv);
    both rules against the oracle's, on a real crop where they differ."""
    import os
    from conftest import GOLDEN
    from stm_amd import bmp_io, host_api
    L = np.ascontiguousarray(bmp_io.read_bmp(os.path.join(GOLDEN, "bud_2.bmp"))[120:248, 200:456])
    R = np.ascontiguousarray(bmp_io.read_bmp(os.path.join(GOLDEN, "bud_3.bmp"))[120:248, 200:456])
    D, zd, usd, lsd = 32, 16, 17, 8
    cl, cr = host_api.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    cross_l, al = host_api.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
    cross_r, ar = host_api.ca_cross(R, cr, 6.0, 20.0, usd, lsd)
    dl, dr = host_api.dc_wta(al, zd), host_api.dc_wta(ar, zd)
    ol, _ = host_api.dr_dcc(dl, dr)
    lib = stm.lib()
    res = {}
    for mode in (0, 1):
        lib.stm_set_irv_paper_ratio(mode)
        orc.set_irv_paper_ratio(mode)
        try:
            g = host_api.dr_irv(dl, ol, cross_l, 20, 0.4, D, zd, usd, 5)
            w = orc.dr_irv(dl, ol, cross_l, 20, 0.4, D, zd, usd, 5, device_flavour=False)
        finally:
            lib.stm_set_irv_paper_ratio(0)
            orc.set_irv_paper_ratio(0)
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
        res[mode] = g
    assert not np.array_equal(res[0][1], res[1][1])  # the two rules accept different pixels here
