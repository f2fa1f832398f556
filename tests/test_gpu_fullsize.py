"""Full-size parity: the BASELINE configurations at their real sizes, HIP frame pipeline vs the CPU oracle, bit for bit.

The oracle (oracle/stm_oracle.c, OpenMP) does a whole 1920x1080, D=64 frame in a few seconds on the GPU box's host cores,
so the headline shape needs no size-independent stand-in: every output of every stage setting is compared with it.
One oracle run serves stages 1, 2 and 3 (it returns the raw WTA maps, the refined maps and the interlaced frame).

  C2  1080p, D=64, cost + aggregation + WTA                      stages=1
  C3  1080p, D=64, + DCC / IRV / bilateral (and with HSLO)       stages=2, 3|0x100
  C4  1080p, D=128, full frame                                   stages=3
  C5  3840x2160, D=256, full frame                               stages=3 (+ a small-frame D=256 case incl. HSLO)
plus the row-tile instantiation of the per-stage aggregation used for 1024 < W <= 2048.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(sbs, p, stages, H, W):
    import torch
    from stm_amd import device_api as dev
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=stages)
    torch.cuda.synchronize()
    return dl.cpu().numpy(), dr.cpu().numpy(), out.cpu().numpy()


def _oracle(orc, sbs, p, H, W, D, zd, hslo=False):
    return orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd,
                            p.thresh_s, p.thresh_h, hslo=hslo)


@pytest.fixture(scope="module")
def frame_1080p_d64(gpu_ready, orc):
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 1080, 1920, 64, 32
    sbs, _ = synth.sbs_frame(H, W, D, zd)  # the frame bench.py times
    p = dev.FrameParams(num_disp=D, zero_disp=zd)  # usd=34, lsd=17, 8 views
    return sbs, p, _oracle(orc, sbs, p, H, W, D, zd)


@pytest.mark.parametrize("stages", [1, 2, 3])
def test_1080p_d64_vs_oracle(frame_1080p_d64, stages):
    """BASELINE configs 2 / 3 / headline at 1920x1080, D=64 (the shape the metric is quoted on)."""
    sbs, p, want = frame_1080p_d64
    dl, dr, out = _run(sbs, p, stages, 1080, 1920)
    if stages == 1:
        assert np.array_equal(dl, want["wta_l"]) and np.array_equal(dr, want["wta_r"])
    else:
        assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    if stages == 3:
        assert np.array_equal(out, want["interlaced"])


def test_1080p_d64_legacy_aggregation_vs_oracle(frame_1080p_d64, stm):
    """The vector-ALU aggregation kernels (stm_set_agg_variant(10000), also the per-stage API's kernels) at the same shape."""
    sbs, p, want = frame_1080p_d64
    lib = stm.lib()
    lib.stm_set_agg_variant(10000)
    try:
        dl, dr, _ = _run(sbs, p, 1, 1080, 1920)
    finally:
        lib.stm_set_agg_variant(0)
    assert np.array_equal(dl, want["wta_l"]) and np.array_equal(dr, want["wta_r"])


def test_1080p_d64_with_hslo_vs_oracle(gpu_ready, orc):
    """BASELINE config 3 with scanline optimisation between aggregation and WTA (parity unpinned: the oracle defines HSLO)."""
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 1080, 1920, 64, 32
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 3)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    want = _oracle(orc, sbs, p, H, W, D, zd, hslo=True)
    dl, dr, out = _run(sbs, p, 3 | 0x100, H, W)
    assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    assert np.array_equal(out, want["interlaced"])


def test_1080p_d128_full_frame_vs_oracle(gpu_ready, orc):
    """BASELINE config 4: D=128 (IRV histograms beyond 65 bins, two chunk sets in the aggregation kernels), 8 views."""
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 1080, 1920, 128, 64
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 4)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    want = _oracle(orc, sbs, p, H, W, D, zd)
    dl, dr, out = _run(sbs, p, 3, H, W)
    assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    assert np.array_equal(out, want["interlaced"])


def test_4k_d256_full_frame_vs_oracle(gpu_ready, orc, stm):
    """BASELINE config 5 on one GPU: 3840x2160, D=256, refinement with 256-bin IRV histograms, 8-view render and mux at
    W=3840, against the oracle (about a minute of host time)."""
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 2160, 3840, 256, 128
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 5)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    dl, dr, out = _run(sbs, p, 3, H, W)
    stm.lib().stm_release_workspace()
    want = _oracle(orc, sbs, p, H, W, D, zd)
    assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    assert np.array_equal(out, want["interlaced"])


@pytest.mark.parametrize("hslo", [False, True])
def test_small_frame_d256_vs_oracle(gpu_ready, orc, hslo):
    """D=256 on a small frame, with and without HSLO (four hypotheses per lane in the scanline kernels)."""
    from stm_amd import device_api as dev, synth
    H, W, D, zd = 96, 256, 256, 128
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 6)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
    want = _oracle(orc, sbs, p, H, W, D, zd, hslo=hslo)
    dl, dr, out = _run(sbs, p, 3 | (0x100 if hslo else 0), H, W)
    assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    assert np.array_equal(out, want["interlaced"])


@pytest.mark.parametrize("H,W,D", [(6, 1500, 5), (5, 2048, 4), (4, 1025, 9)])
def test_ca_cross_rows_between_1024_and_2048(gpu_ready, orc, H, W, D):
    """Per-stage ca_cross with 1024 < W <= 2048: the <512 threads, 4 pixels per thread> row-tile kernel 1080p uses."""
    from conftest import rand_pair
    from stm_amd import host_api as api
    L, _ = rand_pair(8, W, 17 + D)
    L = np.ascontiguousarray(L[:H])
    cost = (np.random.RandomState(W + D).random_sample((D, H, W)) * 2).astype(np.float32)
    x, a = api.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    ox, oa = orc.ca_cross(L, cost, 6.0, 20.0, 34, 17)
    assert np.array_equal(x, ox) and np.array_equal(a, oa)


@pytest.mark.parametrize("H,W,D,zd,usd,lsd", [(70, 131, 20, 9, 34, 17), (49, 67, 17, 8, 40, 20), (33, 258, 64, 32, 5, 2),
                                              (130, 40, 80, 40, 60, 30), (16, 16, 16, 8, 3, 1), (18, 21, 33, 0, 9, 4),
                                              (40, 50, 8, 4, 150, 60), (300, 24, 12, 6, 110, 40), (24, 36, 5, 2, 1, 1),
                                              (1, 1, 1, 0, 3, 1), (2, 3, 2, 1, 5, 2), (3, 5, 1, 0, 34, 17), (1, 70, 4, 2, 9, 4),
                                              (70, 1, 3, 1, 9, 4), (17, 4, 64, 63, 4, 2), (5, 260, 65, 0, 20, 10),
                                              (12, 700, 32, 16, 36, 18), (9, 1003, 7, 3, 20, 10)])
def test_matrix_pipe_aggregation_shapes(gpu_ready, orc, H, W, D, zd, usd, lsd):
    """The frame pipeline's aggregation kernels (stm_kernels_aggm.hip) on ragged shapes: W % 4 != 0 (partial pixel groups),
    D % 16 != 0 (padded chunks), D > 64 (several chunk sets), arms longer than the image, zd at the range edge, usd = 110 (the
    longest arm whose two row rings fit the LDS), usd = 150 (falls back to the vector-ALU kernels), usd = 1; the last two: rows of
    44 and 63 tiles, which the register-ring horizontal kernel (stm_kernels_aggh.hip) splits over four and six waves, usd = 36."""
    from stm_amd import device_api as dev, synth
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=H * 1000 + W)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=usd, lsd=lsd)
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd,
                            p.thresh_s, p.thresh_h, stop_after_wta=True)
    dl, dr, _ = _run(sbs, p, 1, H, W)
    assert np.array_equal(dl, want["wta_l"]) and np.array_equal(dr, want["wta_r"])


def test_allocation_failure_in_error_mode_1_is_clean(gpu_ready, stm):
    """Error mode 1 (record and return): a workspace allocation that cannot succeed must not launch anything on a null or
    stale slab -- the failure is sticky for the call, every launch of it is skipped, and the next call works."""
    import torch
    from stm_amd import device_api as dev, synth
    lib = stm.lib()
    H, W, D, zd = 48, 64, 16, 8
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    good = _run(sbs, p, 3, H, W)
    lib.stm_set_error_mode(1)
    try:
        d_sbs = torch.from_numpy(sbs).cuda()
        dl = torch.full((H, W), 7.0, dtype=torch.float32, device="cuda")
        dr = torch.full((H, W), 7.0, dtype=torch.float32, device="cuda")
        out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
        # absurd geometry: ~10^13 bytes of cost volumes; the tensors above are far too small, but nothing may touch them
        import ctypes as C
        lib.stm_d_adcensus_stm(C.c_void_p(d_sbs.data_ptr()), C.c_void_p(dl.data_ptr()), C.c_void_p(dr.data_ptr()),
                               C.c_void_p(out.data_ptr()), 40000, 80000, 40000, 40000, 40000, 3, 8, 18.43, 256, 128, 10.0, 30.0,
                               6.0, 20.0, 34, 17, 20, 0.4, 3)
        torch.cuda.synchronize()
        assert b"allocation failed" in lib.stm_last_error() or b"memory" in lib.stm_last_error().lower()
        assert float(dl.min()) == 7.0 and float(dr.max()) == 7.0  # untouched
        again = _run(sbs, p, 3, H, W)  # the thread's workspace is usable again
    finally:
        lib.stm_set_error_mode(0)
    for a, b in zip(good, again):
        assert np.array_equal(a, b)


def test_two_host_threads_first_frames_together(gpu_ready, orc, stm):
    """Two host threads whose FIRST call is the full frame pipeline (gaussian-mask norm cache, profiler records, tables): the
    shared caches are filled under locks, results equal the oracle's."""
    import threading
    import torch
    from stm_amd import device_api as dev, synth
    jobs = [(64, 96, 16, 8, 21), (56, 120, 24, 12, 22)]
    frames = [synth.sbs_frame(H, W, D, zd, seed=s)[0] for (H, W, D, zd, s) in jobs]
    results = [None, None]
    dev.prof_reset()
    dev.prof_enable(True)

    def work(i):
        H, W, D, zd, _ = jobs[i]
        p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for _ in range(4):
                results[i] = _run(frames[i], p, 3, H, W)
        stm.lib().stm_release_workspace()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dev.prof_enable(False)
    dev.prof_reset()
    for i, (H, W, D, zd, _) in enumerate(jobs):
        p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=17, lsd=8)
        want = _oracle(orc, frames[i], p, H, W, D, zd)
        dl, dr, out = results[i]
        assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"]) and np.array_equal(out, want["interlaced"])


# ----------------------------------------------------------------------------- BASELINE config 1 at its real size
@pytest.fixture(scope="module")
def c1():
    import os
    from conftest import GOLDEN
    from stm_amd import bmp_io
    g = dict(np.load(os.path.join(GOLDEN, "bud_c1_golden.npz")))
    L, R = bmp_io.read_bmp(os.path.join(GOLDEN, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(GOLDEN, "bud_3.bmp"))
    D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = [float(x) for x in g["params"]]
    return g, L, R, dict(D=int(D), zd=int(zd), ad=ad, ce=ce, ucd=ucd, lcd=lcd, usd=int(usd), lsd=int(lsd), ts=int(ts), th=th, N=int(N), angle=angle)


def test_c1_bud_pair_through_adcensus_stm(gpu_ready, c1):
    """The reference's own img/bud_2 + img/bud_3 pair (640x384, D=32, 8 views) through the blocking host entry point
    stm_adcensus_stm (adcensus_stm, d_io.cu:7-238), against the committed oracle vectors."""
    from stm_amd import host_api as api
    g, L, R, p = c1
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    dl, dr, out = api.adcensus_stm(sbs, 640, 384, 640, p["N"], p["angle"], p["D"], p["zd"], p["ad"], p["ce"], p["ucd"], p["lcd"],
                                   p["usd"], p["lsd"], p["ts"], p["th"])
    assert np.array_equal(dl, g["frame_disp_l"]) and np.array_equal(dr, g["frame_disp_r"])
    assert np.array_equal(out, g["frame_mux"])


def test_c1_bud_pair_through_the_image_io_stage_chain(gpu_ready, c1):
    """The same pair through the per-stage HOST API in the still-image driver's order and constants (image_io.cpp:171-292:
    IRV x1, bilateral 7/7/7, host-flavour dibr_dbm), every stage output against the committed oracle vectors."""
    from stm_amd import host_api as api
    g, L, R, p = c1
    D, zd, N = p["D"], p["zd"], p["N"]
    cl, cr = api.ci_adcensus(L, R, p["ad"], p["ce"], D, zd)
    xl, al = api.ca_cross(L, cl, p["ucd"], p["lcd"], p["usd"], p["lsd"])
    xr, ar = api.ca_cross(R, cr, p["ucd"], p["lcd"], p["usd"], p["lsd"])
    dl, dr = api.dc_wta(al, zd), api.dc_wta(ar, zd)
    assert np.array_equal(xl, g["chain_cross_l"]) and np.array_equal(dl, g["chain_wta_l"].astype(np.float32))
    ol, orr = api.dr_dcc(dl, dr)
    assert np.array_equal(ol, g["chain_outl_l"])
    dl, ol = api.dr_irv(dl, ol, xl, p["ts"], p["th"], D, zd, p["usd"], 1)
    dr, orr = api.dr_irv(dr, orr, xr, p["ts"], p["th"], D, zd, p["usd"], 1)
    dl, dr = api.filter_bilateral_1(dl, 7, 7.0, 7.0, D), api.filter_bilateral_1(dr, 7, 7.0, 7.0, D)
    assert np.array_equal(dl, g["chain_disp_l"]) and np.array_equal(dr, g["chain_disp_r"])
    occl_l, occl_r = api.dibr_occl(dl, dr)
    occl_l, occl_r = api.filter_bleed_1(occl_l, 1), api.filter_bleed_1(occl_r, 1)
    ml, mr = api.dibr_occl_to_mask(occl_l, occl_r)
    views = [R]
    for v in range(1, N - 1):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))
        views.append(api.dibr_dbm(L, R, dl, dr, occl_l, occl_r, ml, mr, shift))
    views.append(L)
    assert np.array_equal(views[3], g["chain_view_3"])
    assert np.array_equal(api.mux_multiview(views, p["angle"], 384, 640), g["chain_mux"])


def test_1080p_real_content_vs_oracle(gpu_ready, orc):
    """The reference's own img/bud_2 + bud_3 pair (committed data fixtures) tiled to 1920x1080, D=64, zd=32 -- the frame
    bench.py's `real_content` leg times: real-image arm lengths and 7x the synthetic frame's outlier density (the IRV list
    and its dirty-tile pruning see a very different load).  Full frame vs the oracle, every element."""
    import os
    from conftest import GOLDEN
    from stm_amd import bmp_io, device_api as dev, synth
    H, W, D, zd = 1080, 1920, 64, 32
    L, R = bmp_io.read_bmp(os.path.join(GOLDEN, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(GOLDEN, "bud_3.bmp"))
    sbs = synth.tiled_sbs_frame(L, R, H, W)
    assert sbs.shape == (H, 2 * W, 3) and np.array_equal(sbs[:384, :640], L) and np.array_equal(sbs[384:768, :640], L[::-1])
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    want = _oracle(orc, sbs, p, H, W, D, zd)
    dl, dr, out = _run(sbs, p, 3, H, W)
    assert np.array_equal(dl, want["disp_l"]) and np.array_equal(dr, want["disp_r"])
    assert np.array_equal(out, want["interlaced"])
    dl1, dr1, _ = _run(sbs, p, 1, H, W)
    assert np.array_equal(dl1, want["wta_l"]) and np.array_equal(dr1, want["wta_r"])
