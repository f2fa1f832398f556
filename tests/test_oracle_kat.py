"""CPU tests of the oracle: hand-computable known answers, the degenerate identical
pair, and the committed golden vectors.  No GPU, no HIP library."""
import os

import numpy as np
import pytest

from conftest import REF_IMG, rand_pair


def test_hamdist_kats(orc):
    # alu_hamdist_64 keeps the low 32 bits and weighs bit 31 by 33 (d_alu.cu:7-15, SURVEY A-Q1)
    assert orc.hamdist64(0, 0x80000000) == 33
    assert orc.hamdist64(0, 1 << 32) == 0
    assert orc.hamdist64(0, 0x7FFFFFFF) == 31
    assert orc.hamdist64(0, 0xFFFFFFFF) == 64
    assert orc.hamdist64(0xFFFF00000000, 0) == 0
    rng = np.random.RandomState(1)
    for _ in range(200):
        a, b = int(rng.randint(0, 2 ** 48, dtype=np.int64)), int(rng.randint(0, 2 ** 48, dtype=np.int64))
        assert orc.hamdist64(a, b) == orc.hamdist64_closed(a, b)


def test_grey_kats(orc):
    img = np.zeros((2, 2, 3), np.uint8)
    img[0, 0] = (255, 255, 255)
    img[0, 1] = (3, 3, 3)
    img[1, 0] = (1, 0, 0)
    img[1, 1] = (10, 20, 30)
    g = orc.grey(img)
    assert g[0, 0] == 255 and g[1, 0] == 0
    # three rounded products then truncation (d_mux_common.cu:16-20)
    c = np.float32(0.33333334)
    for (y, x) in [(0, 1), (1, 1)]:
        b, gg, r = [np.float32(v) * c for v in img[y, x]]
        assert g[y, x] == int(np.float32(np.float32(b + gg) + r))


def test_census_constant_image_is_zero(orc):
    g = np.full((20, 30), 77, np.uint8)
    assert not orc.census(g).any()


def test_census_bit_layout(orc):
    # one darker pixel at (y-1, x-4) relative to the centre -> sequence number 16 -> bit 31 of the low word
    g = np.full((15, 15), 100, np.uint8)
    g[6, 3] = 10
    c = orc.census(g)
    assert c[7, 7] == 1 << 31
    # (y+3, x+4) is the last appended bit -> bit 0 ; (y-3, x-4) is the first -> bit 47
    g = np.full((15, 15), 100, np.uint8); g[10, 11] = 10
    assert orc.census(g)[7, 7] == 1
    g = np.full((15, 15), 100, np.uint8); g[4, 3] = 10
    assert orc.census(g)[7, 7] == 1 << 47
    # centre row and column are skipped entirely (x != 0 && y != 0, d_ci_census.cu:41)
    g = np.full((15, 15), 100, np.uint8); g[7, 3] = 10; g[4, 7] = 10
    assert orc.census(g)[7, 7] == 0


def test_arms_constant_image(orc):
    H, W, usd = 23, 31, 9
    img = np.full((H, W, 3), 50, np.uint8)
    x = orc.cross_arms(img, 6, 20, usd, 4)
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.array_equal(x[0], np.minimum(usd, yy))
    assert np.array_equal(x[1], np.minimum(usd, H - 1 - yy))
    assert np.array_equal(x[2], np.minimum(usd, xx))
    assert np.array_equal(x[3], np.minimum(usd, W - 1 - xx))


def test_arm_includes_first_failing_pixel(orc):
    # value recorded before the colour test (SURVEY A-Q9): a step edge 3 px to the right gives arm 3, not 2
    img = np.full((5, 20, 3), 50, np.uint8)
    img[:, 10:] = 200
    x = orc.cross_arms(img, 6, 20, 9, 4)
    assert x[3][2, 7] == 3 and x[3][2, 9] == 1 and x[2][2, 10] == 1


def test_hsum_all_ones_is_arm_sum(orc):
    L, _ = rand_pair(24, 40, 3)
    x = orc.cross_arms(L, 6, 20, 9, 4)
    ones = np.ones((3, 24, 40), np.float32)
    h = orc.agg_hpass(ones, x)
    v = orc.agg_vpass(ones, x)
    assert np.array_equal(h[1], (x[2].astype(np.float32) + x[3]))  # half-open window: armL + armR elements
    assert np.array_equal(v[2], (x[0].astype(np.float32) + x[1]))


def test_wta_first_lowest_wins(orc):
    c = np.ones((5, 2, 3), np.float32)
    c[3, 0, 0] = 0.5
    c[1, 0, 1] = 0.5
    c[4, 0, 1] = 0.5  # tie with d=1 -> d=1 wins (strict >, d_dc_wta.cu:28)
    d = orc.dc_wta(c, 2)
    assert d[0, 0] == 1 and d[0, 1] == -1 and d[1, 2] == -2


def test_fish_identical_pair_known_answer(orc, golden):
    F, D, zd = golden["fish"], int(golden["params"][0]), int(golden["params"][1])
    cl, cr = orc.ci_adcensus(F, F, 10, 30, D, zd)
    assert np.array_equal(cl, golden["fish_cost_l"])
    assert not cl[zd].any() and not cr[zd].any()  # AD = 0 and census = 0 at zero offset
    x, a = orc.ca_cross(F, cl, 6, 20, 9, 4)
    assert not a[zd].any()
    disp = orc.dc_wta(a, zd)
    assert (disp <= 0).all()  # strict '>' picks the first zero, which is at or below zd


def test_dcc_classes(orc):
    dl = np.zeros((1, 8), np.float32)
    dr = np.zeros((1, 8), np.float32)
    dl[0, 2] = 3
    ol, orr = orc.dr_dcc(dl, dr)
    # L(2) -> R(5): |3 - 0| > 1 -> outlier.  Every dR = 0 maps x -> x, so L(2) is still hit -> class 1 (mismatch).
    assert ol[0, 2] == 1 and ol.sum() == 1
    # R(2) -> L(2): |0 - 3| > 1 -> outlier; L(2) maps to R(5), so nothing lands on R(2) -> class 2 (occlusion).
    assert orr[0, 2] == 2 and orr.sum() == 2


def test_gaussian_tables(orc):
    k = orc.gaussian_kernel_2d(2, 1.5)
    assert k.shape == (5, 5) and np.allclose(k, k.T) and k[2, 2] == k.max()
    v = np.float32(1.5) ** 2
    assert np.isclose(k[2, 2], 1.0 / (2 * np.float32(3.14159265359) * v), rtol=1e-6)
    k1 = orc.gaussian_kernel_1d(8, 5.0)
    assert np.all(np.diff(k1) < 0)


def test_bleed_rule(orc):
    img = np.zeros((7, 7), np.uint8)
    img[3, 2:5] = 1  # three set pixels in the 3x3 window of (3,3) and of (2,3)/(4,3)
    out = orc.filter_bleed_1(img, 1)
    assert out[2, 3] == 1 and out[4, 3] == 1 and out[0, 0] == 0


def test_mux_view_pattern(orc):
    N, H, W = 8, 16, 32
    views = [np.full((H, W, 3), v * 10, np.uint8) for v in range(N)]
    out = orc.mux_multiview(views, 18.43, H, W, 2)
    yi = orc.mux_y_interval(N, 18.43)
    assert abs(yi - 8.0) < 0.05
    # r_view = (3*tx + int(yv)) % N, g = r+1, b = r+2 (d_mux_multiview.cu:60-73); channels are B,G,R
    for ty in (0, 5):
        for tx in (0, 1, 7, 31):
            yv = np.float32(np.float32(ty % int(round(yi)) + 1.0) * np.float32(N)) * np.float32(np.float32(1.0) / np.float32(yi))
            r = (tx * 3 + int(yv)) % N
            assert tuple(out[ty, tx]) == (((r + 2) % N) * 10, ((r + 1) % N) * 10, r * 10)


def test_golden_vectors_regression(orc, golden):
    """The oracle must keep reproducing the committed vectors (generated by tests/golden/make_golden.py)."""
    g = golden
    D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = g["params"]
    D, zd, usd, lsd, ts, N = int(D), int(zd), int(usd), int(lsd), int(ts), int(N)
    L, R = g["L"], g["R"]
    assert np.array_equal(orc.grey(L), g["grey_l"])
    assert np.array_equal(orc.census(g["grey_l"]), g["census_l"])
    cl, cr = orc.ci_adcensus(L, R, ad, ce, D, zd)
    assert np.array_equal(cl, g["cost_l"]) and np.array_equal(cr, g["cost_r"])
    xl, al = orc.ca_cross(L, cl, ucd, lcd, usd, lsd)
    assert np.array_equal(xl, g["cross_l"]) and np.array_equal(al, g["acost_l"])
    assert np.array_equal(orc.dc_wta(al, zd), g["wta_l"])
    ol, orr = orc.dr_dcc(g["wta_l"], g["wta_r"])
    assert np.array_equal(ol, g["outl_l"]) and np.array_equal(orr, g["outl_r"])
    il, iol = orc.dr_irv(g["wta_l"], ol, xl, ts, th, D, zd, usd, 5, True)
    assert np.array_equal(il, g["irv_l"]) and np.array_equal(iol, g["irv_outl_l"])
    assert np.array_equal(orc.filter_bilateral_1(il, 7, 5.0, 10.0, D), g["bil_l"])
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    fr = orc.adcensus_stm(sbs, L.shape[0], L.shape[1], N, angle, D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th)
    assert np.array_equal(fr["interlaced"], g["frame_mux"]) and np.array_equal(fr["disp_l"], g["frame_disp_l"])


def test_cost_init_lut_equals_direct(orc):
    """rho via the 766/65-entry tables (what the HIP path indexes) == rho evaluated per pixel."""
    L, R = rand_pair(20, 33, 5)
    la, lc = orc.rho_luts(10.0, 30.0)
    cl, cr = orc.ci_adcensus(L, R, 10.0, 30.0, 6, 3)
    g_l, g_r = orc.grey(L), orc.grey(R)
    c_l, c_r = orc.census(g_l), orc.census(g_r)
    H, W = g_l.shape
    for d in (0, 3, 5):
        o = d - 3
        xr = np.clip(np.arange(W) + o, 0, W - 1)
        ad = np.abs(L.astype(np.int32) - R[:, xr].astype(np.int32)).sum(axis=2)
        x = (c_l ^ c_r[:, xr]).astype(np.uint64) & np.uint64(0xFFFFFFFF)
        ham = np.array([[bin(int(v) & 0x7FFFFFFF).count("1") + 33 * (int(v) >> 31) for v in row] for row in x])
        assert np.array_equal(cl[d], la[ad] + lc[ham])


@pytest.mark.skipif(not os.path.isdir(REF_IMG), reason="reference img/ only exists in the build container")
def test_bud_pair_matches_survey_understanding_check(orc, stm):
    """background plateau at offset -6 (~52 %), mean arm ~9 px, L/R outliers ~14.5 %."""
    L = stm.bmp_io.read_bmp(os.path.join(REF_IMG, "bud_2.bmp"))
    R = stm.bmp_io.read_bmp(os.path.join(REF_IMG, "bud_3.bmp"))
    assert L.shape == (384, 640, 3)
    cl, cr = orc.ci_adcensus(L, R, 10, 30, 32, 16)
    xl, al = orc.ca_cross(L, cl, 6, 20, 17, 8)
    xr, ar = orc.ca_cross(R, cr, 6, 20, 17, 8)
    dl, dr = orc.dc_wta(al, 16), orc.dc_wta(ar, 16)
    assert 0.50 < (dl == -6).mean() < 0.54
    assert 8.5 < xl.mean() < 9.5
    ol, _ = orc.dr_dcc(dl, dr)
    assert 0.13 < (ol > 0).mean() < 0.16
    assert 1.0e6 < al.max() < 1.2e6


def test_median_sorts_truncated_values_and_samples_by_flat_index(orc):
    """d_filter.cu:7-45.  (1) integer-valued interior pixels get the true 3x3 median; (2) the column step at x = 0
    wraps into the previous row of the flat buffer; (3) the sort compares int-truncated values and swaps write the
    truncated values back, checked against a literal Python transcription of that rule on a fractional image."""
    rng = np.random.RandomState(11)
    a = rng.randint(-20, 20, size=(6, 7)).astype(np.float32)
    out = orc.filter_median(a)
    for y in range(1, 5):
        for x in range(1, 6):
            assert out[y, x] == np.median(a[y - 1:y + 2, x - 1:x + 2])
    flat = a.ravel()
    y, x, W = 3, 0, 7
    win = [flat[(x + dx) + (y + dy) * W] for dy in (-1, 0, 1) for dx in (-1, 0, 1)]   # dx = -1 -> previous row's end
    assert out[y, x] == sorted(win)[4]

    f = (rng.random_sample((5, 6)) * 9 - 3).astype(np.float32)
    HW = f.size
    want = np.empty_like(f)
    for y in range(5):
        for x in range(6):
            v = [float(f.ravel()[min(max((x + dx) + (y + dy) * 6, 0), HW - 1)]) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
            for i in range(9):
                cur = int(v[i])
                for j in range(i, 9):
                    comp = int(v[j])
                    if comp < cur:
                        v[j], v[i], cur = float(cur), float(comp), comp
            want[y, x] = np.float32(v[4])
    assert np.array_equal(orc.filter_median(f), want)


def test_generate_gaussian_kernel_matches_oracle_table(orc, stm):
    """generateGaussianKernel (d_filter_gaussian.h:30) is a host-only helper of the library: no GPU needed."""
    import ctypes as C
    for r, s in [(10, 15.0), (7, 10.0), (0, 1.0)]:
        k = np.zeros((2 * r + 1) ** 2, np.float32)
        stm.lib().stm_generate_gaussian_kernel(k.ctypes.data_as(C.POINTER(C.c_float)), r, s)
        assert np.array_equal(k, orc.gaussian_kernel_2d(r, s).ravel())


def test_c1_full_size_golden_regression(orc):
    """BASELINE config 1 at its real size (bud_2 + bud_3, 640x384, D=32, 8 views): the oracle keeps reproducing the
    committed vectors of tests/golden/make_golden_c1.py, and the pair's statistics match SURVEY 8c's independent check
    (background plateau at offset -6 for about half of the pixels, about 14 % L/R outliers)."""
    import os
    from conftest import GOLDEN
    from stm_amd import bmp_io
    g = dict(np.load(os.path.join(GOLDEN, "bud_c1_golden.npz")))
    L, R = bmp_io.read_bmp(os.path.join(GOLDEN, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(GOLDEN, "bud_3.bmp"))
    assert L.shape == (384, 640, 3) and R.shape == (384, 640, 3)
    D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = [float(x) for x in g["params"]]
    D, zd, usd, lsd, ts, N = int(D), int(zd), int(usd), int(lsd), int(ts), int(N)
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    fr = orc.adcensus_stm(sbs, 384, 640, N, angle, D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th)
    assert np.array_equal(fr["wta_l"], g["frame_wta_l"].astype(np.float32)) and np.array_equal(fr["wta_r"], g["frame_wta_r"].astype(np.float32))
    assert np.array_equal(fr["disp_l"], g["frame_disp_l"]) and np.array_equal(fr["disp_r"], g["frame_disp_r"])
    assert np.array_equal(fr["interlaced"], g["frame_mux"])
    mode, share, outl = g["stats"]
    assert mode == -6 and 0.45 < share < 0.58 and 0.10 < outl < 0.20


# ---------------------------------------------------------------------------------------------------------------------
# Non-default oracle modes: ref_quirks (A-Q7) and paper_ratio (A-Q17 iv)

def _ref_tile_partner(kind, D, zd, W, x, d, side):
    """Which (image, column) the reference's live cost kernels read for pixel x, hypothesis d: their shared-tile index
    arithmetic replayed on a tile of (image, column) tags.  kind 'census': ci_census_kernel_6 (d_ci_census.cu:221-246) with
    the padding of d_ci_adcensus.cu:117-120; kind 'ad': ci_ad_kernel_5 (d_ci_ad.cu:100-144) with d_ci_adcensus.cu:57-59.
    Both tiles of a block sit back to back in ONE dynamic shared array, so an index one past a tile lands in the other."""
    bw = 160
    gx0, tx = x - x % bw, x % bw
    cl = lambda g: min(max(g, 0), W - 1)
    if kind == "census":
        pad_l, pad_r = zd - 1, D - zd
        cols = bw + D - 1
        sm = [("L", cl(gx0 - pad_r + j)) for j in range(cols)] + [("R", cl(gx0 - pad_l + j)) for j in range(cols)]
        idx = cols + tx + pad_l + (d - zd) if side == "l" else tx + pad_r - (d - zd)
    else:
        P = (D - zd) if (D - zd) > zd else zd - 1
        cols = bw + 2 * P
        sm = [("L", cl(gx0 - P + j)) for j in range(cols)] + [("R", cl(gx0 - P + j)) for j in range(cols)]
        idx = cols + tx + P + (d - zd) if side == "l" else tx + P - (d - zd)
    assert 0 <= idx < len(sm), "the read leaves the dynamic shared allocation: undefined, not a quirk"
    return sm[idx]


@pytest.mark.parametrize("D,zd", [(32, 16), (32, 15), (32, 20), (64, 32), (64, 10)])
def test_ref_quirk_index_claims(D, zd):
    """SURVEY Appendix A, last paragraph: with block width 160 the census kernel strays at (tx = 0, d = 0) on the left cost
    and (tx = 159, d = 0) on the right cost for every (D, zd); the AD kernel exactly when D - zd <= zd; every other read is
    the clean clamped partner."""
    W = 320
    cl = lambda g: min(max(g, 0), W - 1)
    for kind in ("census", "ad"):
        strays = set()
        for x in range(W):
            for d in range(D):
                if _ref_tile_partner(kind, D, zd, W, x, d, "l") != ("R", cl(x + d - zd)):
                    strays.add(("l", x % 160, d))
                    assert _ref_tile_partner(kind, D, zd, W, x, d, "l") == ("L", cl(x + 160 + zd - 2))
                if _ref_tile_partner(kind, D, zd, W, x, d, "r") != ("L", cl(x - (d - zd))):
                    strays.add(("r", x % 160, d))
                    assert _ref_tile_partner(kind, D, zd, W, x, d, "r") == ("R", cl(x - 158 - zd))
        want = {("l", 0, 0), ("r", 159, 0)} if (kind == "census" or D - zd <= zd) else set()
        assert strays == want, (kind, D, zd, strays)


@pytest.mark.parametrize("D,zd", [(32, 16), (64, 10)])
def test_ref_quirks_mode_of_the_oracle(orc, D, zd):
    """orc.set_ref_quirks(1): plane 0 differs from the clean volume in columns 160 k (left) / 160 k + 159 (right) only, and
    there it is the cost of the pair the reference's tile arithmetic reads (replayed above), rebuilt from the oracle's own
    grey / census / Hamming / rho pieces."""
    H, W = 3, 320
    L, R = rand_pair(H, W, seed=5)
    clean_l, clean_r = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    orc.set_ref_quirks(1)
    try:
        q_l, q_r = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    finally:
        orc.set_ref_quirks(0)
    assert np.array_equal(q_l[1:], clean_l[1:]) and np.array_equal(q_r[1:], clean_r[1:])
    cols_l, cols_r = np.arange(0, W, 160), np.arange(159, W, 160)
    keep_l, keep_r = np.ones(W, bool), np.ones(W, bool)
    keep_l[cols_l] = False
    keep_r[cols_r] = False
    assert np.array_equal(q_l[0][:, keep_l], clean_l[0][:, keep_l]) and np.array_equal(q_r[0][:, keep_r], clean_r[0][:, keep_r])
    lut_ad, lut_c = orc.rho_luts(10.0, 30.0)
    cen = {"L": orc.census(orc.grey(L)), "R": orc.census(orc.grey(R))}
    img = {"L": L.astype(np.int32), "R": R.astype(np.int32)}
    for side, own, vol, cols in (("l", "L", q_l, cols_l), ("r", "R", q_r, cols_r)):
        for y in range(H):
            for x in cols:
                ci, cx = _ref_tile_partner("census", D, zd, W, int(x), 0, side)
                ai, ax = _ref_tile_partner("ad", D, zd, W, int(x), 0, side)
                sad = int(np.abs(img[own][y, x] - img[ai][y, ax]).sum())
                ham = orc.hamdist64(int(cen[own][y, x]), int(cen[ci][y, cx]))
                assert vol[0, y, x] == np.float32(lut_ad[sad] + lut_c[ham]), (side, y, x)
    assert not np.array_equal(q_l[0], clean_l[0])  # the mode does change something on a random pair


def test_paper_ratio_mode_of_the_oracle(orc):
    """orc.set_irv_paper_ratio(1): region voting accepts on (winning count) / S instead of (winning bin index) / S
    (d_dr_irv.cu:36, SURVEY A-Q17 iv).  A 9x9 flat patch with one outlier in the middle: 80 votes for bin 3 + zd.  With
    zd = 0 and thresh_h = 0.4 the reference's rule sees 3 / 80 < 0.4 (reject), the paper's 80 / 80 > 0.4 (accept)."""
    H = W = 9
    disp = np.full((H, W), 3.0, np.float32)
    disp[4, 4] = 7.0
    outl = np.zeros((H, W), np.uint8)
    outl[4, 4] = 1
    cross = np.zeros((4, H, W), np.uint8)
    for y in range(H):
        for x in range(W):
            cross[:, y, x] = (y, H - 1 - y, x, W - 1 - x)
    d0, o0 = orc.dr_irv(disp, outl, cross, 20, 0.4, 16, 0, 8, 1)
    assert d0[4, 4] == 7.0 and o0[4, 4] == 1
    orc.set_irv_paper_ratio(1)
    try:
        d1, o1 = orc.dr_irv(disp, outl, cross, 20, 0.4, 16, 0, 8, 1)
    finally:
        orc.set_irv_paper_ratio(0)
    assert d1[4, 4] == 3.0 and o1[4, 4] == 0
