"""Second, independent restatements of stages whose only statement so far was oracle/stm_oracle.c.

Each function below is written in plain numpy / Python loops from the prose of SURVEY.md (section 8a and Appendix A),
not from the C code, and is run on small inputs.  Integer stages must agree exactly.  Float stages are restated in
float64 ("textbook" arithmetic), so they pin the semantics -- window shape, border rule, weights, LUT index -- and
are compared with a tolerance; the exact float32 operation order is pinned by the GPU-vs-oracle parity tests.
CPU only: these run in the `-m "not gpu"` suite.
"""
import numpy as np
import pytest

from conftest import rand_pair


def _arms_py(img, ucd, lcd, usd, lsd):
    """Appendix A-Q9: record arm = k first, then test; near tier (k <= lsd): anchor-vs-cur or prev-vs-cur > lcd;
    far tier: anchor-vs-cur > ucd; stop at the border keeping the last value."""
    H, W, _ = img.shape
    im = img.astype(np.int32)
    out = np.zeros((4, H, W), np.uint8)
    steps = [(-1, 0), (1, 0), (0, -1), (0, 1)]  # UP, DOWN, LEFT, RIGHT
    for a, (dy, dx) in enumerate(steps):
        for y in range(H):
            for x in range(W):
                arm, prev = 0, im[y, x]
                for k in range(1, usd + 1):
                    yy, xx = y + dy * k, x + dx * k
                    if not (0 <= yy < H and 0 <= xx < W):
                        break
                    cur = im[yy, xx]
                    arm = k
                    d_anchor = int(np.max(np.abs(cur - im[y, x])))
                    if k > lsd:
                        if d_anchor > ucd:
                            break
                    else:
                        if d_anchor > lcd or int(np.max(np.abs(cur - prev))) > lcd:
                            break
                        prev = cur
                out[a, y, x] = arm
    return out


def test_arms_second_opinion(orc):
    L, _ = rand_pair(14, 19, 5)
    L = (L // 24 * 24).astype(np.uint8)  # quantised colours: plateaus, so arms of every length occur
    cost = np.zeros((1, 14, 19), np.float32)
    for (ucd, lcd, usd, lsd) in [(6.0, 20.0, 9, 4), (30.0, 10.0, 5, 5), (0.0, 0.0, 17, 1)]:
        cross, _ = orc.ca_cross(L, cost, ucd, lcd, usd, lsd)
        assert np.array_equal(cross, _arms_py(L, ucd, lcd, usd, lsd))


def _aggregate_py(cost, cross):
    """Appendix A-Q10/Q11: passes H, V, V, H; half-open windows [p - arm_lo, p + arm_hi); no normalisation."""
    D, H, W = cost.shape
    up, down, left, right = [cross[i].astype(int) for i in range(4)]
    a = cost.astype(np.float64)

    def hpass(v):
        o = np.zeros_like(v)
        for y in range(H):
            for x in range(W):
                o[:, y, x] = v[:, y, x - left[y, x]:x + right[y, x]].sum(axis=1)
        return o

    def vpass(v):
        o = np.zeros_like(v)
        for y in range(H):
            for x in range(W):
                o[:, y, x] = v[:, y - up[y, x]:y + down[y, x], x].sum(axis=1)
        return o

    return hpass(vpass(vpass(hpass(a))))


def test_aggregation_second_opinion(orc):
    L, _ = rand_pair(13, 17, 9)
    L = (L // 32 * 32).astype(np.uint8)
    cost = np.random.RandomState(3).random_sample((3, 13, 17)).astype(np.float32)
    cross, acost = orc.ca_cross(L, cost, 6.0, 20.0, 7, 3)
    want = _aggregate_py(cost, cross)
    assert np.allclose(acost, want, rtol=2e-5, atol=1e-6)
    # the right border pixel of a row excludes itself in the H passes (arm right = 0 there): spot-check the rule
    assert np.all(cross[3][:, -1] == 0) and np.all(cross[2][:, 0] == 0)


def _irv_py(disp, outl, cross, thresh_s, thresh_h, zd, usd, nbins):
    """8a row a16 / Appendix A-Q17, one vote + one apply (host flavour with iterations = 1)."""
    H, W = disp.shape
    up, down, left, right = [cross[i].astype(int) for i in range(4)]
    new_d, new_o = disp.copy(), outl.copy()
    for y in range(H):
        for x in range(W):
            if outl[y, x] == 0:
                continue
            hist = np.zeros(nbins, np.int64)
            for yy in range(y - min(up[y, x], usd), y + down[y, x] + 1):       # rows inclusive
                for xx in range(x - left[yy, x], x + right[yy, x] + 1):        # arms of the row's pixel in column x
                    if outl[yy, xx] == 0:
                        hist[int(disp[yy, xx]) + zd] += 1
            S = int(hist.sum())
            max_d = int(disp[y, x])
            best = 0
            for b in range(nbins):
                if hist[b] > best:
                    best, max_d = int(hist[b]), b - zd
            ratio = np.float32(max_d + zd) / np.float32(S) if S else np.float32(np.inf)
            if S > thresh_s and ratio > np.float32(thresh_h):   # bin INDEX over S, not the winning count (Q17 iv)
                new_d[y, x] = np.float32(max_d)
                new_o[y, x] = 0
    return new_d, new_o


@pytest.mark.parametrize("thresh_s,thresh_h", [(3, 0.4), (0, 0.0), (12, 1.5)])
def test_irv_second_opinion(orc, thresh_s, thresh_h):
    H, W, D, zd, usd, lsd = 16, 21, 9, 4, 6, 3
    L, _ = rand_pair(H, W, 21)
    L = (L // 40 * 40).astype(np.uint8)
    rng = np.random.RandomState(8)
    cross, _ = orc.ca_cross(L, np.zeros((1, H, W), np.float32), 6.0, 20.0, usd, lsd)
    disp = rng.randint(-zd, D - zd, size=(H, W)).astype(np.float32)
    outl = (rng.random_sample((H, W)) < 0.3).astype(np.uint8) * rng.randint(1, 3, size=(H, W)).astype(np.uint8)
    got_d, got_o = orc.dr_irv(disp, outl, cross, thresh_s, thresh_h, D, zd, usd, 1, device_flavour=False)
    want_d, want_o = _irv_py(disp, outl, cross, thresh_s, thresh_h, zd, usd, max(D, 65))
    assert np.array_equal(got_o, want_o)
    assert np.array_equal(got_d, want_d)
    assert (want_o != outl).any() or thresh_s == 12  # the lenient settings must actually accept something


def test_bilateral_second_opinion(orc):
    """8a row a17 / A-Q18 in float64: weights Gs * Gc[(int)|v0 - v|], clamp-to-edge, sum(w v) / sum(w)."""
    rng = np.random.RandomState(5)
    H, W, r, sc, ss, D = 11, 13, 3, 5.0, 10.0, 16
    img = (rng.random_sample((H, W)) * 12 - 6).astype(np.float32)
    got = orc.filter_bilateral_1(img, r, sc, ss, D)
    pi = 3.14159265359
    gc = np.array([np.exp(-(i * i) / (2 * sc * sc)) / np.sqrt(2 * pi * sc * sc) for i in range(D)])
    want = np.zeros((H, W))
    for y in range(H):
        for x in range(W):
            num = den = 0.0
            for dy in range(-r, r + 1):
                for dx in range(-r, r + 1):
                    v = float(img[min(max(y + dy, 0), H - 1), min(max(x + dx, 0), W - 1)])
                    gs = np.exp(-(dx * dx + dy * dy) / (2 * ss * ss)) / (2 * pi * ss * ss)
                    w = gs * gc[int(abs(float(img[y, x]) - v))]
                    num += w * v
                    den += w
            want[y, x] = num / den
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_hit_maps_masks_and_view_synthesis_second_opinion(orc):
    """8a rows a18, a20-a22 (A-Q19..Q22): hit maps, mask, backward warp with truncated sample x, grow-only gaussian
    blend weight m = max(1 - maskR, blur(1 - maskR)), merge with separately truncated terms and u8 wrap-around."""
    H, W = 9, 23
    rng = np.random.RandomState(13)
    L = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    R = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    dl = (rng.randint(-5, 4, size=(H, W)) + rng.random_sample((H, W)) * 0.9).astype(np.float32)
    dr = (rng.randint(-5, 4, size=(H, W)) + rng.random_sample((H, W)) * 0.9).astype(np.float32)

    occl_l, occl_r = orc.dibr_occl(dl, dr)
    wl, wr = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    for y in range(H):
        for x in range(W):
            wr[y, min(max(x + int(dl[y, x] * np.float32(1.0)), 0), W - 1)] = 1    # left pixels land in the right view
            wl[y, min(max(x + int(dr[y, x] * np.float32(-1.0)), 0), W - 1)] = 1   # right pixels land in the left view
    assert np.array_equal(occl_l, wl) and np.array_equal(occl_r, wr)

    ml, mr = orc.dibr_occl_to_mask(occl_l, occl_r)
    assert np.array_equal(ml, (occl_l == 1).astype(np.float32)) and np.array_equal(mr, (occl_r == 1).astype(np.float32))

    shift, g_r, g_s = np.float32(0.6), 2, 1.5
    got = orc.dibr_dbm(L, R, dl, dr, ml, mr, float(shift), g_r, g_s)
    blend = orc.filter_gaussian_1((np.float32(1.0) - mr).astype(np.float32), g_r, g_s)   # its own KAT: below
    want = np.zeros_like(L)
    for y in range(H):
        for x in range(W):
            sxl = int(min(max(np.float32(x) + dr[y, x] * np.float32(-shift), np.float32(0)), np.float32(W - 1)))
            sxr = int(min(max(np.float32(x) + dl[y, x] * np.float32(np.float32(1.0) - shift), np.float32(0)), np.float32(W - 1)))
            m = blend[y, x]
            for c in range(3):
                out_l = np.uint8(np.float32(L[y, sxl, c]) * mr[y, x])
                out_r = np.uint8(np.float32(R[y, sxr, c]) * ml[y, x])
                # merge(b = outL, a = outR, m): (u8)((1 - m) * b) + (u8)(m * a), u8 wrap-around
                t0 = int(np.float32(np.float32(1.0) - m) * np.float32(out_l))
                t1 = int(m * np.float32(out_r))
                want[y, x, c] = (t0 + t1) & 0xFF
    assert np.array_equal(got, want)


def test_grow_only_gaussian_second_opinion(orc):
    """A-Q21: clamp border, blur = sum(w v) / sum(w) (d_filter_gaussian.cu:77-84), result max(in, blur)."""
    rng = np.random.RandomState(2)
    H, W, r, s = 8, 10, 2, 1.2
    img = (rng.random_sample((H, W)) > 0.6).astype(np.float32)
    got = orc.filter_gaussian_1(img, r, s)
    pi = 3.14159265359
    want = np.zeros((H, W))
    for y in range(H):
        for x in range(W):
            acc = den = 0.0
            for dy in range(-r, r + 1):
                for dx in range(-r, r + 1):
                    v = float(img[min(max(y + dy, 0), H - 1), min(max(x + dx, 0), W - 1)])
                    w = np.exp(-(dx * dx + dy * dy) / (2 * s * s)) / (2 * pi * s * s)
                    acc += v * w
                    den += w
            want[y, x] = max(float(img[y, x]), acc / den)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_wta_and_dcc_second_opinion(orc):
    """A-Q13 / A-Q16: first strictly-lowest hypothesis; L/R check with (int) truncation and threshold 1.0."""
    rng = np.random.RandomState(17)
    D, H, W, zd = 7, 6, 15, 3
    vol = rng.randint(0, 4, size=(D, H, W)).astype(np.float32)   # many ties
    assert np.array_equal(orc.dc_wta(vol, zd), (np.argmin(vol, axis=0) - zd).astype(np.float32))
    dl = (rng.randint(-3, 4, size=(H, W)) + rng.random_sample((H, W)) * 0.5).astype(np.float32)
    dr = (rng.randint(-3, 4, size=(H, W)) + rng.random_sample((H, W)) * 0.5).astype(np.float32)
    ol, orr = orc.dr_dcc(dl, dr)
    hit_l, hit_r = np.ones((H, W), np.uint8), np.ones((H, W), np.uint8)   # 1 = never hit
    bad_l, bad_r = np.zeros((H, W), bool), np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            xl = min(max(x + int(dl[y, x]), 0), W - 1)
            xr = min(max(x - int(dr[y, x]), 0), W - 1)
            bad_l[y, x] = abs(dl[y, x] - dr[y, xl]) > 1.0
            bad_r[y, x] = abs(dr[y, x] - dl[y, xr]) > 1.0
            hit_r[y, xl] = 0     # a left pixel maps onto the right view at xl
            hit_l[y, xr] = 0
    want_l = np.where(bad_l, np.where(hit_l == 1, 2, 1), 0).astype(np.uint8)
    want_r = np.where(bad_r, np.where(hit_r == 1, 2, 1), 0).astype(np.uint8)
    assert np.array_equal(ol, want_l) and np.array_equal(orr, want_r)
