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


# ------------------------------------------------------------------------------------------------ HSLO (a14)
def _hslo_py(cost, img_l, img_r, T, H1, H2, zd):
    """Mei et al. section 3.3 as DESIGN.md section 2 fixes it: four scan directions; along direction r
    Cr(p,d) = C(p,d) + min(Cr(p-r,d), Cr(p-r,d+-1) + P1, min_k Cr(p-r,k) + P2) - min_k Cr(p-r,k); the first pixel of a line
    takes C(p,d); (P1, P2) from the colour steps D1 (own image, integer mean of B,G,R) and D2 (other image at the matched
    pixel x + (d - zd), clamped; float mean): both < T -> (H1, H2); exactly one strictly on each side -> / 4; else / 10.
    Final cost = mean of the four directions; WTA with the first strictly-lowest rule.  float64 throughout."""
    D, H, W = cost.shape
    c = cost.astype(np.float64)
    mean_l = (img_l.astype(np.int64).sum(axis=2) // 3).astype(np.float64)            # u8 integer average
    # the other image's mean is a float32 value ((float)(sum / 3.0)) and so is the difference of two of them: at D2 == T the
    # class depends on that rounding, so it is part of the stage's definition
    mean_r = (img_r.astype(np.int64).sum(axis=2).astype(np.float64) / 3.0).astype(np.float32)
    total = np.zeros_like(c)
    for (dx, dy) in [(1, 0), (-1, 0), (0, 1), (0, -1)]:
        acc = np.zeros_like(c)
        lines = range(H) if dx else range(W)
        n = W if dx else H
        for line in lines:
            prev = None
            for i in range(n):
                if dx:
                    y, x = line, (i if dx > 0 else W - 1 - i)
                else:
                    x, y = line, (i if dy > 0 else H - 1 - i)
                if prev is None:
                    cur = c[:, y, x].copy()
                else:
                    px, py = x - dx, y - dy
                    m = prev.min()
                    D1 = abs(mean_l[y, x] - mean_l[py, px])
                    cur = np.empty(D)
                    for d in range(D):
                        o = d - zd
                        qx, qpx = min(max(x + o, 0), W - 1), min(max(px + o, 0), W - 1)
                        D2 = float(np.abs(np.float32(mean_r[y, qx] - mean_r[py, qpx])))
                        if D1 < T and D2 < T:
                            P1, P2 = H1, H2
                        elif (D1 < T and D2 > T) or (D1 > T and D2 < T):
                            P1, P2 = H1 / 4.0, H2 / 4.0
                        else:
                            P1, P2 = H1 / 10.0, H2 / 10.0
                        best = prev[d]
                        if d > 0:
                            best = min(best, prev[d - 1] + P1)
                        if d < D - 1:
                            best = min(best, prev[d + 1] + P1)
                        best = min(best, m + P2)
                        cur[d] = c[d, y, x] + best - m
                acc[:, y, x] = cur
                prev = cur
        total += acc
    total *= 0.25
    return total


def test_hslo_second_opinion(orc):
    """The path costs of the four directions and their mean (float64 restatement vs the float32 oracle), and the WTA map
    wherever the float64 minimum is clear of float32 rounding."""
    H, W, D, zd = 9, 13, 6, 2
    L, R = rand_pair(H, W, 31)
    cl, _ = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    disp, vol = orc.dc_hslo(cl, L, R, 15.0, 1.0, 3.0, zd, return_cost=True)
    want = _hslo_py(cl, L, R, 15.0, 1.0, 3.0, zd)
    assert np.max(np.abs(vol.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)) < 1e-5
    srt = np.sort(want, axis=0)
    clear = (srt[1] - srt[0]) > 1e-4  # pixels whose best hypothesis is not a near-tie
    assert clear.mean() > 0.5
    assert np.array_equal(disp[clear], (np.argmin(want, axis=0) - zd).astype(np.float32)[clear])
    # penalty classes: same costs, an image with a strong vertical edge (steps > T: penalties / 10 there) vs a flat image
    flat = np.full((H, W, 3), 90, np.uint8)
    edge = flat.copy()
    edge[:, W // 2:] = 200
    _, v_flat = orc.dc_hslo(cl, flat, flat, 15.0, 1.0, 3.0, zd, return_cost=True)
    _, v_edge = orc.dc_hslo(cl, edge, edge, 15.0, 1.0, 3.0, zd, return_cost=True)
    assert np.allclose(v_flat, _hslo_py(cl, flat, flat, 15.0, 1.0, 3.0, zd), rtol=1e-5, atol=1e-5)
    assert np.allclose(v_edge, _hslo_py(cl, edge, edge, 15.0, 1.0, 3.0, zd), rtol=1e-5, atol=1e-5)
    assert float(v_edge.mean()) < float(v_flat.mean())  # smaller penalties across the edge: cheaper paths on average


# ------------------------------------------------------------------------------------------------ forward warp (a23)
def _fwarp_py(img_l, disp_l, shift):
    """Scatter out[clamp(x + int(disp * shift))] = in[x]; among sources that land on one target the LARGEST x wins (the
    deterministic rule DESIGN.md fixes for the reference's racy scatter); untouched targets stay 0."""
    H, W, _ = img_l.shape
    out = np.zeros_like(img_l)
    for y in range(H):
        owner = {}
        for x in range(W):
            sd = int(np.float32(disp_l[y, x]) * np.float32(shift))  # int() truncates toward zero like the C cast
            owner[min(max(x + sd, 0), W - 1)] = x
        for t, x in owner.items():
            out[y, t] = img_l[y, x]
    return out


def test_forward_warp_second_opinion(orc):
    rng = np.random.RandomState(8)
    H, W = 11, 23
    L, R = rand_pair(H, W, 8)
    dl = rng.randint(-9, 10, size=(H, W)).astype(np.float32) + rng.choice([0.0, 0.25, 0.5], size=(H, W)).astype(np.float32)
    for shift in (0.5, 1.0, -0.75, 0.0):
        assert np.array_equal(orc.dibr_dfm(L, R, dl, dl, shift), _fwarp_py(L, dl, shift))


# ------------------------------------------------------------------------------------------------ 3x3 "median" (d_filter.cu:7-45)
def _median_py(img):
    """Nine samples in[(x+dx) + (y+dy) W] of the FLAT buffer (a column step off the row wraps into the neighbouring row; a
    flat index outside the buffer is clamped), selection sort on the values truncated to int where a swap stores the
    truncated ints back; the result is slot 4."""
    H, W = img.shape
    flat = img.ravel()
    out = np.empty(H * W, np.float32)
    for y in range(H):
        for x in range(W):
            v = []
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    q = min(max((x + dx) + (y + dy) * W, 0), H * W - 1)
                    v.append(float(flat[q]))
            for i in range(9):
                cur = int(v[i])
                for j in range(i, 9):
                    comp = int(v[j])
                    if comp < cur:
                        v[j], v[i] = float(cur), float(comp)
                        cur = comp
            out[y * W + x] = v[4]
    return out.reshape(H, W)


@pytest.mark.parametrize("H,W", [(7, 9), (3, 3), (1, 5)])
def test_median_second_opinion(orc, H, W):
    rng = np.random.RandomState(H * 31 + W)
    img = (rng.randint(-6, 7, size=(H, W)) + rng.choice([0.0, 0.5, 0.75], size=(H, W))).astype(np.float32)
    assert np.array_equal(orc.filter_median(img), _median_py(img))


# ------------------------------------------------------------------------------------------------ tx_scale (d_tx_scale.cu:8-52)
def _bilinear_py(img, out_rows, out_cols, as_u8):
    """Sample position (gx / out_cols * in_cols, gy / out_rows * in_rows) clamped to the image, floor corner, +1 neighbour
    clamped at the border, weights from the fractional parts; u8 images truncate the blend."""
    H, W = img.shape[:2]
    src = img.astype(np.float64)
    out = np.zeros((out_rows, out_cols) + img.shape[2:], np.float64)
    for gy in range(out_rows):
        for gx in range(out_cols):
            xs = min(max(float(np.float32(np.float32(gx) / np.float32(out_cols)) * np.float32(W)), 0.0), W - 1.0)
            ys = min(max(float(np.float32(np.float32(gy) / np.float32(out_rows)) * np.float32(H)), 0.0), H - 1.0)
            x0, y0 = int(np.floor(xs)), int(np.floor(ys))
            x1, y1 = min(x0 + 1, W - 1), min(y0 + 1, H - 1)
            wx, wy = xs - x0, ys - y0
            top = src[y0, x0] * (1 - wx) + src[y0, x1] * wx
            bot = src[y1, x0] * (1 - wx) + src[y1, x1] * wx
            out[gy, gx] = top * (1 - wy) + bot * wy
    return out


def test_tx_scale_second_opinion(orc):
    L, _ = rand_pair(12, 20, 3)
    for (h, w) in [(6, 10), (24, 40), (7, 13), (12, 20)]:
        got = orc.tx_scale_bilinear(L, h, w).astype(np.float64)
        want = _bilinear_py(L, h, w, True)
        # u8 truncation of a float32 blend: equal to the float64 blend's floor except where the blend sits within rounding
        # distance of an integer
        diff = got - np.floor(want + 1e-4)
        near_int = np.abs(want - np.round(want)) < 1e-3
        assert np.all((diff == 0) | near_int) and np.abs(got - want).max() < 1.0 + 1e-3
    rng = np.random.RandomState(5)
    d = (rng.randint(-8, 9, size=(9, 14)) + rng.random_sample((9, 14))).astype(np.float32)
    for (h, w, scale) in [(18, 28, 2.0), (9, 14, 1.0), (13, 9, 0.5)]:
        got = orc.tx_disp_scale(d, h, w, scale).astype(np.float64)
        want = _bilinear_py(d, h, w, False) * scale
        assert np.max(np.abs(got - want)) < 1e-4
