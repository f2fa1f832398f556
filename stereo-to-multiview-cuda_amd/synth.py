"""Seeded synthetic stereo pairs: value-noise left image, piecewise-constant
integer disparity field (background plane + rectangles), right image by forward mapping with z-order.

Convention: hypothesis index d means horizontal offset o = d - zero_disp and the left
cost pairs L(x) with R(x + o), so a left pixel with offset `off` appears in the right image at x + off.
"""
import numpy as np

SEED = 0x5EED0001


def _value_noise(rng, H, W, cell):
    gh, gw = H // cell + 2, W // cell + 2
    g = rng.random_sample((gh, gw)).astype(np.float32)
    ys = np.arange(H, dtype=np.float32) / cell
    xs = np.arange(W, dtype=np.float32) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    fy = fy * fy * (3 - 2 * fy)
    fx = fx * fx * (3 - 2 * fx)
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def left_image(H, W, seed=SEED):
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    base = 0.6 * _value_noise(rng, H, W, 64) + 0.3 * _value_noise(rng, H, W, 16) + 0.1 * _value_noise(rng, H, W, 4)
    img = np.empty((H, W, 3), np.float32)
    for c in range(3):
        own = 0.6 * _value_noise(rng, H, W, 64) + 0.3 * _value_noise(rng, H, W, 16) + 0.1 * _value_noise(rng, H, W, 4)
        img[:, :, c] = 0.7 * base + 0.3 * own
    lo, hi = img.min(), img.max()
    return np.clip((img - lo) / (hi - lo) * 255.0, 0, 255).astype(np.uint8)


def disparity_field(H, W, num_disp, zero_disp, seed=SEED, n_rect=12):
    rng = np.random.RandomState((seed ^ 0x9E3779B9) & 0x7FFFFFFF)
    lo, hi = -(zero_disp - 1) + 2, (num_disp - zero_disp - 1) - 2
    if lo > hi:
        lo = hi = 0
    off = np.full((H, W), max(min(-4, hi), lo), np.int32)
    for _ in range(n_rect):
        rh = int(H * rng.uniform(0.05, 0.25))
        rw = int(W * rng.uniform(0.05, 0.25))
        y0 = rng.randint(0, max(H - rh, 1))
        x0 = rng.randint(0, max(W - rw, 1))
        off[y0:y0 + rh, x0:x0 + rw] = rng.randint(lo, hi + 1)
    return off


def right_image(left, off):
    H, W, _ = left.shape
    right = np.zeros_like(left)
    filled = np.zeros((H, W), bool)
    xs = np.arange(W)[None, :].repeat(H, 0)
    for o in sorted(np.unique(off).tolist(), reverse=True):  # far (large o) first, near (very negative o) last
        m = off == o
        tx = xs + o
        ok = m & (tx >= 0) & (tx < W)
        yy, xx = np.nonzero(ok)
        right[yy, xx + o] = left[yy, xx]
        filled[yy, xx + o] = True
    # holes take the value of their left neighbour (first column: the left image's pixel)
    for x in range(W):
        h = ~filled[:, x]
        if not h.any():
            continue
        right[h, x] = left[h, x] if x == 0 else right[h, x - 1]
    return right


def stereo_pair(H, W, num_disp, zero_disp, seed=SEED):
    """Returns (left, right, offset_field): BGR u8 [H][W][3] x2 and the integer ground-truth offsets."""
    L = left_image(H, W, seed)
    off = disparity_field(H, W, num_disp, zero_disp, seed)
    R = right_image(L, off)
    return L, R, off


def sbs_frame(H, W, num_disp, zero_disp, seed=SEED):
    """Side-by-side frame [H][2W][3] as adcensus_stm expects (left half = left view, d_demux_common.cu:16-31)."""
    L, R, off = stereo_pair(H, W, num_disp, zero_disp, seed)
    return np.ascontiguousarray(np.concatenate([L, R], axis=1)), off


def tiled_pair(left, right, H, W):
    """A real-content stereo pair of any size from a small one (the reference's 640x384 img/bud_2 + bud_3): repeated
    horizontally (keeps the sign of every disparity), mirrored vertically on every other repeat (a vertical flip leaves
    a rectified pair rectified), cropped to H x W.  The seams are ordinary depth/colour edges; what matters is that arm
    lengths, outlier density and disparity statistics are those of real content (14.5 % L/R outliers on this pair against
    2 % on the value-noise frame)."""
    h, w, _ = left.shape
    ny, nx = (H + h - 1) // h, (W + w - 1) // w

    def tile(img):
        rows = [np.concatenate([img if (j % 2 == 0) else img[::-1] for _ in range(nx)], axis=1) for j in range(ny)]
        return np.ascontiguousarray(np.concatenate(rows, axis=0)[:H, :W])
    return tile(left), tile(right)


def tiled_sbs_frame(left, right, H, W):
    L, R = tiled_pair(left, right, H, W)
    return np.ascontiguousarray(np.concatenate([L, R], axis=1))
