"""Region statistics of the outliers of the real-content 1080p frame (CPU, oracle): how much of an outlier's cross region (rows
y - up .. y + down of its column) does it share with the outlier directly above it?  usage: python tools/irv_stats.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import stm_amd
from stm_amd import synth, bmp_io, device_api as dev
from oracle import pyoracle as orc
G = os.path.join("tests", "golden")
H, W, D, zd = 1080, 1920, 64, 32
for name in ("real", "synthetic"):
    if name == "real": sbs = synth.tiled_sbs_frame(bmp_io.read_bmp(os.path.join(G, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(G, "bud_3.bmp")), H, W)
    else: sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    t = time.time()
    r = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd, p.thresh_s, p.thresh_h, stop_after_wta=True)
    ol, orr = orc.dr_dcc(r["wta_l"], r["wta_r"])
    L, R = orc.demux_sbs(sbs, W)
    for vname, img, o in (("L", L, ol), ("R", R, orr)):
        arms = orc.cross_arms(img, p.ucd, p.lcd, p.usd, p.lsd).astype(np.int32)  # up, down, left, right ?
        up, down = arms[0], arms[1]
        ys, xs = np.nonzero(o)
        n = len(ys)
        top = ys - np.minimum(up[ys, xs], ys); bot = ys + np.minimum(down[ys, xs], H - 1 - ys)
        rows = bot - top + 1
        above = (ys > 0) & (o[np.maximum(ys - 1, 0), xs] != 0)
        ya = np.maximum(ys - 1, 0)
        top_a = ya - np.minimum(up[ya, xs], ya); bot_a = ya + np.minimum(down[ya, xs], H - 1 - ya)
        delta = np.abs(top - top_a) + np.abs(bot - bot_a)
        inc = np.where(above, np.minimum(delta, rows), rows)  # row steps with incremental update from the outlier above
        left = (xs > 0) & (o[ys, np.maximum(xs - 1, 0)] != 0)
        print("%s %s: outliers %d (%.1f%%), mean region rows %.1f; with an outlier directly above %.1f%%, identical row range %.1f%%, mean |delta| %.2f; row steps: %.0f k now, %.0f k incremental (x%.2f); with outlier to the left %.1f%%"
              % (name, vname, n, 100.0 * n / (H * W), rows.mean(), 100.0 * above.mean(), 100.0 * (above & (delta == 0)).mean(), delta[above].mean() if above.any() else 0,
                 rows.sum() / 1e3, inc.sum() / 1e3, rows.sum() / max(inc.sum(), 1), 100.0 * left.mean()), flush=True)
    print("  (%.1f s)" % (time.time() - t))

# ---- the same restricted to the outliers the pruning (stm_k_irv_rowcount / colprefix / compact) keeps in the list
print("listed outliers only (nmax / S0 > thresh_h):")
for name in ("real", "synthetic"):
    if name == "real": sbs = synth.tiled_sbs_frame(bmp_io.read_bmp(os.path.join(G, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(G, "bud_3.bmp")), H, W)
    else: sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    r = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd, p.thresh_s, p.thresh_h, stop_after_wta=True)
    ol, orr = orc.dr_dcc(r["wta_l"], r["wta_r"])
    L, R = orc.demux_sbs(sbs, W)
    for vname, img, o, disp in (("L", L, ol, r["wta_l"]), ("R", R, orr, r["wta_r"])):
        arms = orc.cross_arms(img, p.ucd, p.lcd, p.usd, p.lsd).astype(np.int32)
        up, down, left, right = arms[0], arms[1], arms[2], arms[3]
        rel = (o == 0).astype(np.int64)
        pre = np.concatenate([np.zeros((H, 1), np.int64), np.cumsum(rel, axis=1)], axis=1)  # pre[y][x] = reliable in [0, x)
        X = np.arange(W)[None, :].repeat(H, 0)
        x0 = np.maximum(X - left, 0); x1 = np.minimum(X + right, W - 1)
        cnt = np.take_along_axis(pre, x1 + 1, 1) - np.take_along_axis(pre, x0, 1)
        vp = np.concatenate([np.zeros((1, W), np.int64), np.cumsum(cnt, axis=0)], axis=0)
        ys, xs = np.nonzero(o)
        top = ys - np.minimum(up[ys, xs], ys); bot = ys + np.minimum(down[ys, xs], H - 1 - ys)
        S0 = vp[bot + 1, xs] - vp[top, xs]
        nmax = np.maximum(64, disp[ys, xs].astype(np.int64) + zd)
        keep = (S0 == 0) | (nmax.astype(np.float32) / np.maximum(S0, 1).astype(np.float32) > np.float32(p.thresh_h))
        listed = np.zeros((H, W), bool); listed[ys[keep], xs[keep]] = True
        ys, xs, top, bot = ys[keep], xs[keep], top[keep], bot[keep]
        rows = bot - top + 1
        ya = np.maximum(ys - 1, 0)
        above = (ys > 0) & listed[ya, xs]
        top_a = ya - np.minimum(up[ya, xs], ya); bot_a = ya + np.minimum(down[ya, xs], H - 1 - ya)
        delta = np.abs(top - top_a) + np.abs(bot - bot_a)
        inc = np.where(above, np.minimum(delta, rows), rows)
        print("%s %s: listed %d of %d outliers, thresh_h %.3f thresh_s %d, mean region rows %.1f; listed outlier directly above %.1f%%; row steps %.0f k now, %.0f k incremental (x%.2f)"
              % (name, vname, keep.sum(), len(keep), p.thresh_h, p.thresh_s, rows.mean(), 100.0 * above.mean(), rows.sum() / 1e3, inc.sum() / 1e3, rows.sum() / max(inc.sum(), 1)), flush=True)
