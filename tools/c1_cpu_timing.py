#!/usr/bin/env python
"""BASELINE config 1 (bud_2 + bud_3, 640x384, D=32, 8 views) on the CPU oracle: frames/s of the frame pipeline and per-stage
milliseconds of the still-image driver's stage chain (image_io.cpp:171-292), with the thread count stated.
usage: python tools/c1_cpu_timing.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stm_amd
from oracle import pyoracle as orc

GOLD = os.path.join(ROOT, "tests", "golden")
g = dict(np.load(os.path.join(GOLD, "bud_c1_golden.npz")))
L, R = stm_amd.bmp_io.read_bmp(os.path.join(GOLD, "bud_2.bmp")), stm_amd.bmp_io.read_bmp(os.path.join(GOLD, "bud_3.bmp"))
D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th, N, angle = [float(x) for x in g["params"]]
D, zd, usd, lsd, ts, N = int(D), int(zd), int(usd), int(lsd), int(ts), int(N)
H, W, _ = L.shape
orc.limit_threads_to_usable_cpus()
stage = {}
def timed(name, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); stage[name] = stage.get(name, 0.0) + (time.perf_counter() - t0) * 1e3; return r
orc.ci_adcensus(L, R, ad, ce, D, zd)  # warm-up (library load, OpenMP pool)
cl, cr = timed("ci_adcensus", orc.ci_adcensus, L, R, ad, ce, D, zd)
xl, al = timed("ca_cross x2", orc.ca_cross, L, cl, ucd, lcd, usd, lsd)
xr, ar = timed("ca_cross x2", orc.ca_cross, R, cr, ucd, lcd, usd, lsd)
dl = timed("dc_wta x2", orc.dc_wta, al, zd); dr = timed("dc_wta x2", orc.dc_wta, ar, zd)
ol, orr = timed("dr_dcc", orc.dr_dcc, dl, dr)
dl, ol = timed("dr_irv x2", orc.dr_irv, dl, ol, xl, ts, th, D, zd, usd, 1, device_flavour=False)
dr, orr = timed("dr_irv x2", orc.dr_irv, dr, orr, xr, ts, th, D, zd, usd, 1, device_flavour=False)
dl = timed("filter_bilateral_1 x2", orc.filter_bilateral_1, dl, 7, 7.0, 7.0, D); dr = timed("filter_bilateral_1 x2", orc.filter_bilateral_1, dr, 7, 7.0, 7.0, D)
occl_l, occl_r = timed("dibr_occl+bleed+mask", orc.dibr_occl, dl, dr)
occl_l = timed("dibr_occl+bleed+mask", orc.filter_bleed_1, occl_l, 1); occl_r = timed("dibr_occl+bleed+mask", orc.filter_bleed_1, occl_r, 1)
ml, mr = timed("dibr_occl+bleed+mask", orc.dibr_occl_to_mask, occl_l, occl_r)
views = [R]
for v in range(1, N - 1):
    shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))
    views.append(timed("dibr_dbm x6", orc.dibr_dbm, L, R, dl, dr, ml, mr, shift, 7, 10.0))
views.append(L)
mux = timed("mux_multiview", orc.mux_multiview, views, angle, H, W, 2)
ok_chain = bool(np.array_equal(mux, g["chain_mux"]))
sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    fr = orc.adcensus_stm(sbs, H, W, N, angle, D, zd, ad, ce, ucd, lcd, usd, lsd, ts, th)
dt = (time.perf_counter() - t0) / reps
ok_frame = bool(np.array_equal(fr["interlaced"], g["frame_mux"]) and np.array_equal(fr["disp_l"], g["frame_disp_l"]))
res = {"config": "BASELINE config 1: img/bud_2.bmp + img/bud_3.bmp, %dx%d, D=%d, zd=%d, usd=%d, %d views, CPU oracle (oracle/stm_oracle.c)" % (W, H, D, zd, usd, N),
       "threads": orc.num_threads(), "frame_pipeline_fps": 1.0 / dt, "frame_pipeline_ms": dt * 1e3,
       "image_io_chain_stage_ms": {k: round(v, 3) for k, v in stage.items()}, "image_io_chain_total_ms": round(sum(stage.values()), 3),
       "parity": {"frame_pipeline_equals_committed_golden": ok_frame, "stage_chain_equals_committed_golden": ok_chain}}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
