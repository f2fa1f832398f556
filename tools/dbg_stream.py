"""debugging aid: the two frame-stream tests back to back, with switches (argv[1] = mode)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import conftest  # noqa
from stm_amd import device_api as dev, host_api, synth, video

mode = sys.argv[1] if len(sys.argv) > 1 else "a"

def t1():
    H, W, D, zd = 64, 96, 8, 4
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + k)[0] for k in range(5)]
    got = list(video.process_sequence(iter(frames), p))
    if "h" in mode:
        for k, f in enumerate(frames):
            host_api.adcensus_stm(f, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd, p.thresh_s, p.thresh_h)

def t2():
    H, W, D, zd = 48, 80, 8, 4
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=9, lsd=4)
    frames = [synth.sbs_frame(H, W, D, zd, seed=synth.SEED + 100 + k)[0] for k in range(9)]
    fs = video.FrameStream(H, W, p)
    got = []
    rng = np.random.default_rng(77)
    big_l = rng.integers(0, 256, (300, 500, 3), dtype=np.uint8); big_r = rng.integers(0, 256, (300, 500, 3), dtype=np.uint8)
    for k, f in enumerate(frames):
        if k >= 2:
            got.append(fs.collect())
        fs.submit(f)
        if k in (3, 6) and "b" in mode:
            host_api.ci_adcensus(big_l, big_r, 10.0, 30.0, 24, 12)
    got.append(fs.collect()); got.append(fs.collect())
    fs.close()
    print("t2 ok", [g[0] for g in got])

if "1" in mode: t1()
t2()
print("done", mode)
