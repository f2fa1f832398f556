"""Randomised frame-pipeline parity sweep on the GPU: random small shapes and parameters, stages 3 with and without HSLO,
every output compared with the CPU oracle.  usage: python tools/random_parity_sweep.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import conftest  # noqa: F401  (path setup, oracle build)
from stm_amd import device_api as dev, synth
from oracle import pyoracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for case in range(n):
    H = int(rng.randint(3, 90)); W = int(rng.randint(3, int(os.environ.get("WMAX", "400")))); D = int(rng.choice([3, 8, 16, 17, 31, 64, 65, 100]))
    zd = int(rng.randint(0, D)); usd = int(rng.choice([1, 5, 17, 34, 40, 63])); lsd = int(rng.randint(1, usd + 1))
    hslo = bool(rng.randint(0, 2))
    sbs, _ = synth.sbs_frame(H, W, D, zd, seed=1000 + case)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=usd, lsd=lsd)
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda"); dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3 | (0x100 if hslo else 0))
    torch.cuda.synchronize()
    want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, usd, lsd,
                            p.thresh_s, p.thresh_h, hslo=hslo)
    ok = (np.array_equal(dl.cpu().numpy(), want["disp_l"]) and np.array_equal(dr.cpu().numpy(), want["disp_r"])
          and np.array_equal(out.cpu().numpy(), want["interlaced"]))
    print("case %2d H=%3d W=%3d D=%3d zd=%3d usd=%2d lsd=%2d hslo=%d : %s" % (case, H, W, D, zd, usd, lsd, hslo, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
