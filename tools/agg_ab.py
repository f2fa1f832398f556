#!/usr/bin/env python
"""A/B of aggregation kernel variants inside ONE process (interleaved rounds), on the bench workload
(1080p, D=64, stages=1).  Prints avg ms of each named kernel per variant and checks that every variant
produces bit-identical disparity maps.   usage: python tools/agg_ab.py 0 1 2 3 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stm_amd  # noqa: E402
from stm_amd import device_api as dev, synth  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0]
    H, W, D = int(os.environ.get("AB_H", 1080)), int(os.environ.get("AB_W", 1920)), int(os.environ.get("AB_D", 64))
    zd = D // 2
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
    dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
    lib = stm_amd.lib()
    ref = None
    res = {v: {} for v in variants}
    for rnd in range(4):
        for v in variants:
            lib.stm_set_agg_variant(v)
            dev.prof_reset()
            dev.prof_enable(rnd > 0)
            for _ in range(3):
                dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=1)
            torch.cuda.synchronize()
            dev.prof_enable(False)
            got = (dl.cpu().numpy().copy(), dr.cpu().numpy().copy())
            if ref is None:
                ref = got
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), "variant %d changes the result" % v
            if rnd > 0:
                for name in ("agg_h", "agg_v", "agg_hw", "cost_init"):
                    n, ms = dev.prof_read(name)
                    if n:
                        res[v].setdefault(name, []).append(ms / n)
    for v in variants:
        print("variant %3d: " % v + "  ".join("%s med=%.4f min=%.4f" % (k, float(np.median(a)), min(a)) for k, a in res[v].items()))


if __name__ == "__main__":
    main()
