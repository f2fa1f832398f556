"""Every oracle stage over the shapes the GPU parity tests use (edge cases, regular cases, the seeded parameter sweep),
without a GPU: meant to run under the sanitizer build (tools/oracle_sanitize.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from conftest import rand_pair  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from test_gpu_parity import CASES, EDGE, _fuzz_cases  # noqa: E402
import stm_amd  # noqa: E402,F401
from stm_amd import synth  # noqa: E402

for (H, W, D, zd, usd, lsd) in EDGE + CASES:
    L, R = rand_pair(max(H, 8), max(W, 8), 101 + H + W)
    L, R = np.ascontiguousarray(L[:H, :W]), np.ascontiguousarray(R[:H, :W])
    cl, cr = orc.ci_adcensus(L, R, 10.0, 30.0, D, zd)
    xl, al = orc.ca_cross(L, cl, 6.0, 20.0, usd, lsd)
    xr, ar = orc.ca_cross(R, cr, 6.0, 20.0, usd, lsd)
    dl, dr = orc.dc_wta(al, zd), orc.dc_wta(ar, zd)
    ol, orr = orc.dr_dcc(dl, dr)
    for flavour in (False, True):
        orc.dr_irv(dl, ol, xl, 1, 0.0, D, zd, usd, 2, device_flavour=flavour)
    orc.filter_bilateral_1(dl, 7, 5.0, 10.0, max(D, 2))
    orc.filter_median(dl)
    orc.filter_gaussian_1(dl, 3, 2.0)
    oc = orc.dibr_occl(dl, dr)
    m = orc.dibr_occl_to_mask(orc.filter_bleed_1(oc[0], 1), orc.filter_bleed_1(oc[1], 2))
    orc.dibr_dbm(L, R, dl, dr, m[0], m[1], 0.3, 2, 1.5)
    orc.dc_hslo(al, L, R, 15.0, 1.0, 3.0, zd)
    print("stages %dx%d D=%d zd=%d ok" % (H, W, D, zd))

for c in _fuzz_cases(28, 20261004):
    H, W, D, zd = c["H"], c["W"], c["D"], c["zd"]
    if c["noise"]:
        L, R = rand_pair(max(H, 8), max(W, 8), c["seed"])
        sbs = np.ascontiguousarray(np.concatenate([L[:H, :W], R[:H, :W]], axis=1))
    else:
        sbs, _ = synth.sbs_frame(H, W, D, min(max(zd, 0), D - 1), seed=c["seed"])
    orc.adcensus_stm(sbs, H, W, c["views"], c["angle"], D, zd, c["ad"], c["cen"], c["ucd"], c["lcd"], c["usd"], c["lsd"],
                     c["ts"], c["th"])
    print("frame %dx%d D=%d ok" % (H, W, D))

sbs, _ = synth.sbs_frame(40, 56, 9, 4)
orc.adcensus_stm_2(sbs, 40, 56, 20, 28, 0.5, 8, 18.43, 9, 4, 10.0, 30.0, 6.0, 20.0, 9, 4, 20, 0.4)
orc.adcensus_stm(sbs, 40, 56, 8, 18.43, 9, 4, 10.0, 30.0, 6.0, 20.0, 9, 4, 20, 0.4, hslo=True)
print("oracle sweep: no sanitizer report")
