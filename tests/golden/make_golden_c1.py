"""BASELINE config 1 at its real size: the reference's own pair img/bud_2.bmp (left) + img/bud_3.bmp (right), 640x384,
32 disparities (zero_disp 16), 8 views.

  * copies the two BMP files (DATA the reference holds, 0.74 MB each) to tests/golden/ so the GPU box has them;
  * writes tests/golden/bud_c1_golden.npz: what the CPU oracle produces for (a) the device-resident frame pipeline
    (adcensus_stm, d_io.cu:7-238) and (b) the still-image driver's stage chain (image_io.cpp:171-292: IRV x1 host flavour,
    bilateral 7/7/7, host-flavour dibr_dbm with the 7/10 mask gaussian).
The reference cannot be built here (nvcc / OpenCV / a CUDA GPU are absent): these are oracle outputs, not CUDA outputs.

Run from the repo root (build container only):  python tests/golden/make_golden_c1.py
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stm_amd  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

# parameters of SURVEY 8c's understanding check on this pair
P = dict(D=32, zd=16, ad=10.0, ce=30.0, ucd=6.0, lcd=20.0, usd=17, lsd=8, ts=20, th=0.4, N=8, angle=18.43)
GOLD = os.path.join(ROOT, "tests", "golden")


def image_io_chain(o, L, R, p=P):
    """image_io.cpp:171-292 on any object `o` that offers the host-flavour stage API (the oracle or stm_amd.host_api)."""
    D, zd = p["D"], p["zd"]
    cl, cr = o.ci_adcensus(L, R, p["ad"], p["ce"], D, zd)
    xl, al = o.ca_cross(L, cl, p["ucd"], p["lcd"], p["usd"], p["lsd"])
    xr, ar = o.ca_cross(R, cr, p["ucd"], p["lcd"], p["usd"], p["lsd"])
    dl, dr = o.dc_wta(al, zd), o.dc_wta(ar, zd)
    ol, orr = o.dr_dcc(dl, dr)
    return cl, xl, al, dl, dr, ol, orr, xr


def main():
    for n in ("bud_2.bmp", "bud_3.bmp"):
        shutil.copyfile(os.path.join("/root/reference/img", n), os.path.join(GOLD, n))
        os.chmod(os.path.join(GOLD, n), 0o644)
    rd = stm_amd.bmp_io.read_bmp
    L, R = rd(os.path.join(GOLD, "bud_2.bmp")), rd(os.path.join(GOLD, "bud_3.bmp"))
    H, W, _ = L.shape
    assert (H, W) == (384, 640)
    D, zd, N = P["D"], P["zd"], P["N"]
    out = {"params": np.array([P[k] for k in ("D", "zd", "ad", "ce", "ucd", "lcd", "usd", "lsd", "ts", "th", "N", "angle")], np.float64)}
    # (a) frame pipeline
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    fr = orc.adcensus_stm(sbs, H, W, N, P["angle"], D, zd, P["ad"], P["ce"], P["ucd"], P["lcd"], P["usd"], P["lsd"], P["ts"], P["th"])
    out["frame_wta_l"], out["frame_wta_r"] = fr["wta_l"].astype(np.int8), fr["wta_r"].astype(np.int8)  # integers in [-zd, D-1-zd]
    out["frame_disp_l"], out["frame_disp_r"], out["frame_mux"] = fr["disp_l"], fr["disp_r"], fr["interlaced"]
    # (b) still-image chain
    cl, xl, al, dl, dr, ol, orr, xr = image_io_chain(orc, L, R)
    out["chain_cross_l"] = xl
    out["chain_wta_l"] = dl.astype(np.int8)
    out["chain_outl_l"] = ol
    dl, ol = orc.dr_irv(dl, ol, xl, P["ts"], P["th"], D, zd, P["usd"], 1, device_flavour=False)
    dr, orr = orc.dr_irv(dr, orr, xr, P["ts"], P["th"], D, zd, P["usd"], 1, device_flavour=False)
    dl, dr = orc.filter_bilateral_1(dl, 7, 7.0, 7.0, D), orc.filter_bilateral_1(dr, 7, 7.0, 7.0, D)
    out["chain_disp_l"], out["chain_disp_r"] = dl, dr
    occl_l, occl_r = orc.dibr_occl(dl, dr)
    ml, mr = orc.dibr_occl_to_mask(orc.filter_bleed_1(occl_l, 1), orc.filter_bleed_1(occl_r, 1))
    views = [R]
    for v in range(1, N - 1):
        shift = float(np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(N) - 1.0)))
        views.append(orc.dibr_dbm(L, R, dl, dr, ml, mr, shift, 7, 10.0))
    views.append(L)
    out["chain_view_3"] = views[3]
    out["chain_mux"] = orc.mux_multiview(views, P["angle"], H, W, 2)
    # statistics SURVEY 8c's independent check reported for this pair (plateau at offset -6, ~14.5 % L/R outliers)
    vals, cnt = np.unique(fr["wta_l"], return_counts=True)
    out["stats"] = np.array([vals[cnt.argmax()], cnt.max() / float(H * W), (ol > 0).mean()], np.float64)
    np.savez_compressed(os.path.join(GOLD, "bud_c1_golden.npz"), **out)
    print("wrote", {k: (v.shape, str(v.dtype)) for k, v in out.items()}, "stats", out["stats"])


if __name__ == "__main__":
    main()
