"""Generates tests/golden/bud_crop_golden.npz.

Inputs : 64x48 crops of the reference's own test pair img/bud_2.bmp (left) + img/bud_3.bmp (right) and of the
         degenerate identical pair img/fish_1.bmp / img/fish_2.bmp (read in place from /root/reference/img, which
         only exists in the build container -- the crops are DATA, committed so the GPU box can use them).
Outputs: what the CPU oracle (oracle/stm_oracle.c) produces for every stage on those crops.
The reference cannot be built or run here (nvcc / OpenCV / a CUDA GPU are absent), so these vectors pin the
oracle against regressions and the HIP path against the oracle; they are not outputs of the CUDA binary.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stm_amd  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

P = dict(D=8, zd=5, ad=10.0, ce=30.0, ucd=6.0, lcd=20.0, usd=9, lsd=4, ts=20, th=0.4, N=8, angle=18.43)


def main():
    rd = stm_amd.bmp_io.read_bmp
    Lf, Rf = rd("/root/reference/img/bud_2.bmp"), rd("/root/reference/img/bud_3.bmp")
    y0, x0, H, W = 170, 290, 48, 64
    L, R = np.ascontiguousarray(Lf[y0:y0 + H, x0:x0 + W]), np.ascontiguousarray(Rf[y0:y0 + H, x0:x0 + W])
    F = np.ascontiguousarray(rd("/root/reference/img/fish_1.bmp")[100:100 + H, 200:200 + W])
    F2 = np.ascontiguousarray(rd("/root/reference/img/fish_2.bmp")[100:100 + H, 200:200 + W])
    assert np.array_equal(F, F2)
    D, zd = P["D"], P["zd"]
    out = {"L": L, "R": R, "fish": F, "params": np.array([P[k] for k in ("D", "zd", "ad", "ce", "ucd", "lcd", "usd", "lsd", "ts", "th", "N", "angle")], np.float64)}
    out["grey_l"] = orc.grey(L)
    out["census_l"] = orc.census(out["grey_l"])
    cl, cr = orc.ci_adcensus(L, R, P["ad"], P["ce"], D, zd)
    out["cost_l"], out["cost_r"] = cl, cr
    xl, al = orc.ca_cross(L, cl, P["ucd"], P["lcd"], P["usd"], P["lsd"])
    xr, ar = orc.ca_cross(R, cr, P["ucd"], P["lcd"], P["usd"], P["lsd"])
    out["cross_l"], out["cross_r"], out["acost_l"], out["acost_r"] = xl, xr, al, ar
    dl, dr = orc.dc_wta(al, zd), orc.dc_wta(ar, zd)
    out["wta_l"], out["wta_r"] = dl, dr
    ol, orr = orc.dr_dcc(dl, dr)
    out["outl_l"], out["outl_r"] = ol, orr
    il, iol = orc.dr_irv(dl, ol, xl, P["ts"], P["th"], D, zd, P["usd"], 5, True)
    ir, ior = orc.dr_irv(dr, orr, xr, P["ts"], P["th"], D, zd, P["usd"], 5, True)
    out["irv_l"], out["irv_outl_l"], out["irv_r"], out["irv_outl_r"] = il, iol, ir, ior
    bl, br = orc.filter_bilateral_1(il, 7, 5.0, 10.0, D), orc.filter_bilateral_1(ir, 7, 5.0, 10.0, D)
    out["bil_l"], out["bil_r"] = bl, br
    ocl, ocr = orc.dibr_occl(bl, br)
    ocl, ocr = orc.filter_bleed_1(ocl, 1), orc.filter_bleed_1(ocr, 1)
    out["occl_l"], out["occl_r"] = ocl, ocr
    ml, mr = orc.dibr_occl_to_mask(ocl, ocr)
    views = [R]
    for v in range(1, P["N"] - 1):
        shift = np.float32(1.0 - (1.0 * np.float32(v)) / (np.float32(P["N"]) - 1.0))
        views.append(orc.dibr_dbm(L, R, bl, br, ml, mr, float(shift), 10, 15.0))
    views.append(L)
    out["views"] = np.stack(views)
    out["mux"] = orc.mux_multiview(views, P["angle"], H, W, 2)
    out["hslo_l"] = orc.dc_hslo(cl, L, R, 15.0, 1.0, 3.0, zd)
    out["dfm"] = orc.dibr_dfm(L, R, bl, br, 0.5)
    # whole-frame pipeline on the side-by-side frame
    sbs = np.ascontiguousarray(np.concatenate([L, R], axis=1))
    fr = orc.adcensus_stm(sbs, H, W, P["N"], P["angle"], D, zd, P["ad"], P["ce"], P["ucd"], P["lcd"], P["usd"], P["lsd"],
                          P["ts"], P["th"], want_views=False)
    out["frame_disp_l"], out["frame_disp_r"], out["frame_mux"] = fr["disp_l"], fr["disp_r"], fr["interlaced"]
    assert np.array_equal(fr["interlaced"], out["mux"])
    # degenerate identical pair (known-answer: cost at d = zd is exactly 0)
    fcl, _ = orc.ci_adcensus(F, F, P["ad"], P["ce"], D, zd)
    out["fish_cost_l"] = fcl
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bud_crop_golden.npz"), **out)
    print("wrote", {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
