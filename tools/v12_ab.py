"""Parity of the matrix-pipe aggregation (small ragged frames vs the oracle) + per-kernel times at 1080p for a list of agg variants.
   python tools/v12_ab.py [variant ...]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth
from oracle import pyoracle as orc
lib = stm_amd.lib()
variants = [int(x) for x in sys.argv[1:]] or [0]

def run(H, W, D, zd, usd, lsd, stages, variant):
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=usd, lsd=lsd)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
    lib.stm_set_agg_variant(variant)
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=stages)
    torch.cuda.synchronize()
    return sbs, p, dl.cpu().numpy(), dr.cpu().numpy(), out.cpu().numpy()

cases = [(48, 64, 16, 8, 9, 4), (33, 70, 5, 2, 5, 2), (96, 160, 16, 8, 17, 8), (90, 200, 24, 10, 34, 17), (64, 131, 64, 32, 34, 17),
         (150, 330, 70, 30, 20, 10), (40, 37, 130, 64, 34, 17), (17, 19, 3, 1, 3, 1), (200, 260, 32, 16, 17, 8), (300, 100, 16, 8, 60, 30),
         (130, 90, 16, 8, 1, 1), (260, 64, 32, 16, 100, 50)]
bad = 0
for v in variants:
    for (H, W, D, zd, usd, lsd) in cases:
        sbs, p, dl, dr, _ = run(H, W, D, zd, usd, lsd, 1, v)
        want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd,
                                p.thresh_s, p.thresh_h, stop_after_wta=True)
        ml = int((dl != want["wta_l"]).sum()); mr = int((dr != want["wta_r"]).sum())
        print("variant %d case %dx%d D=%d usd=%d: mismatches L %d R %d of %d" % (v, H, W, D, usd, ml, mr, H * W), flush=True)
        bad += ml + mr
print("TOTAL MISMATCH", bad, flush=True)
H, W, D, zd = 1080, 1920, 64, 32
ref = None
for v in variants:
    sbs, p, dl, dr, o = run(H, W, D, zd, 34, 17, 3, v)
    if ref is None: ref = (dl, dr, o)
    else: print("1080p variant %d vs %d: disp_l %d disp_r %d interlaced %d" % (v, variants[0], int((dl != ref[0]).sum()), int((dr != ref[1]).sum()), int((o != ref[2]).sum())), flush=True)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl_ = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr_ = torch.zeros_like(dl_); out_ = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
    for _ in range(3): dev.d_adcensus_stm(d_sbs, dl_, dr_, out_, p, stages=3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): dev.d_adcensus_stm(d_sbs, dl_, dr_, out_, p, stages=3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
    print("variant", v, "ms/frame %.3f fps %.1f" % (dt * 1e3, 1 / dt), flush=True)
    dev.prof_reset(); dev.prof_enable(True)
    for _ in range(5): dev.d_adcensus_stm(d_sbs, dl_, dr_, out_, p, stages=3)
    torch.cuda.synchronize()
    dev.prof_enable(False)
    for name in ("pq_cost", "pq_h", "pq_vtab", "pq_v12", "pq_htab", "pq_hw", "cross_arms", "irv", "bilateral"):
        n, ms = dev.prof_read(name)
        if n: print("   %-12s %3d launches, avg %.4f ms" % (name, n, ms / n), flush=True)
sys.exit(1 if bad else 0)
