"""pq_v12 with ONE view (the per-stage d_ca_cross: 1920 strips x chunks = one round of waves) against the frame's two views.
   python tools/v12_one_view.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
H, W, D, zd = 1080, 1920, 64, 32
L, R, _ = synth.stereo_pair(H, W, D, zd)
dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
slab = torch.zeros(2, D, H, W, dtype=torch.float32, device='cuda')
scratch = torch.zeros(D, H, W, dtype=torch.float32, device='cuda')
cross = torch.zeros(4, H, W, dtype=torch.uint8, device='cuda')
tl, tr = dev.d_ci_adcensus(dL, dR, slab, 10.0, 30.0, D, zd)
for _ in range(2): dev.d_ca_cross(dL, tl, scratch, cross, 6.0, 20.0, 34, 17, D)
torch.cuda.synchronize()
dev.prof_reset(); dev.prof_enable(True)
for _ in range(5): dev.d_ca_cross(dL, tl, scratch, cross, 6.0, 20.0, 34, 17, D)
torch.cuda.synchronize()
dev.prof_enable(False)
for name in ("pq_h", "pq_vtab", "pq_v12", "pq_hw"):
    n, ms = dev.prof_read(name)
    if n: print("one view  %-8s %3d launches, avg %.4f ms" % (name, n, ms / n), flush=True)
