"""Per-stage device API at 1920x1080, D=64: d_ci_adcensus -> d_ca_cross -> d_dc_wta (what d_io.cu:103-132 / image_io.cpp:209-223
call), matrix-pipe kernels (default) vs the vector-ALU kernels (variant 10000); results compared with each other.
   python tools/stage_time.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth
lib = stm_amd.lib()
H, W, D, zd = 1080, 1920, 64, 32
L, R, _ = synth.stereo_pair(H, W, D, zd)
dL, dR = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
res = {}
for variant in (0, 10000):
    lib.stm_set_agg_variant(variant)
    slab = torch.zeros(2, D, H, W, dtype=torch.float32, device='cuda')
    scratch = torch.zeros(D, H, W, dtype=torch.float32, device='cuda')
    cross = torch.zeros(4, H, W, dtype=torch.uint8, device='cuda')
    disp = torch.zeros(H, W, dtype=torch.float32, device='cuda')
    def chain():
        tl, tr = dev.d_ci_adcensus(dL, dR, slab, 10.0, 30.0, D, zd)
        t0 = time.perf_counter()
        dev.d_ca_cross(dL, tl, scratch, cross, 6.0, 20.0, 34, 17, D)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dev.d_dc_wta(tl, disp, D, zd)
        torch.cuda.synchronize()
        return t1 - t0
    for _ in range(2): chain()
    ts = [chain() for _ in range(10)]
    res[variant] = (disp.cpu().numpy().copy(), slab[0].cpu().numpy().copy())
    print("variant %5d: d_ca_cross (one view, arms + H V V H, result in the input planes) %.3f ms (min of 10, incl. the host sync)" % (variant, min(ts) * 1e3), flush=True)
print("disparities equal:", bool(np.array_equal(res[0][0], res[10000][0])), " aggregated volumes equal:", bool(np.array_equal(res[0][1], res[10000][1])))
