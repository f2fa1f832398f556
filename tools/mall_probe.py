"""Does a smaller volume (one that fits the 256 MB memory-side cache) run the aggregation kernels faster per element?
   Per-kernel times of the frame for several (H, D); ns per (pixel x hypothesis)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
lib = stm_amd.lib()
W = 1920
for (H, D) in ((1080, 64), (1080, 32), (1080, 16), (540, 64), (540, 32), (540, 16), (270, 64), (270, 16)):
    zd = D // 2
    sbs, _ = synth.sbs_frame(H, W, D, zd)
    p = dev.FrameParams(num_disp=D, zero_disp=zd)
    d_sbs = torch.from_numpy(sbs).cuda()
    dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
    for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    dev.prof_reset(); dev.prof_enable(True)
    for _ in range(5): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    torch.cuda.synchronize(); dev.prof_enable(False)
    vol_mb = 2 * D * H * W * 4 / 1e6
    s = "H %4d D %2d volume(both views) %6.0f MB:" % (H, D, vol_mb)
    for name in ("pq_h", "pq_v12", "pq_hw"):
        n, ms = dev.prof_read(name)
        ms /= max(n, 1)
        s += "  %s %.4f ms (%.3f ps/elem)" % (name, ms, ms * 1e9 / (2 * D * H * W))
    print(s, flush=True)
