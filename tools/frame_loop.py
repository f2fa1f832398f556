"""N full 1080p frames through stm_d_adcensus_stm (for rocprofv3).  usage: python tools/frame_loop.py [frames] [variant] [stages]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, W, D, zd = 1080, 1920, 64, 32
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
stm_amd.lib().stm_set_agg_variant(variant)
for _ in range(n): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=stages)
torch.cuda.synchronize()
print("done", n)
