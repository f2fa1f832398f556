"""Three 1080p frames of the default pipeline (for rocprofv3 counter runs).  usage: python tools/frame_once.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
H, W, D, zd = 1080, 1920, 64, 32
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
torch.cuda.synchronize()
