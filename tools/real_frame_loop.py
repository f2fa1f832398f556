"""N full 1080p frames of REAL content (tests/golden bud pair tiled, bench.py's real_content leg) through stm_d_adcensus_stm, for rocprofv3.
usage: python tools/real_frame_loop.py [frames]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth, bmp_io
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, D, zd = 1080, 1920, 64, 32
g = os.path.join(os.getcwd(), "tests", "golden")
sbs = synth.tiled_sbs_frame(bmp_io.read_bmp(os.path.join(g, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(g, "bud_3.bmp")), H, W)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
for _ in range(n): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
torch.cuda.synchronize()
from stm_amd import host_api
L = np.ascontiguousarray(sbs[:, :W])
cross, _ = host_api.ca_cross(L, np.zeros((1, H, W), np.float32), p.ucd, p.lcd, p.usd, p.lsd)
print("mean arms U D L R", [float(c.mean()) for c in cross])
dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=1); torch.cuda.synchronize()
wl, wr = dl.cpu().numpy(), dr.cpu().numpy()
ol, orr = host_api.dr_dcc(wl, wr)
print("outliers L %.3f R %.3f" % ((ol != 0).mean(), (orr != 0).mean()))
print("done", n)
