"""Does replaying the frame's 20 launches as ONE hipGraph shorten the frame?  (torch.cuda.CUDAGraph around stm_d_adcensus_stm, inputs resident)
usage: python tools/graph_frame_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth
H, W, D, zd = 1080, 1920, 64, 32
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
for _ in range(5): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
torch.cuda.synchronize()
ref = (dl.clone(), dr.clone(), out.clone())
def rate(f, n=100):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
t_eager = rate(lambda: dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3))
print("eager  %.4f ms/frame  %.1f frames/s" % (t_eager * 1e3, 1 / t_eager), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
dl.zero_(); dr.zero_(); out.zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay equals eager:", bool((dl == ref[0]).all() and (dr == ref[1]).all() and (out == ref[2]).all()), flush=True)
t_graph = rate(g.replay)
print("graph  %.4f ms/frame  %.1f frames/s" % (t_graph * 1e3, 1 / t_graph), flush=True)
