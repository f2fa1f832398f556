"""Timing experiments on stm_k_pq_hsr (timing library: STM_LIB=timing).  usage: STM_LIB=timing python tools/hsr_time.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
lib = stm_amd.lib()
H, W, D, zd = 1080, 1920, 64, 32
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
def t(knob, parts):
    if parts: os.environ["STM_HSR_PARTS"] = str(parts)
    else: os.environ.pop("STM_HSR_PARTS", None)
    lib.stm_set_agg_variant(knob * 100000)
    for _ in range(2): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    dev.prof_reset(); dev.prof_enable(True)
    for _ in range(5): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
    torch.cuda.synchronize(); dev.prof_enable(False)
    r = []
    for name in ("pq_htab", "pq_hw"):
        n, ms = dev.prof_read(name)
        r.append(ms / max(n, 1))
    print("knob %d parts %s: pq_htab %.4f pq_hw %.4f ms" % (knob, parts or "auto", r[0], r[1]), flush=True)
for parts in (0, 2, 4, 6, 8, 10, 12, 15, 20, 24): t(0, parts)
for knob in (1, 2, 3, 4, 5, 6, 7, 8): t(knob, 0)
