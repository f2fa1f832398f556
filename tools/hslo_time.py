"""Per-kernel times of the 1080p frame with HSLO for a list of agg variants.  usage: python tools/hslo_time.py v1 v2 ..."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
lib = stm_amd.lib()
H, W, D, zd = 1080, int(os.environ.get("W", "1920")), int(os.environ.get("D", "64")), int(os.environ.get("ZD", "32"))
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
d_sbs = torch.from_numpy(sbs).cuda()
dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
for variant in [int(x) for x in sys.argv[1:]]:
    lib.stm_set_agg_variant(variant)
    for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=259)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=259)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    import zlib
    crc = zlib.crc32(dl.cpu().numpy().tobytes()) ^ zlib.crc32(dr.cpu().numpy().tobytes()) ^ zlib.crc32(out.cpu().numpy().tobytes())
    print("variant", variant, "ms/frame %.3f fps %.1f  crc %08x" % (dt * 1e3, 1 / dt, crc), flush=True)
    dev.prof_reset(); dev.prof_enable(True)
    for _ in range(5): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=259)
    torch.cuda.synchronize()
    dev.prof_enable(False)
    for name in ("pq_hw", "hslo_classes", "hslo_lr", "hslo_rl", "hslo_tb", "hslo_bt"):
        n, ms = dev.prof_read(name)
        if n: print("   %-12s %3d launches, avg %.4f ms" % (name, n, ms / n), flush=True)
