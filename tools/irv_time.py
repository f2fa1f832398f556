"""Region voting (all IRV kernels, ProfScope "irv") and the frame on the synthetic and on the real-content 1080p frame;
   prints a checksum of the refined disparity maps (to compare builds quickly; parity proper: tests/).  usage: python tools/irv_time.py [variant ...]"""
import sys, os, time, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth, bmp_io
lib = stm_amd.lib()
H, W, D, zd = 1080, 1920, 64, 32
g = os.path.join(os.getcwd(), "tests", "golden")
frames = {"synthetic": synth.sbs_frame(H, W, D, zd)[0],
          "real": synth.tiled_sbs_frame(bmp_io.read_bmp(os.path.join(g, "bud_2.bmp")), bmp_io.read_bmp(os.path.join(g, "bud_3.bmp")), H, W)}
p = dev.FrameParams(num_disp=D, zero_disp=zd)
for variant in [int(x) for x in sys.argv[1:]] or [0]:
    lib.stm_set_agg_variant(variant)
    for name, sbs in frames.items():
        d_sbs = torch.from_numpy(sbs).cuda()
        dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
        out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
        for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        dev.prof_reset(); dev.prof_enable(True)
        for _ in range(5): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
        torch.cuda.synchronize(); dev.prof_enable(False)
        n, ms = dev.prof_read("irv")
        crc = zlib.crc32(dl.cpu().numpy().tobytes()) ^ zlib.crc32(dr.cpu().numpy().tobytes()) ^ zlib.crc32(out.cpu().numpy().tobytes())
        print("variant %d %-9s frame %.3f ms (%.1f fps)  irv %.4f ms  crc %08x" % (variant, name, dt * 1e3, 1 / dt, ms / max(n, 1), crc), flush=True)
