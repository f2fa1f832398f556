"""Throughput of the frame pipeline when two host threads (own stream + own workspace each) feed the GPU concurrently,
vs one thread: how much the latency-bound tail of one frame overlaps the aggregation of another."""
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import torch, stm_amd
from stm_amd import device_api as dev, synth
H, W, D, zd = 1080, 1920, 64, 32
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)
N = 60
def worker(nframes, res, idx):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        d_sbs = torch.from_numpy(sbs).cuda()
        dl = torch.zeros(H, W, dtype=torch.float32, device='cuda'); dr = torch.zeros_like(dl)
        out = torch.zeros(H, W, 3, dtype=torch.uint8, device='cuda')
        for _ in range(3): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
        stream.synchronize()
        res[idx] = "ready"
        barrier.wait()
        for _ in range(nframes): dev.d_adcensus_stm(d_sbs, dl, dr, out, p, stages=3)
        stream.synchronize()
for nthreads in (1, 2, 3):
    barrier = threading.Barrier(nthreads + 1)
    res = [None] * nthreads
    ths = [threading.Thread(target=worker, args=(N, res, i)) for i in range(nthreads)]
    for t in ths: t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.1f frames/s (%.3f ms per frame)" % (nthreads, nthreads * N / dt, dt / (nthreads * N) * 1e3), flush=True)
