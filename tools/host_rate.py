"""PCIe-inclusive rates of the host-buffer boundary at the bench workload (1920x1080, D=64, 8 views):
  (a) stm_adcensus_stm  - one blocking call per frame: upload, compute, download (the reference's adcensus_stm contract)
  (b) stm_stream_*      - the same frames through the double-buffered stream (upload k+1 || compute k || download k-1)
  (c) the same stream through stm_stream_input_buffer / stm_stream_collect_view (no host-side copies)
bench.py's `value` excludes the transfers; these numbers are what DESIGN.md section 5 quotes beside it.
Usage (GPU box):  python tools/host_rate.py [frames [height width disp]]      (STM_STREAM_GRAPH=0: no hipGraph replay)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import stm_amd
from stm_amd import device_api as dev, host_api, synth, video

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H, W, D = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1080, 1920, 64)
zd = D // 2
sbs, _ = synth.sbs_frame(H, W, D, zd)
p = dev.FrameParams(num_disp=D, zero_disp=zd)


def per_call():
    return host_api.adcensus_stm(sbs, W, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd,
                                 p.usd, p.lsd, p.thresh_s, p.thresh_h)


ref = per_call()
per_call()
t = time.perf_counter()
for _ in range(n):
    per_call()
dt_call = (time.perf_counter() - t) / n

fs = video.FrameStream(H, W, p)
for _ in range(2):
    fs.submit(sbs)
fs.collect()
fs.collect()
t = time.perf_counter()
pending = 0
last = None
for _ in range(n):
    if pending == 2:
        last = fs.collect()
        pending -= 1
    fs.submit(sbs)
    pending += 1
while pending:
    last = fs.collect()
    pending -= 1
dt_stream = (time.perf_counter() - t) / n
fs.close()
assert np.array_equal(last[3], ref[2]) and np.array_equal(last[1], ref[0])

# (c) the same stream without the two host-side copies: frames written into the pinned input buffer, results read in place
fs = video.FrameStream(H, W, p)
for _ in range(2):
    fs.input_buffer()[...] = sbs
    fs.submit_inplace()
fs.collect_view()
fs.collect_view()
src = fs.input_buffer()
t = time.perf_counter()
pending = 0
for _ in range(n):
    if pending == 2:
        last = fs.collect_view()
        pending -= 1
    # a decoder would write the frame here; the synthetic frame is already in both pinned buffers
    fs.submit_inplace()
    pending += 1
while pending:
    last = fs.collect_view()
    pending -= 1
dt_zc = (time.perf_counter() - t) / n
assert np.array_equal(last[3], ref[2]) and np.array_equal(last[1], ref[0])
fs.close()
print(json.dumps({"frames": n, "size": [H, W, D], "graph": os.environ.get("STM_STREAM_GRAPH", "1") != "0", "host_call_ms": round(dt_call * 1e3, 3), "host_call_fps": round(1 / dt_call, 1),
                  "stream_ms": round(dt_stream * 1e3, 3), "stream_fps": round(1 / dt_stream, 1),
                  "stream_zero_copy_ms": round(dt_zc * 1e3, 3), "stream_zero_copy_fps": round(1 / dt_zc, 1),
                  "bytes_in": int(sbs.nbytes), "bytes_out": int(2 * H * W * 4 + H * W * 3)}))
