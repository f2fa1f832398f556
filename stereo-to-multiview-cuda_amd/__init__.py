"""MI355X-native stereo -> multiview hot path (AD-Census cost volume, cross-based aggregation, WTA / HSLO,
DCC + IRV + bilateral refinement, DIBR view synthesis, multiview interlacing).

The compute lives in libstm_hip.so (hand-written HIP for gfx950, C ABI in include/stm_hip.h).  This package is
the host-side mirror of the reference's per-stage API:
  host_api    numpy, host flavour  (image_io.cpp call sites)
  device_api  torch device tensors, device flavour (adcensus_stm call sites)
  bmp_io      the reference's img/*.bmp format without OpenCV
  synth       seeded synthetic stereo pairs
  video       pipelined side-by-side frame sequences + writers (headless video_io.cpp)
  sharding    frame-batch sharding across GPUs (one process per GPU, torch.distributed)
"""
from . import bmp_io  # noqa: F401
from ._lib import LIB_PATH, lib  # noqa: F401
from .build import build  # noqa: F401
