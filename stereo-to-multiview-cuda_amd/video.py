"""Side-by-side frame sequences (SURVEY.md section 8f rows N1 + N4): the headless analogue of the reference's
video driver (video_io.cpp:42-224).  The OpenCV capture / window is replaced by an iterator of frames (a directory
of BMPs, a list of arrays, ...) and writers for what the viewer would show (interlaced frame, disparity maps
normalised like cv::normalize(..., 0, 1, CV_MINMAX), video_io.cpp:163-164)."""
import ctypes as C
import glob
import os

import numpy as np

from . import bmp_io
from ._lib import f32p, lib, u8p


class FrameStream:
    """Pipelined adcensus_stm over a sequence: submit() frames, collect() results in order (two in flight)."""

    def __init__(self, num_rows, num_cols, params, out_rows=None, out_cols=None):
        self.H, self.W = num_rows, num_cols
        self.Ho, self.Wo = out_rows or num_rows, out_cols or num_cols
        p = params
        self._h = lib().stm_stream_create(num_rows, 2 * num_cols, num_cols, self.Ho, self.Wo, 3, p.num_views, p.angle,
                                          p.num_disp, p.zero_disp, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, p.usd, p.lsd,
                                          p.thresh_s, p.thresh_h)

    def submit(self, sbs):
        sbs = np.ascontiguousarray(sbs, dtype=np.uint8)
        assert sbs.shape == (self.H, 2 * self.W, 3)
        return int(lib().stm_stream_submit(self._h, sbs.ctypes.data_as(u8p)))

    def input_buffer(self):
        """The pinned buffer the next submit() will use, as an (H, 2W, 3) uint8 view (None while that slot is uncollected):
        write the frame into it and call submit_inplace() -- no host copy."""
        p = lib().stm_stream_input_buffer(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, u8p), shape=(self.H, 2 * self.W, 3))

    def submit_inplace(self):
        return int(lib().stm_stream_submit(self._h, None))

    def collect_view(self):
        """Like collect(), but returns views of the stream's pinned result buffers (valid until the frame after the next
        one is submitted) instead of copies."""
        pl, pr, po = f32p(), f32p(), u8p()
        k = int(lib().stm_stream_collect_view(self._h, C.byref(pl), C.byref(pr), C.byref(po)))
        if k < 0:
            return None
        return (k, np.ctypeslib.as_array(pl, shape=(self.H, self.W)), np.ctypeslib.as_array(pr, shape=(self.H, self.W)),
                np.ctypeslib.as_array(po, shape=(self.Ho, self.Wo, 3)))

    def collect(self):
        dl = np.empty((self.H, self.W), np.float32)
        dr = np.empty((self.H, self.W), np.float32)
        out = np.empty((self.Ho, self.Wo, 3), np.uint8)
        k = int(lib().stm_stream_collect(self._h, dl.ctypes.data_as(f32p), dr.ctypes.data_as(f32p), out.ctypes.data_as(u8p)))
        return (k, dl, dr, out) if k >= 0 else None

    def close(self):
        if self._h:
            lib().stm_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def process_sequence(frames, params, out_rows=None, out_cols=None):
    """Generator: yields (index, disp_l, disp_r, interlaced) for every side-by-side frame of `frames`."""
    fs = None
    pending = 0
    for sbs in frames:
        if fs is None:
            fs = FrameStream(sbs.shape[0], sbs.shape[1] // 2, params, out_rows, out_cols)
        if pending == 2:
            yield fs.collect()
            pending -= 1
        fs.submit(sbs)
        pending += 1
    while fs is not None and pending:
        yield fs.collect()
        pending -= 1
    if fs is not None:
        fs.close()


def read_bmp_sequence(directory, pattern="*.bmp"):
    for path in sorted(glob.glob(os.path.join(directory, pattern))):
        yield bmp_io.read_bmp(path)


def normalize_minmax_u8(a):
    """cv::normalize(src, dst, 0, 1, CV_MINMAX) followed by the 8-bit display scaling the viewer applies."""
    lo, hi = float(a.min()), float(a.max())
    if hi <= lo:
        return np.zeros(a.shape, np.uint8)
    return np.clip((a - lo) / (hi - lo) * 255.0, 0, 255).astype(np.uint8)


def write_outputs(out_dir, index, disp_l, disp_r, interlaced):
    os.makedirs(out_dir, exist_ok=True)
    bmp_io.write_bmp(os.path.join(out_dir, "interlaced_%05d.bmp" % index), interlaced)
    bmp_io.write_bmp(os.path.join(out_dir, "disp_l_%05d.bmp" % index), normalize_minmax_u8(disp_l))
    bmp_io.write_bmp(os.path.join(out_dir, "disp_r_%05d.bmp" % index), normalize_minmax_u8(disp_r))
