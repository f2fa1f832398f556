#!/usr/bin/env python
"""Headless counterpart of the reference's video driver (video_io.cpp): same positional parameters, a directory of
side-by-side BMP frames instead of a video file, BMP outputs instead of a window.

usage: stm_video.py <frames dir> <num views> <angle> <out width> <out height> <num disp> <zero disp> <ad coeff>
                    <census coeff> <ucd> <lcd> <usd> <lsd> <thresh_s> <thresh_h> [out dir]
(the 15 arguments of video_io.cpp:49-109; frames are *.bmp, sorted by name)
The angle is truncated to an integer as the reference does (adcensus_stm declares `int angle`, d_io.h:36, and video_io.cpp:158
passes it a float); set STM_EXACT_ANGLE=1 to keep the fractional slant."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv):
    if len(argv) not in (16, 17):
        print(__doc__)
        return -1
    import stm_amd  # noqa: F401
    from stm_amd import device_api as dev, video
    a = argv[1:]
    angle = float(a[2]) if os.environ.get("STM_EXACT_ANGLE") == "1" else float(int(float(a[2])))  # SURVEY A-Q24
    p = dev.FrameParams(num_views=int(a[1]), angle=angle, num_disp=int(a[5]), zero_disp=int(a[6]), ad_coeff=float(a[7]),
                        census_coeff=float(a[8]), ucd=float(a[9]), lcd=float(a[10]), usd=int(a[11]), lsd=int(a[12]),
                        thresh_s=int(a[13]), thresh_h=float(a[14]))
    out_w, out_h = int(a[3]), int(a[4])
    out_dir = a[15] if len(a) > 15 else os.path.join(a[0], "out")
    t0 = time.perf_counter()
    n = 0
    for (k, dl, dr, inter) in video.process_sequence(video.read_bmp_sequence(a[0]), p, out_h, out_w):
        video.write_outputs(out_dir, k, dl, dr, inter)
        n += 1
    dt = time.perf_counter() - t0
    print("%d frames in %.3f s (%.1f frames/s including BMP I/O)" % (n, dt, n / dt if dt > 0 else 0.0))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
