"""24-bit BMP reader / writer (BITMAPINFOHEADER, BI_RGB, bottom-up or top-down).

The reference reads its img/ pairs with cv::imread (image_io.cpp:95-96), which hands
the kernels interleaved BGR u8, row-major, no row padding.
This module produces exactly that layout without OpenCV.  Trailing bytes after the
pixel array (bud_1.bmp / bud_5.bmp carry two) are tolerated.
The C++ twin lives in csrc/stm_bmp.cpp (stm_bmp_read / stm_bmp_write).
"""
import struct

import numpy as np


def read_bmp(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] != b"BM":
        raise ValueError("%s: not a BMP file" % path)
    data_off = struct.unpack_from("<I", buf, 10)[0]
    hdr_sz, width, height, planes, bpp, comp = struct.unpack_from("<IiiHHI", buf, 14)
    if hdr_sz < 40 or bpp != 24 or comp != 0:
        raise ValueError("%s: only uncompressed 24-bit BMP is supported (hdr=%d bpp=%d comp=%d)" % (path, hdr_sz, bpp, comp))
    top_down = height < 0
    H, W = abs(height), width
    stride = (W * 3 + 3) & ~3
    need = data_off + stride * H
    if len(buf) < need:
        raise ValueError("%s: truncated pixel array" % path)
    rows = np.frombuffer(buf, dtype=np.uint8, count=stride * H, offset=data_off).reshape(H, stride)
    img = rows[:, : W * 3].reshape(H, W, 3)
    if not top_down:
        img = img[::-1]
    return np.ascontiguousarray(img)


def write_bmp(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    H, W, E = img.shape
    if E != 3:
        raise ValueError("write_bmp wants [H][W][3] BGR")
    stride = (W * 3 + 3) & ~3
    rows = np.zeros((H, stride), np.uint8)
    rows[:, : W * 3] = img[::-1].reshape(H, W * 3)
    hdr = struct.pack("<2sIHHI", b"BM", 54 + stride * H, 0, 0, 54)
    info = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, stride * H, 2835, 2835, 0, 0)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(info)
        f.write(rows.tobytes())
