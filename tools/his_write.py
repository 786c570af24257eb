"""Writes synthetic HIS projection files for the tools (the layout paris_amd/host/paris/his.h reads and its save() writes:
a packed little-endian 68-byte header -- u16 0x7000, u16 68, u16 version, u32 file size, u16 image header size, u16 ulx uly brx bry
(1-based, inclusive), u16 frames, u16 correction, f64 integration time, u16 number type, 34 bytes of padding -- then per frame
`image_header` bytes and the pixels, x fastest)."""
import struct

import numpy as np

NUMBER_TYPE = {np.dtype(np.uint8): 2, np.dtype(np.uint16): 4, np.dtype(np.uint32): 32, np.dtype(np.float64): 64, np.dtype(np.float32): 128}


def write_his(path, frames, image_header=0):
    """frames: array (n, rows, cols) of one of the five pixel types"""
    frames = np.ascontiguousarray(frames)
    n, rows, cols = frames.shape
    size = 68 + n * (image_header + rows * cols * frames.dtype.itemsize)
    head = struct.pack("<HHHIHHHHHHHdH", 0x7000, 68, 100, size & 0xFFFFFFFF, image_header, 1, 1, cols, rows, n, 0, 0.0,
                       NUMBER_TYPE[frames.dtype]) + bytes(34)
    assert len(head) == 68
    with open(path, "wb") as f:
        f.write(head)
        for k in range(n):
            if image_header:
                f.write(b"\xab" * image_header)
            f.write(frames[k].astype(frames.dtype.newbyteorder("<"), copy=False).tobytes())
