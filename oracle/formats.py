"""File formats of the reference, restated in Python for the tests (TEST INFRASTRUCTURE ONLY).

HIS reader  : src/his.cpp:105-198   DDBVF writer layout: src/ddbvf.cpp:43-58,72-99,125-153
"""
import struct

import numpy as np

HIS_ID = 0x7000            # src/his.cpp:50
HIS_HEADER_SIZE = 68       # :46
HIS_TYPES = {2: np.uint8, 4: np.uint16, 32: np.uint32, 64: np.float64, 128: np.float32}  # :70-78
_HIS_HEADER = "<HHHIHHHHHHHdH34s"  # :52-68, packed little endian = 68 bytes


def his_header_bytes(n_frames, ulx, uly, brx, bry, number_type, image_header_size=0, file_type=HIS_ID,
                     header_size=HIS_HEADER_SIZE):
    return struct.pack(_HIS_HEADER, file_type, header_size, 100, 0, image_header_size, ulx, uly, brx, bry, n_frames, 0,
                       0.0, number_type, b"\0" * 34)


def his_file_bytes(frames, number_type, image_header_size=0, ulx=1, uly=1, **kw):
    """frames: (n, h, w) array. Builds the bytes the reference's reader expects: the image header precedes EVERY
    frame (src/his.cpp:155-159)."""
    frames = np.asarray(frames)
    n, h, w = frames.shape
    out = [his_header_bytes(n, ulx, uly, ulx + w - 1, uly + h - 1, number_type, image_header_size, **kw)]
    for f in frames:
        out.append(b"\xAB" * image_header_size)
        out.append(np.ascontiguousarray(f.astype(HIS_TYPES.get(number_type, np.float32))).tobytes())
    return b"".join(out)


def his_read(path):
    """src/his.cpp:105-198 -> list of float32 (h, w) arrays; [] for a non-HIS file."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < HIS_HEADER_SIZE:
        return []
    (file_type, header_size, _ver, _fsize, img_hdr, ulx, uly, brx, bry, n_frames, _corr, _t, number_type,
     _rest) = struct.unpack(_HIS_HEADER, raw[:HIS_HEADER_SIZE])
    if file_type != HIS_ID or header_size != HIS_HEADER_SIZE or number_type == 0xFFFF:  # :130-144
        return []
    w, h = brx - ulx + 1, bry - uly + 1                                                 # :146-151
    pos = HIS_HEADER_SIZE
    frames = []
    for _ in range(n_frames):
        pos += img_hdr                                                                    # :155-159
        if number_type not in HIS_TYPES:                                                  # :188-190
            return frames
        dt = np.dtype(HIS_TYPES[number_type])
        a = np.frombuffer(raw, dt, count=w * h, offset=pos).reshape(h, w)
        frames.append(a.astype(np.float32))                                               # :99
        pos += w * h * dt.itemsize
    return frames


def ddbvf_header_bytes(dim_x, dim_y, dim_z):
    """src/ddbvf.cpp:80-92: id, int version, dims, offset 8, 8 zero bytes = 32 bytes."""
    return struct.pack("<IiIIII8s", 0xEFDDDAFA, 0x0010, dim_x, dim_y, dim_z, 8, b"\0" * 8)


def ddbvf_read(path):
    with open(path, "rb") as f:
        head = f.read(32)
        magic, version, dx, dy, dz, off = struct.unpack("<IiIIII", head[:24])
        assert magic == 0xEFDDDAFA and version == 0x10 and off == 8
        data = np.fromfile(f, np.float32)
    return head, data[:dx * dy * dz].reshape(dz, dy, dx)
