"""HIS projection reader/writer, DDBVF sink, directory source and angle files of the C++ host layer
(paris_amd/host/paris/{his,ddbvf,source,sink}.h through paris_amd/lib/libparis_io.so), checked against the format
restatement in oracle/formats.py. Pure host code: runs without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import formats as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


@pytest.fixture(scope="module")
def io():
    lib = C.CDLL(os.environ.get("PARIS_IO_LIB") or os.path.join(ROOT, "paris_amd", "lib", "libparis_io.so"))  # `make sanitize`
    lib.paris_io_his_load.argtypes = [C.c_char_p, _u32p, _u32p, _u32p, C.POINTER(_fp)]
    lib.paris_io_free.argtypes = [C.c_void_p]
    lib.paris_io_his_save.argtypes = [C.c_char_p, _fp, C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint16]
    lib.paris_io_ddbvf_write.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _fp, C.c_uint32, C.c_uint32]
    lib.paris_io_read_angles.argtypes = [C.c_char_p, _u32p, C.POINTER(_fp)]
    lib.paris_io_source_scan.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint16, C.c_uint32, _u32p, _u32p, _fp, _fp, _u32p]
    return lib


def his_load(io, path):
    n, w, h, data = C.c_uint32(), C.c_uint32(), C.c_uint32(), _fp()
    rc = io.paris_io_his_load(str(path).encode(), C.byref(n), C.byref(w), C.byref(h), C.byref(data))
    if rc:
        raise OSError("cannot open")
    if n.value == 0:
        return []
    a = np.ctypeslib.as_array(data, shape=(n.value, h.value, w.value)).copy()
    io.paris_io_free(data)
    return list(a)


@pytest.mark.parametrize("number_type", [2, 4, 32, 64, 128])
@pytest.mark.parametrize("image_header", [0, 32])
def test_his_reader_all_number_types(io, tmp_path, number_type, image_header):
    rng = np.random.default_rng(number_type)
    hi = {2: 255, 4: 65535, 32: 2 ** 32 - 1, 64: 1e6, 128: 1e6}[number_type]
    frames = (rng.random((3, 5, 7)) * hi).astype(F.HIS_TYPES[number_type])
    p = tmp_path / "a.his"
    p.write_bytes(F.his_file_bytes(frames, number_type, image_header, ulx=3, uly=9))  # rectangle need not start at 1
    got = his_load(io, p)
    want = F.his_read(p)
    assert len(got) == 3 == len(want)
    for g, w_, f in zip(got, want, frames):
        assert np.array_equal(g, w_) and np.array_equal(g, f.astype(np.float32))


def test_his_reader_rejects_what_the_reference_rejects(io, tmp_path):
    good = np.ones((1, 2, 2), np.float32)
    cases = {"wrong_id.his": F.his_file_bytes(good, 128, file_type=0x7001),
             "wrong_header_size.his": F.his_file_bytes(good, 128, header_size=100),
             "type_not_implemented.his": F.his_file_bytes(good, 0xFFFF),
             "unknown_type.his": F.his_file_bytes(good, 8),
             "empty.his": b"", "text.txt": b"hello"}
    for name, raw in cases.items():
        p = tmp_path / name
        p.write_bytes(raw)
        assert his_load(io, p) == [] == F.his_read(p), name
    with pytest.raises(OSError):
        his_load(io, tmp_path / "missing.his")


def test_his_writer_round_trip(io, tmp_path):
    rng = np.random.default_rng(0)
    frames = rng.random((4, 6, 10), dtype=np.float32)
    for number_type, conv in ((128, np.float32), (4, np.uint16), (64, np.float64)):
        src = (frames * 1000).astype(np.float32)
        p = tmp_path / ("w%d.his" % number_type)
        assert io.paris_io_his_save(str(p).encode(), src.ctypes.data_as(_fp), 4, 10, 6, number_type, 32) == 0
        want = [f.astype(conv).astype(np.float32) for f in src]
        for got, a, b in zip(F.his_read(p), his_load(io, p), want):  # oracle reader and own reader agree with the input
            assert np.array_equal(got, a) and np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.float64, np.float32])
def test_tools_his_writer_is_read_back(io, tmp_path, dtype):
    """tools/his_write.py (the data sets of tools/e2e_bench.py and tools/shared_source_bench.py are written with it, not with the
    oracle): what it writes is what the host reader loads, and -- the unused file-size field aside -- the oracle's encoding."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from his_write import NUMBER_TYPE, write_his
    fr = np.random.default_rng(3).integers(0, 250, size=(4, 6, 11)).astype(dtype)
    path = tmp_path / "t.his"
    write_his(str(path), fr, 24)
    got = his_load(io, path)
    assert len(got) == 4 and all(np.array_equal(g, f.astype(np.float32)) for g, f in zip(got, fr))
    a, b = bytearray(path.read_bytes()), bytearray(F.his_file_bytes(fr, NUMBER_TYPE[np.dtype(dtype)], 24))
    a[6:10] = b[6:10] = b"\0\0\0\0"
    assert a == b


def test_ddbvf_layout_and_slab_offsets(io, tmp_path):
    dx, dy, dz = 5, 3, 7
    vol = np.arange(dx * dy * dz, dtype=np.float32).reshape(dz, dy, dx)
    base = str(tmp_path / "vol")
    # three slabs written out of order at their own first slice (the reference writes all at 0: SURVEY Q4)
    for create, (z0, z1) in ((1, (4, 7)), (0, (0, 2)), (0, (2, 4))):
        slab = np.ascontiguousarray(vol[z0:z1])
        assert io.paris_io_ddbvf_write(base.encode(), create, dx, dy, dz, slab.ctypes.data_as(_fp), z1 - z0, z0) == 0
    head, data = F.ddbvf_read(base + ".ddbvf")
    assert head == F.ddbvf_header_bytes(dx, dy, dz)  # byte-exact 32-byte header (int version: Q13)
    assert np.array_equal(data, vol)
    assert os.path.getsize(base + ".ddbvf") == 32 + vol.nbytes
    # reference error behaviour: start out of bounds / wrong dims -> runtime_error (src/ddbvf.cpp:131-135)
    assert io.paris_io_ddbvf_write(base.encode(), 0, dx, dy, dz, vol.ctypes.data_as(_fp), 1, dz) == 2
    assert io.paris_io_ddbvf_write(base.encode(), 0, dx, dy, dz, vol.ctypes.data_as(_fp), dz + 1, 0) == 2


def read_angles(io, path):
    n, data = C.c_uint32(), _fp()
    assert io.paris_io_read_angles(str(path).encode(), C.byref(n), C.byref(data)) == 0
    a = [data[i] for i in range(n.value)]
    io.paris_io_free(data)
    return a


def test_angle_file_quirks(io, tmp_path):
    p = tmp_path / "angles.txt"
    p.write_text("0.5\n1.5\n2.25")            # ends right after the last digit: exactly three values
    assert read_angles(io, p) == [0.5, 1.5, 2.25]
    p.write_text("0.5\n1.5\n2.25\n")          # trailing newline: the reference's loop appends one 0 (Q15)
    assert read_angles(io, p) == [0.5, 1.5, 2.25, 0.0]
    p.write_text("0,5 1,5\n2,25\n")           # decimal comma (de_DE) when the first line has a ','
    assert read_angles(io, p) == [0.5, 1.5, 2.25, 0.0]
    assert read_angles(io, tmp_path / "missing.txt") == []  # unopenable: defaults are used (src/source.cpp:44-48)


def test_source_order_stride_and_skips(io, tmp_path):
    d = tmp_path / "proj"
    d.mkdir()
    # three files, 2 + 3 + 1 frames; first pixel encodes the global frame number; names force the sort order
    k = 0
    for name, n in (("b_002.his", 3), ("a_001.his", 2), ("c_003.his", 1)):
        pass
    for name, n in (("a_001.his", 2), ("b_002.his", 3), ("c_003.his", 1)):
        fr = np.zeros((n, 2, 2), np.float32)
        for i in range(n):
            fr[i, 0, 0] = k
            k += 1
        (d / name).write_bytes(F.his_file_bytes(fr, 128, 32))
    (d / "a_000_notes.txt").write_text("not a projection")  # sorted first, skipped with a warning
    ang = tmp_path / "ang.txt"
    ang.write_text(" ".join(str(10.0 * i) for i in range(6)))

    def scan(quality, angles):
        cap = 16
        n, skipped = C.c_uint32(), C.c_uint32()
        idx = (C.c_uint32 * cap)()
        phi = (C.c_float * cap)()
        first = (C.c_float * cap)()
        rc = io.paris_io_source_scan(str(d).encode(), int(angles), str(ang).encode(), quality, cap, C.byref(n), idx, phi, first,
                                     C.byref(skipped))
        assert rc == 0
        return list(idx[:n.value]), list(phi[:n.value]), list(first[:n.value]), skipped.value

    idx, phi, first, skipped = scan(1, False)
    assert idx == [0, 1, 2, 3, 4, 5] and first == [0, 1, 2, 3, 4, 5] and skipped == 1 and phi == [0.0] * 6
    idx, phi, first, skipped = scan(2, True)   # quality stride keeps the original index (src/source.cpp:105-113)
    assert idx == [0, 2, 4] and first == [0, 2, 4] and phi == [0.0, 20.0, 40.0]
    idx, _, _, _ = scan(4, False)
    assert idx == [0, 4]


@pytest.mark.parametrize("number_type", [2, 4, 32, 64, 128])
@pytest.mark.parametrize("image_header", [0, 32])
def test_frame_stream_equals_the_queueing_source(io, tmp_path, number_type, image_header):
    """frame_stream (one frame at a time into caller memory, optional row band, stride frames seeked over) hands out
    exactly the frames, indices, angles and skipped files of the reference-shaped source, and touches only the band."""
    io.paris_io_stream_scan.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32, _u32p, _u32p, _fp, _fp, _u32p]
    rng = np.random.default_rng(number_type + image_header)
    d = tmp_path / "proj"
    d.mkdir()
    w, h = 7, 10
    frames = []
    for name, n in (("a.his", 3), ("b.his", 1), ("c.his", 4)):
        fr = rng.integers(0, 250, size=(n, h, w)).astype(np.float32)
        frames += list(fr)
        (d / name).write_bytes(F.his_file_bytes(fr, number_type, image_header))
    (d / "b_broken.his").write_bytes(b"\x00" * 100)            # wrong id: skipped
    truncated = F.his_file_bytes(rng.integers(0, 250, size=(2, h, w)).astype(np.float32), number_type, image_header)
    cut = len(truncated) - (w * h * {2: 1, 4: 2, 32: 4, 64: 8, 128: 4}[number_type]) // 2 - 3
    (d / "d_truncated.his").write_bytes(truncated[:cut])       # second frame half there: zeros for the missing pixels
    want_tail = his_load(io, d / "d_truncated.his")
    assert len(want_tail) == 2 and np.any(want_tail[1] != 0) and want_tail[1][-1, -1] == 0
    frames += want_tail
    ang = tmp_path / "ang.txt"
    ang.write_text(" ".join(str(3.0 * i) for i in range(len(frames))))

    for quality in (1, 2, 3):
        for first, count in ((0, h), (2, 5), (9, 1), (4, 0)):
            cap = 16
            n, skipped = C.c_uint32(), C.c_uint32()
            idx = (C.c_uint32 * cap)()
            phi = (C.c_float * cap)()
            data = np.full((cap, h, w), -7.0, np.float32)
            rc = io.paris_io_stream_scan(str(d).encode(), 1, str(ang).encode(), quality, w, h, first, count, cap, C.byref(n), idx,
                                         phi, data.ctypes.data_as(_fp), C.byref(skipped))
            assert rc == 0 and skipped.value == 1
            keep = list(range(0, len(frames), quality))
            assert list(idx[:n.value]) == keep
            assert list(phi[:n.value]) == [np.float32(3.0 * i) for i in keep]
            for j, i in enumerate(keep):
                assert np.array_equal(data[j, first:first + count], frames[i][first:first + count])
                assert np.all(data[j, :first] == -7.0) and np.all(data[j, first + count:] == -7.0)
    # a frame of another size is reported, not written
    rc = io.paris_io_stream_scan(str(d).encode(), 0, None, 1, w + 1, h, 0, h, 0, C.byref(n), idx, phi, data.ctypes.data_as(_fp),
                                 C.byref(skipped))
    assert rc == 3


@pytest.mark.parametrize("seed", range(12))
def test_his_reader_random_files(io, tmp_path, seed):
    """Seeded random HIS files: every number type, rectangle origins other than (1, 1), image headers of any size,
    1..5 frames of random size. The C++ reader == the Python restatement of src/his.cpp:105-198, value for value."""
    rng = np.random.default_rng(900 + seed)
    number_type = int(rng.choice([2, 4, 32, 64, 128]))
    w, h, n = int(rng.integers(1, 70)), int(rng.integers(1, 50)), int(rng.integers(1, 6))
    ulx, uly = int(rng.integers(0, 2000)), int(rng.integers(0, 2000))
    img_hdr = int(rng.choice([0, 1, 32, 100, 513]))
    if number_type in (64, 128):
        frames = (rng.standard_normal((n, h, w)) * 1000).astype(np.float64 if number_type == 64 else np.float32)
    else:
        hi = {2: 256, 4: 65536, 32: 2 ** 32}[number_type]
        frames = rng.integers(0, hi, size=(n, h, w), dtype=np.uint64)
    p = tmp_path / "f.his"
    p.write_bytes(F.his_file_bytes(frames, number_type, img_hdr, ulx=ulx, uly=uly))
    got, want = his_load(io, p), F.his_read(p)
    assert len(got) == len(want) == n
    for a, b in zip(got, want):
        assert a.shape == b.shape == (h, w)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("capacity,delays,expect_reread", [(32, (0, 0, 0), False), (2, (0, 0, 3000), True),
                                                           (4, (0, 300, 0, 900, 0, 0, 100, 0), None), (1, (0, 0, 0, 0), None),
                                                           (32, (0,) * 8, False)])
def test_shared_frames_read_each_frame_once(io, tmp_path, capacity, delays, expect_reread):
    """shared_frames (the read-once source of the multi-device driver): the consumers, each with its
    own detector row band, get exactly frame_stream's frames, indices and angles; every frame is converted from the files
    at most once by the shared streams (several frames are in production at a time, one thread each); a consumer that lags
    further than the ring is deep falls back to its own stream and still gets the same data. Eight consumers on a ring of
    four (and four on a ring of one) stress the claim order, stragglers and recycling; there only the data is checked."""
    io.paris_io_shared_scan.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                        _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p, _fp, _fp, C.POINTER(C.c_uint64)]
    rng = np.random.default_rng(7)
    d = tmp_path / "proj"
    d.mkdir()
    w, h = 9, 12
    frames = []
    for name, n in (("a.his", 5), ("b.his", 2), ("c.his", 6)):
        fr = rng.integers(0, 60000, size=(n, h, w)).astype(np.float32)
        frames += list(fr)
        (d / name).write_bytes(F.his_file_bytes(fr, 4, 16))
    (d / "b_broken.his").write_bytes(b"\x01" * 90)  # skipped by every stream
    ang = tmp_path / "ang.txt"
    ang.write_text(" ".join(str(2.5 * i) for i in range(len(frames))))
    nt = len(delays)
    bands = [[(0, h), (3, 4), (10, 2), (5, 7)][t % 4] for t in range(nt)]
    for quality in (1, 2):
        cap = 16
        n = (C.c_uint32 * nt)()
        idx = (C.c_uint32 * (nt * cap))()
        phi = (C.c_float * (nt * cap))()
        data = np.full((nt, cap, h, w), -7.0, np.float32)
        counters = (C.c_uint64 * 3)()
        rc = io.paris_io_shared_scan(str(d).encode(), 1, str(ang).encode(), quality, w, h, nt, (C.c_uint32 * nt)(*[b[0] for b in bands]),
                                     (C.c_uint32 * nt)(*[b[1] for b in bands]), (C.c_uint32 * nt)(*delays), capacity, cap, n, idx, phi,
                                     data.ctypes.data_as(_fp), counters)
        assert rc == 0
        keep = list(range(0, len(frames), quality))
        for t, (first, count) in enumerate(bands):
            assert n[t] == len(keep)
            assert list(idx[t * cap:t * cap + n[t]]) == keep
            assert list(phi[t * cap:t * cap + n[t]]) == [np.float32(2.5 * i) for i in keep]
            for j, i in enumerate(keep):
                assert np.array_equal(data[t, j, first:first + count], frames[i][first:first + count])
                assert np.all(data[t, j, :first] == -7.0) and np.all(data[t, j, first + count:] == -7.0)
        produced, served, reread = counters[0], counters[1], counters[2]
        assert produced <= len(keep)                       # no kept frame converted by the shared streams twice
        assert served + reread == nt * len(keep)           # every request answered
        if expect_reread is not None:
            assert produced == len(keep)                   # consumers in step: every frame went through the ring
            assert (reread > 0) == expect_reread


@pytest.mark.timeout(120)
def test_shared_frames_error_reaches_every_consumer(io, tmp_path):
    """when reading a frame fails inside the shared source (here the angle file is shorter than the frame set, so
    the lookup of frame 4's angle throws, as the reference's own loop would run off its angle vector: src/source.cpp:115-118),
    the consumers parked on that frame must be woken and report the error too -- not wait for ever. One consumer is slowed
    down so that it is certainly parked behind the producer at some point."""
    io.paris_io_shared_scan.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                        _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p, _fp, _fp, C.POINTER(C.c_uint64)]
    d = tmp_path / "proj"
    d.mkdir()
    w, h = 5, 4
    fr = np.arange(8 * h * w, dtype=np.float32).reshape(8, h, w)
    (d / "a.his").write_bytes(F.his_file_bytes(fr, 128, 0))
    ang = tmp_path / "ang.txt"
    ang.write_text("0 10 20")  # + the reader's trailing duplicate (Q15): 4 angles for 8 frames
    for delays in ((0, 0), (0, 2000), (2000, 0)):
        cap, nt = 16, 2
        n = (C.c_uint32 * nt)()
        idx = (C.c_uint32 * (nt * cap))()
        phi = (C.c_float * (nt * cap))()
        data = np.zeros((nt, cap, h, w), np.float32)
        counters = (C.c_uint64 * 3)()
        rc = io.paris_io_shared_scan(str(d).encode(), 1, str(ang).encode(), 1, w, h, nt, (C.c_uint32 * nt)(0, 0), (C.c_uint32 * nt)(h, h),
                                     (C.c_uint32 * nt)(*delays), 4, cap, n, idx, phi, data.ctypes.data_as(_fp), counters)
        assert rc == 1  # every worker came back (join returned) and at least one reported the failure
