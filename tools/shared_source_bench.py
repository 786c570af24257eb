#!/usr/bin/env python3
"""Throughput of the read-once frame source of the multi-device driver (paris_amd/host/paris/source.h: shared_frames) on this
host's cores, no GPU involved: N consumer threads, each with a detector row band of 1/N of the rows (what N z-slabs ask for, give
or take the cone's overlap), drain a set of 16-bit HIS frames through one shared_frames object.

  python tools/shared_source_bench.py [n] [frames] [workdir] [threads ...]        (PARIS_IO_LIB=<another libparis_io.so> for an A/B)
"""
import ctypes as C
import os
import shutil
import sys
import time

import numpy as np

from his_write import write_his

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 96
work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/paris_shared_source"
threads = [int(a) for a in sys.argv[4:]] or [1, 2, 4, 8, 16]
lib = C.CDLL(os.environ.get("PARIS_IO_LIB") or os.path.join(ROOT, "paris_amd", "lib", "libparis_io.so"))
_fp, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
lib.paris_io_shared_scan.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                     _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p, _fp, _fp, C.POINTER(C.c_uint64)]
src = os.path.join(work, "in")
shutil.rmtree(work, ignore_errors=True)
os.makedirs(src)
rng = np.random.default_rng(0)
per_file = 24
for k in range(0, n_frames, per_file):
    write_his(os.path.join(src, "s%03d.his" % (k // per_file)), rng.integers(0, 60000, size=(min(per_file, n_frames - k), n, n), dtype=np.uint16), 32)
try:
    for nt in threads:
        rows = n // nt
        first = (C.c_uint32 * nt)(*[i * rows for i in range(nt)])
        count = (C.c_uint32 * nt)(*[rows] * nt)
        delay = (C.c_uint32 * nt)(*[0] * nt)
        got, idx, phi = (C.c_uint32 * nt)(), (C.c_uint32 * nt)(), (C.c_float * nt)()
        data = np.zeros((nt, 1, n, n), np.float32)
        cnt = (C.c_uint64 * 3)()
        best = None
        for _ in range(3):
            t = time.perf_counter()
            rc = lib.paris_io_shared_scan(src.encode(), 0, b"", 1, n, n, nt, first, count, delay, int(os.environ.get("PARIS_RING", "32")), 1, got, idx, phi, data.ctypes.data_as(_fp), cnt)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        assert rc == 0 and all(g == n_frames for g in got)
        print("%2d consumers: %d frames of %d^2 u16 in %.3f s = %.2f ms per frame (%d produced, %d served from the ring, %d reread)"
              % (nt, n_frames, n, best, best / n_frames * 1e3, cnt[0], cnt[1], cnt[2]), flush=True)
finally:
    shutil.rmtree(work, ignore_errors=True)
