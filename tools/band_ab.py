#!/usr/bin/env python3
"""A/B of the detector row band (f4) on one rank's share of the 8-GPU configuration (C4): slab g of 2048x2048x256 of the
2048^3 volume, step = 8 projections x (copy + weight + filter + backproject), with all 2048 detector rows vs the slab's band.

  python tools/band_ab.py [slab=3] [steps=10]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from paris_amd import backend as B  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 2048
det = B.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 0.25)
nat = B.calculate_volume_geometry(det)
l_vx = float(np.float32(nat.l_vx_x))
vg = B.VolumeGeometry(n, n, n, l_vx, l_vx, l_vx)
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))  # torch's copies and the library's kernels on one explicit stream
be = B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False)
raw = torch.rand((8, n, n), device=dev)
work = torch.empty_like(raw)
vol = torch.zeros((256, n, n), device=dev)
d_vol = be.wrap_volume(vol.data_ptr(), n, n, 256, owner=vol)
projs = [be.wrap_projection(work[b].data_ptr(), n * 4, n, n, owner=work) for b in range(8)]
first, count = B.slab_row_band(det, vg, n, n, 256, 256 * g)
for name, (a, c) in (("all rows", (0, n)), ("band", (first, count)), ("all rows", (0, n)), ("band", (first, count))):
    rows = slice(a, a + c)

    def step(s):
        for b in range(8):
            projs[b].idx = s * 8 + b
            work[b, rows].copy_(raw[b, rows], non_blocking=True)
            B.weight_rows(be, projs[b], det, a, c)
            B.filter_rows(be, projs[b], det, a, c)
            B.backproject(be, projs[b], d_vol, 256 * g, det, vg, False, False, None)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step(1 + s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("slab %d, %-8s rows [%4d, %4d): %.3f ms per 8-projection step = %.1f GVox/s" % (
        g, name, a, a + c, dt * 1e3, 8 * 256 * n * n / dt / 1e9), flush=True)
