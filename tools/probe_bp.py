#!/usr/bin/env python3
"""Backprojection kernel time as a function of where the slab sits in the 2048^3 grid and how thick it is."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from paris_amd import backend as B  # noqa: E402

n = 2048
det = B.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 1440)
nat = B.calculate_volume_geometry(det)
vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
be = B.Backend(0, synchronous=False)
rng = np.random.default_rng(1)
d_p = B.load(be, B.Projection(rng.random((n, n), dtype=np.float32), n, n))
cases = [(256, 0), (256, 256), (256, 512), (256, 896), (256, 1792), (512, 768), (1024, 512), (2048, 0)]
extra = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for slices, z_first in cases + extra:
    d_v = be.make_volume_device(n, n, slices)
    ms = []
    for rep in range(3):
        for a in (0, 17, 45, 90, 200):
            d_p.idx = a * 4
            B.backproject(be, d_p, d_v, z_first, det, vg, False, False, None)
            t = be.last_backproject_ms()
            if rep:
                ms.append(t)
    avg = sum(ms) / len(ms)
    vox = float(n) * n * slices
    print(json.dumps(dict(slices=slices, z_first=z_first, ms=avg, gbs=8 * vox / avg / 1e6, per_angle=[round(8 * vox / m / 1e6) for m in ms[:5]])), flush=True)
    be.free(d_v)
