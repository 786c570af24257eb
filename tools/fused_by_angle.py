#!/usr/bin/env python3
"""Fused backprojection launch time by view angle: --batch consecutive projections (0.25 degrees apart) starting at each given angle,
on a 2048 x 2048 x --slices slab of the 2048^3 grid. GPU box only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from paris_amd import backend as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=512)
ap.add_argument("--angles", default="0,15,30,45,60,75,90,135,180,225,270,315")
ap.add_argument("--vx", type=int, default=0)
ap.add_argument("--tz", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--batch", type=int, default=48)
args = ap.parse_args()
n, P = 2048, args.batch
det = B.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 500, 500, 0.25)
nat = B.calculate_volume_geometry(det)
vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
be = B.Backend(0, synchronous=False)
rng = np.random.default_rng(1)
stack = be.make_projection_device(n, n * P)
be.copy_h2d(B.Projection(rng.random((n * P, n), dtype=np.float32), n, n * P), stack)
d_v = be.make_volume_device(n, n, args.slices)
z_first = (n - args.slices) // 2
be.set_backproject_tuning(args.vx, 0, args.tz, 0)
vox = float(n) * n * args.slices * P
for a in [float(x) for x in args.angles.split(",")]:
    sc = [B.stage_angle(det, int(round(a * 4)) + i) for i in range(P)]
    ms = []
    for rep in range(args.reps + 1):
        be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * n, P, n, n, d_v, z_first, det, vg, False, None,
                             [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        t = be.last_backproject_ms()
        if rep:
            ms.append(t)
    print(json.dumps(dict(angle=a, ms=min(ms), gvox=vox / min(ms) / 1e6)), flush=True)
