#!/usr/bin/env python3
"""Interleaved A/B of single-projection backprojection through the tile kernel (variant 0) and through the fused kernel
run with one projection (variant 4: every slice of a tile in flight before the first store), full 2048^3 volume or --slices."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool turns knobs that only the experiments build compiles in (make -C paris_amd/csrc EXPERIMENTS=1)
os.environ.setdefault("PARIS_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paris_amd", "lib",
                                                          "libparis_hip_experiments.so"))
import numpy as np  # noqa: E402

from paris_amd import backend as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=2048)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--configs", default="0:0:0,4:2:16,4:4:16,4:2:8,4:4:8,4:1:16", help="variant:vx:tz[,...]")
args = ap.parse_args()
n = 2048
det = B.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 500, 500, 0.25)
nat = B.calculate_volume_geometry(det)
vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
be = B.Backend(0, synchronous=False)
d_p = B.load(be, B.Projection(np.random.default_rng(1).random((n, n), dtype=np.float32), n, n))
d_v = be.make_volume_device(n, n, args.slices)
z_first = (n - args.slices) // 2
cfgs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(",")]
ms = {c: [] for c in cfgs}
for rnd in range(args.rounds + 1):
    for c in cfgs:
        be.set_backproject_variant(c[0])
        be.set_backproject_tuning(c[1], 0, c[2], 0)
        for a in (0, 45, 100, 200, 300):
            d_p.idx = a * 4
            B.backproject(be, d_p, d_v, z_first, det, vg, False, False, None)
            t = be.last_backproject_ms()
            if rnd:
                ms[c].append(t)
vox = float(n) * n * args.slices
for c in cfgs:
    med = statistics.median(ms[c])
    print(json.dumps(dict(cfg="variant %d vx %d tz %d" % c, median_ms=med, min_ms=min(ms[c]), gbs=8 * vox / med / 1e6,
                          frac=8 * vox / med / 1e6 / 8000)), flush=True)
