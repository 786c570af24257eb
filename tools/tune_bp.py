#!/usr/bin/env python3
"""Sweeps the tuning knobs of the backprojection kernel on one C4-style slab (2048 x 2048 x Z of the 2048^3 grid,
2048^2 projection) and prints per-launch time and algorithmic GB/s (8 B per voxel-update). GPU box only."""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool turns knobs that only the experiments build compiles in (make -C paris_amd/csrc EXPERIMENTS=1)
os.environ.setdefault("PARIS_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paris_amd", "lib",
                                                          "libparis_hip_experiments.so"))
import numpy as np  # noqa: E402

from paris_amd import backend as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--vol", type=int, default=0, help="volume edge when it differs from the detector width (config 1: --n 512 --vol 256)")
    ap.add_argument("--vx", default="4,2,1")
    ap.add_argument("--unroll", default="1,2,4")
    ap.add_argument("--tz", default="16,32,64")
    ap.add_argument("--lds", default="16384,24576,32768")
    ap.add_argument("--angles", default="0,17,45,90,200")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--order", default="0,1,5")
    ap.add_argument("--fused", default="", help="fused kernel: projections per launch, e.g. 4,8,16 (uses --tz 8|16, --lds)")
    ap.add_argument("--fused-vx", default="0", help="lane widths of the fused kernel to try (0 = default 2; 1 with --tz 32: 32-slice tiles)")
    ap.add_argument("--slice", default="", help="slice kernel shapes, e.g. 16x4,8x2 (empty: tile kernel)")
    ap.add_argument("--nt", default="0,1")
    args = ap.parse_args()
    n = args.n
    det = B.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 1440)
    nat = B.calculate_volume_geometry(det)
    nv = args.vol or n
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(nv))
    vg = B.VolumeGeometry(nv, nv, nv, l_vx, l_vx, l_vx)
    be = B.Backend(0, synchronous=False)
    rng = np.random.default_rng(1)
    h = B.Projection(rng.random((n, n), dtype=np.float32), n, n)
    d_p = B.load(be, h)
    d_v = be.make_volume_device(nv, nv, args.slices)
    z_first = (nv - args.slices) // 2
    angles = [int(a) for a in args.angles.split(",")]
    voxels = float(nv) * nv * args.slices
    results = []
    if args.fused:
        import ctypes as C
        be.set_backproject_variant(0)
        nmax = max(int(x) for x in args.fused.split(","))
        stack = be.make_projection_device(n, n * nmax)
        hs = B.Projection(rng.random((n * nmax, n), dtype=np.float32), n, n * nmax)
        be.copy_h2d(hs, stack)
        for P, tz, lds, vx in itertools.product(*[[int(x) for x in s.split(",")] for s in (args.fused, args.tz, args.lds, args.fused_vx)]):
            be.set_backproject_tuning(vx, 0, tz, lds)
            sc = [B.stage_angle(det, 4 * (a + 3 * i)) for i, a in enumerate(range(P))]
            ms = []
            for rep in range(args.reps + 1):
                be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * n, P, n, n, d_v, z_first, det, vg, False, None,
                                     [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
                t = be.last_backproject_ms()
                if rep > 0:
                    ms.append(t)
            avg = sum(ms) / len(ms)
            r = dict(kernel="fused", P=P, tz=tz, lds=lds, vx=vx, ms=avg, ms_per_proj=avg / P, gvox=voxels * P / avg / 1e6,
                     hbm_gbs=(8.0 / P) * voxels * P / avg / 1e6)
            results.append(r)
            print(json.dumps(r), flush=True)
        print("BEST", json.dumps(max(results, key=lambda r: r["gvox"])))
        return
    shapes = [tuple(int(x) for x in sh.split("x")) for sh in args.slice.split(",") if sh]
    for nw, rpl in shapes:
        be.set_backproject_variant(3)
        be.set_backproject_slice_shape(nw, rpl)
        for lds, order, nt in itertools.product(*[[int(x) for x in s.split(",")] for s in (args.lds, args.order, args.nt)]):
            be.set_backproject_tuning(0, 0, 0, lds)
            be.set_backproject_order(order, nt)
            ms = []
            for rep in range(args.reps + 1):
                for a in angles:
                    d_p.idx = a * 4
                    B.backproject(be, d_p, d_v, z_first, det, vg, False, False, None)
                    t = be.last_backproject_ms()
                    if rep > 0:
                        ms.append(t)
            avg = sum(ms) / len(ms)
            r = dict(kernel="slice", nw=nw, rpl=rpl, lds=lds, order=order, nt=nt, ms=avg, ms_min=min(ms), ms_max=max(ms),
                     gbs=8 * voxels / avg / 1e6, gvox=voxels / avg / 1e6)
            results.append(r)
            print(json.dumps(r), flush=True)
    be.set_backproject_variant(2)
    be.set_backproject_slice_shape()
    for vx, un, tz, lds, order, nt in itertools.product(*[[int(x) for x in s.split(",")] for s in (args.vx, args.unroll, args.tz, args.lds, args.order, args.nt)]):
        be.set_backproject_tuning(vx, un, tz, lds)
        be.set_backproject_order(order, nt)
        ms = []
        for rep in range(args.reps + 1):
            for a in angles:
                d_p.idx = a * 4
                B.backproject(be, d_p, d_v, z_first, det, vg, False, False, None)
                t = be.last_backproject_ms()
                if rep > 0:
                    ms.append(t)
        avg = sum(ms) / len(ms)
        r = dict(vx=vx, unroll=un, tz=tz, lds=lds, order=order, nt=nt, ms=avg, ms_min=min(ms), ms_max=max(ms), gbs=8 * voxels / avg / 1e6,
                 gvox=voxels / avg / 1e6)
        results.append(r)
        print(json.dumps(r), flush=True)
    best = max(results, key=lambda r: r.get("gbs", 0))
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
