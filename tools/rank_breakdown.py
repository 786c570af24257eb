#!/usr/bin/env python3
"""Per-projection breakdown of ONE rank's step of the 8-GPU configurations, measured on one GPU with HIP events around each
stage (second half: is there anything left for filter sharding + an RCCL all-gather to win?).

For rank g of BASELINE config 4 (slab g of 2048 x 2048 x 256 of the 2048^3 grid) and config 5 (slab g of the 2048^3 ROI of the
4096^3 grid, half-precision projections) with the detector row band on (the default): the device copy that stands in for the
upload, the fused weighting + row filter, the backprojection. Next to it: what the same rank would have to RECEIVE per
projection if the filtered frames were all-gathered instead (7/8 of a 16 MiB frame, or of its band), at the xGMI rate a ring
all-gather sustains per link (MI355X_MICROARCH guide: 7 links x ~153 GB/s bidirectional, i.e. ~77 GB/s one way per link;
a ring moves (N - 1)/N of the result over ONE link per rank).

  python tools/rank_breakdown.py [--slabs 0,3,7] [--projections 64]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from paris_amd import backend as B  # noqa: E402

XGMI_ONE_WAY_GBPS = 76.5  # per link, one direction


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slabs", default="0,3,7")
    ap.add_argument("--projections", type=int, default=64)
    args = ap.parse_args()
    n = 2048
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    be = B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False)
    be.set_stage_fusion(True)
    raw = torch.rand((8, n, n), device=dev)
    work = torch.empty_like(raw)
    half = torch.empty((n, n), device=dev, dtype=torch.float16)
    vol = torch.zeros((256, n, n), device=dev)
    d_vol = be.wrap_volume(vol.data_ptr(), n, n, 256, owner=vol)
    projs = [be.wrap_projection(work[b].data_ptr(), n * 4, n, n, owner=work) for b in range(8)]
    for cfg in ("c4", "c5"):
        det = B.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / (1440 if cfg == "c4" else 3600))
        nat = B.calculate_volume_geometry(det)
        grid = 2048 if cfg == "c4" else 4096
        l_vx = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(grid))
        vg = B.VolumeGeometry(grid, grid, grid, l_vx, l_vx, l_vx)
        roi = None if cfg == "c4" else B.RegionOfInterest(1024, 3072, 1024, 3072, 1024, 3072)
        for g in [int(x) for x in args.slabs.split(",")]:
            first, count = B.slab_row_band(det, vg, n, n, 256, 256 * g, roi)
            rows = slice(first, first + count)
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.projections)]

            def one(i, record):
                p = projs[i % 8]
                p.idx = (i * 23) % (1440 if cfg == "c4" else 3600)
                if record:
                    ev[i][0].record()
                work[i % 8, rows].copy_(raw[i % 8, rows], non_blocking=True)
                if record:
                    ev[i][1].record()
                if cfg == "c5":
                    B.weight_filter_rows(be, p, det, first, count, half.data_ptr(), n * 2)
                else:
                    B.weight_rows(be, p, det, first, count)
                    B.filter_rows(be, p, det, first, count)
                if record:
                    ev[i][2].record()
                if cfg == "c5":
                    sn, cs = B.stage_angle(det, p.idx)
                    be.backproject_f16(half.data_ptr(), n * 2, n, n, d_vol, 256 * g, det, vg, True, roi, sn, cs, 0.0, 0.0)
                else:
                    B.backproject(be, p, d_vol, 256 * g, det, vg, False, False, None)
                if record:
                    ev[i][3].record()

            for i in range(8):
                one(i, False)
            torch.cuda.synchronize()
            for i in range(args.projections):
                one(i, True)
            torch.cuda.synchronize()
            t = np.array([[e[k].elapsed_time(e[k + 1]) * 1e3 for k in range(3)] for e in ev])  # us
            copy_us, wf_us, bp_us = t.mean(axis=0)
            total = copy_us + wf_us + bp_us
            px = 2 if cfg == "c5" else 4
            frame_bytes = n * n * px
            band_bytes = count * n * px
            out = {
                "config": cfg, "rank": g, "detector_row_band": [first, count],
                "copy_us": round(float(copy_us), 2), "weight_filter_us": round(float(wf_us), 2), "backproject_us": round(float(bp_us), 1),
                "share_of_step_that_sharding_could_remove": round(float((copy_us + wf_us) * 7.0 / 8.0 / total), 5),
                "allgather_receive_us_whole_frames": round(frame_bytes * 7.0 / 8.0 / (XGMI_ONE_WAY_GBPS * 1e9) * 1e6, 1),
                "allgather_receive_us_band_only": round(band_bytes * 7.0 / 8.0 / (XGMI_ONE_WAY_GBPS * 1e9) * 1e6, 1),
            }
            print(json.dumps(out), flush=True)
    be.close()


if __name__ == "__main__":
    main()
