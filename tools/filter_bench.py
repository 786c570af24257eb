#!/usr/bin/env python3
"""Times cosine weighting + row filter per projection on one GPU: the two launches (weight kernel, then the row filter:
variant 2 = first radix-16 kernel, variant 0 = table-twiddle radix-16 kernel) against the one fused launch
(paris_hip_set_stage_fusion). Output: microseconds per projection and the HBM rate at 8 B per pixel.

  python tools/filter_bench.py [--sizes 2048,1024,512] [--reps 200]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool turns knobs that only the experiments build compiles in (make -C paris_amd/csrc EXPERIMENTS=1)
os.environ.setdefault("PARIS_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paris_amd", "lib",
                                                          "libparis_hip_experiments.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2048,1024,512")
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import torch

    from paris_amd import backend as B
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    be = B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False)
    for n in [int(x) for x in args.sizes.split(",")]:
        det = B.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 0.25)
        work = torch.rand((8, n, n), device=dev, dtype=torch.float32)
        half = torch.empty((n, n), device=dev, dtype=torch.float16)
        projs = [be.wrap_projection(work[b].data_ptr(), n * 4, n, n, owner=work) for b in range(8)]

        def timed(body):
            for i in range(10):
                body(projs[i % 8])
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(args.reps):
                body(projs[i % 8])
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.reps * 1e3  # us

        def two(p):
            B.weight(be, p, det)
            B.filter(be, p, det)

        out = {"detector": "%dx%d" % (n, n), "filter_length": B.filter_size(n)}
        be.set_stage_fusion(False)
        be.set_filter_variant(2)
        out["two_launches_first_r16_us"] = timed(two)
        be.set_filter_variant(0)
        out["two_launches_table_r16_us"] = timed(two)
        out["filter_only_table_r16_us"] = timed(lambda p: B.filter(be, p, det))
        out["weight_only_us"] = timed(lambda p: B.weight(be, p, det))
        be.set_stage_fusion(True)
        out["fused_one_launch_us"] = timed(two)
        out["fused_one_launch_half_store_us"] = timed(lambda p: B.weight_filter_rows(be, p, det, 0, n, half.data_ptr(), n * 2))
        # a group of frames per launch (paris_hip_stage_weight_filter_batch: what bench.py's fused leg and the driver use per group)
        for frames in (8, 16, 48):
            stack = torch.rand((frames, n, n), device=dev, dtype=torch.float32)
            torch.cuda.synchronize()
            per_launch = timed(lambda p, st=stack, k=frames: B.weight_filter_batch(be, st.data_ptr(), n * 4, n * n * 4, k, n, n, det, 0, n))
            out["group_of_%d_frames_us_per_frame" % frames] = per_launch / frames
            del stack
        be.set_stage_fusion(False)
        out["fused_GBps_at_8B_per_pixel"] = 8.0 * n * n / (out["fused_one_launch_us"] * 1e-6) / 1e9
        print(json.dumps(out), flush=True)
    be.close()


if __name__ == "__main__":
    main()
