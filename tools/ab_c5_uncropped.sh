#!/bin/bash
# One rank's slab of BASELINE config 5 without the ROI crop (4096 x 4096 x 512 of the 4096^3 grid, fp16 in): the default tile order
# against the other dealt orders and tile depths on 4096^2 planes, same device, interleaved. Usage: bash tools/ab_c5_uncropped.sh > out.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
C="--workload c5u --as-world 8 --as-rank-base 3 --steps 4 --warmup 1 --batch 18 --spread 1 --fused-steps 4 --cpu-budget 0 --cpu-c1 0 --live-traffic 0 --workloads 0 --paris-loop 0"
python tools/ab_args.py --rounds 2 --common "$C" "" "--order 14" "--order 16" "--order 17" "--order 12" "--tz 8" "--order 18"
echo "## the outermost slab (rank 0: z 0..511, the cone leaves more tiles untouched)"
python tools/ab_args.py --rounds 1 --common "${C/--as-rank-base 3/--as-rank-base 0}" ""
