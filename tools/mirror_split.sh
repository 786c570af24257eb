#!/bin/bash
# PARIS's per-projection loop through paris::hip with the time of every call of one iteration (paris_hip_demo prints the split):
# BASELINE config 1 itself, 512^2 -> 512^3, 1024^2 and 2048^2 natural volumes, for the default build (deferral by reference), the
# snapshotting build (references off: the filter then runs at once too) and the filter-at-once build, interleaved. Usage (GPU box): bash tools/mirror_split.sh > out.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
D=paris_amd/host/demo
for i in 1 2 3; do
  for exe in paris_hip_demo paris_hip_demo_upload_stream paris_hip_demo_snapshots paris_hip_demo_filter_at_once; do
    echo "## $exe: 360 x 512^2 -> 256^3 (BASELINE config 1)"; $D/$exe 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out --vol 256 256 256 0.19973
    echo "## $exe: 360 x 512^2 -> 512^3"; $D/$exe 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out
    echo "## $exe: 1536 x 512^2 -> 512^3"; $D/$exe 512 512 0.2 0.2 0 0 500 500 0.234375 1536 lcg /dev/null --cycle 48 --no-out
    echo "## $exe: 720 x 1024^2 -> 1024 x 1024 x 1029"; $D/$exe 1024 1024 0.2 0.2 0 0 500 500 0.5 720 lcg /dev/null --cycle 48 --no-out
  done
done
for exe in paris_hip_demo paris_hip_demo_upload_stream paris_hip_demo_snapshots paris_hip_demo_filter_at_once; do
  echo "## $exe: 1440 x 2048^2 -> 2048 x 2048 x 2090"; $D/$exe 2048 2048 0.2 0.2 0 0 500 500 0.25 1440 lcg /dev/null --cycle 48 --no-out
  echo "## $exe: 1440 x 2048^2 -> 2048 x 2048 x 256 (the slab of one rank of the 8-GPU job)"; $D/$exe 2048 2048 0.2 0.2 0 0 500 500 0.25 1440 lcg /dev/null --cycle 48 --no-out --vol 2048 2048 256 0.0998
done
