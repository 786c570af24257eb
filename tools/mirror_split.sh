#!/bin/bash
# PARIS's per-projection loop through paris::hip with the time of every call of one iteration (paris_hip_demo prints the split):
# BASELINE config 1 itself, 512^2 -> 512^3, 1024^2 and 2048^2 natural volumes. Usage (GPU box): bash tools/mirror_split.sh > out.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
D=paris_amd/host/demo/paris_hip_demo
for i in 1 2 3; do
  echo "## 360 x 512^2 -> 256^3 (BASELINE config 1)"; $D 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out --vol 256 256 256 0.19973
  echo "## 360 x 512^2 -> 512^3"; $D 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out
  echo "## 720 x 1024^2 -> 1024 x 1024 x 1029"; $D 1024 1024 0.2 0.2 0 0 500 500 0.5 720 lcg /dev/null --cycle 48 --no-out
done
echo "## 1440 x 2048^2 -> 2048 x 2048 x 2090"; $D 2048 2048 0.2 0.2 0 0 500 500 0.25 1440 lcg /dev/null --cycle 48 --no-out
