#!/bin/bash
# Same-device A/B of library builds: swaps each given libparis_hip.so into paris_amd/lib in turn (two rounds, interleaved) and
# prints bench.py's headline value, the fused kernel's rate and its per-launch min / max. GPU box only (run through gpurun);
# the library that was in place is restored at the end. Builds to compare are made beforehand, e.g.
#   make -C paris_amd/csrc EXTRA=-DPARIS_FUSED_PIPELINE=2 && cp paris_amd/lib/libparis_hip.so tools/_ab/libs/depth2.so
# usage: tools/ab_lib.sh tools/_ab/libs/*.so [-- bench.py arguments]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
cp paris_amd/lib/libparis_hip.so /tmp/ab_lib_keep.so
trap 'cp /tmp/ab_lib_keep.so paris_amd/lib/libparis_hip.so' EXIT
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    cp "$lib" paris_amd/lib/libparis_hip.so
    python bench.py --steps 8 --warmup 2 --cpu-budget 0 "$@" 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.readline())
f = j.get('fused_extension', {})
print('$(basename "$lib")', round(j['value'], 1), round(f.get('kernel_GVox_per_s_per_gpu', 0), 1), f.get('kernel_ms_min_max'))"
  done
done
