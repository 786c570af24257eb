#!/bin/bash
# A/B of the deferral's first group size (8, then doubling: round 4's choice) through PARIS's loop (paris_hip_demo), same device, interleaved:
# the experiments build reads PARIS_DEFER_RAMP; the demo binds to it through LD_PRELOAD. Usage: bash tools/ab_defer_ramp.sh > out.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
D=paris_amd/host/demo/paris_hip_demo
X=$PWD/paris_amd/lib/libparis_hip_experiments.so
one() { LD_PRELOAD=$X PARIS_DEFER_RAMP=$1 $D "${@:2:10}" lcg /dev/null --cycle 48 --no-out "${@:12}" | sed -n 2p | sed -e 's/ through paris::load.*//'; }
for round in 1 2 3; do
  for ramp in 8 16 24 48 4; do
    echo "ramp $ramp | 360 x 512^2 -> 512^3 | $(one $ramp 512 512 0.2 0.2 0 0 500 500 1.0 360)"
    echo "ramp $ramp | 360 x 512^2 -> 256^3 | $(one $ramp 512 512 0.2 0.2 0 0 500 500 1.0 360 --vol 256 256 256 0.19973)"
    echo "ramp $ramp | 720 x 1024^2 | $(one $ramp 1024 1024 0.2 0.2 0 0 500 500 0.5 720)"
    echo "ramp $ramp | 180 x 1024^2 | $(one $ramp 1024 1024 0.2 0.2 0 0 500 500 2.0 180)"
  done
done
for ramp in 8 16 48; do echo "ramp $ramp | 1440 x 2048^2 | $(one $ramp 2048 2048 0.2 0.2 0 0 500 500 0.25 1440)"; done
