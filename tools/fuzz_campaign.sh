#!/bin/bash
# The round's fuzz campaign on ONE commit (pass its hash: the GPU box has no .git): seeded random geometries through the default
# kernel path, the forced tile orders / depths and the fused batch against the oracle bit for bit (small and large volumes), the
# planes beyond 1024^2 on every voxel (8-slice tiles, 16-slice tiles with one slice in flight, deep nesting), the row filter's
# random sizes / bands, then smoke(). Product build; the variant subset once more against the experiments build.
#   gpurun --timeout 1100 -- "bash tools/fuzz_campaign.sh $(git rev-parse HEAD) > gpurun_out/r05_fuzz.txt 2>&1"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
echo "# fuzz campaign on commit $1 (library sha256 $(sha256sum paris_amd/lib/libparis_hip.so | cut -c1-16), experiments build $(sha256sum paris_amd/lib/libparis_hip_experiments.so | cut -c1-16))"
run() { echo "## $LABEL: $*"; "$@" 2>&1 | tail -1; }
LABEL="PARIS_FUZZ_SEEDS=6000 (small volumes)" PARIS_FUZZ_SEEDS=6000 run python -m pytest tests/test_gpu_parity.py -q -x -k random_geometries -p no:cacheprovider
LABEL="PARIS_FUZZ_BIG=1 PARIS_FUZZ_SEEDS=1500 (volumes up to 400^3)" PARIS_FUZZ_BIG=1 PARIS_FUZZ_SEEDS=1500 run python -m pytest tests/test_gpu_parity.py -q -x -k random_geometries -p no:cacheprovider
LABEL="PARIS_FUZZ_PLANES=40 (planes beyond 1024^2, every voxel)" PARIS_FUZZ_PLANES=40 run python -m pytest tests/test_gpu_full_volume.py -q -x -k large_planes -p no:cacheprovider
LABEL="PARIS_FILTER_FUZZ_SEEDS=2000 (row filter sizes and bands, two tests)" PARIS_FILTER_FUZZ_SEEDS=2000 run python -m pytest tests/test_gpu_parity.py -q -x -k random_sizes -p no:cacheprovider
echo "## the same small-volume draw against the experiments build (the orders that lost are then really drawn)"
LABEL="experiments build, PARIS_FUZZ_SEEDS=2000" PARIS_HIP_LIBRARY=$PWD/paris_amd/lib/libparis_hip_experiments.so PARIS_FUZZ_SEEDS=2000 run python -m pytest tests/test_gpu_parity.py -q -x -k random_geometries -p no:cacheprovider
LABEL="smoke" run python -c "import __graft_entry__ as g; g.smoke()"
