#!/bin/bash
# Regenerates the measured evidence of a round on the GPU box, into gpurun_out/final/ (copy what is to be judged into profiles/):
#   bench lines of every single-GPU workload, rocprofv3 kernel statistics of the default bench command, HBM traffic (FETCH_SIZE and
#   WRITE_SIZE in separate passes), SQ instruction / LDS counters and the wave-state counters of the backprojection kernels.
# Run through gpurun from the repository root:  gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'
set -e
tag=${1:-rNN}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/final
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
run() { echo "== $*" >&2; "$@"; }
# 1. counters first: bench.py's roofline.traffic and fused_extension.roofline read the newest records under profiles/
#    (one --pmc pass each, nothing else traced); kernel times for the derived figures come from an unprofiled run
run python bench.py --steps 8 --warmup 2 --cpu-budget 0 --cpu-c1 0 --live-traffic 0 --workloads 0 --paris-loop 0 > $out/quick.json
A="--steps 1 --warmup 1 --batch 16 --spread 1 --cpu-budget 0 --cpu-c1 0 --noskip-step 0 --live-traffic 0"  # 16 timed launches that sample the whole circle
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof_fetch -- python3 bench.py $A > /dev/null
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_write -- python3 bench.py $A > /dev/null
python tools/pmc_traffic.py $out/prof_fetch $out/prof_write "2048^3 volume, 1440 projections @ 2048x2048 fp32" $out/${tag}_pmc_traffic_c3.json
echo "traffic done" >&2
S="--steps 1 --warmup 1 --batch 8 --spread 1 --cpu-budget 0 --cpu-c1 0 --noskip-step 0 --live-traffic 0 --fused-steps 2 --deferred-leg 0"
run rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq_a -- python3 bench.py $S > /dev/null
run rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq_b -- python3 bench.py $S > /dev/null
python tools/pmc_sq.py $out/pmc_sq_a $out/pmc_sq_b $out/quick.json $out/${tag}_pmc_sq_counters_c3.json "$S"
run rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_w1 -- python3 bench.py $S > /dev/null
python tools/pmc_wait.py $out/pmc_w1 $out/pmc_sq_a $out/quick.json > $out/${tag}_pmc_wave_states_c3.json
cp $out/${tag}_pmc_traffic_c3.json $out/${tag}_pmc_sq_counters_c3.json $out/${tag}_pmc_wave_states_c3.json profiles/
echo "counters done" >&2
# 2. the bench lines
run python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_c3.json
run python bench.py --steps 20 --warmup 5 --workload c3 --slices 256 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c4_slab_shape_1gpu.json
run python bench.py --steps 20 --warmup 5 --workload c2 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c2_1gpu.json
run python bench.py --steps 20 --warmup 5 --workload c1 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c1_1gpu.json
run python bench.py --steps 20 --warmup 5 --workload c1 --graph 1 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c1_graph_1gpu.json
run python bench.py --steps 20 --warmup 5 --workload c5 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c5_1gpu.json
# config 5 without the ROI crop: rank 3's 4096 x 4096 x 512 slab of the 8-GPU job over the whole 4096^3 grid, 360 launches over the circle
run python bench.py --steps 10 --warmup 1 --workload c5u --as-world 8 --as-rank-base 3 --batch 36 --spread 1 --cpu-budget 0 --cpu-c1 0 > $out/${tag}_bench_c5_uncropped_slab.json
echo "benches done" >&2
# 3. the same default command under the profiler (kernel trace only)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 --cpu-c1 0 --live-traffic 0 > $out/${tag}_bench_under_rocprof_c3.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/${tag}_kernel_stats_bench_c3_whole_job.csv
echo "kernel stats done" >&2
# 4. PARIS's own per-projection loop through the C++ mirror paris::hip, whole circles (tools/demo_bench.py), and the device timeline
#    of one such job (are the fused launches back to back?)
{
  python tools/demo_bench.py 512 360 2 demo,serial,immediate
  python tools/demo_bench.py 512 1536 2 demo,serial,immediate
  python tools/demo_bench.py 1024 720 2 demo,serial,immediate
  python tools/demo_bench.py 2048 1440 1 demo,serial
  python tools/demo_bench.py 2048 240 1 immediate
  echo "## BASELINE config 1 itself (256^3 volume from 512^2 projections, 360 of them): the loop is bound by its own host work per projection"
  for i in 1 2; do paris_amd/host/demo/paris_hip_demo 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out --vol 256 256 256 0.19973 | tr "\n" " "; echo; done
} > $out/${tag}_demo_paris_hip_mirror.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/$out/demo_trace -- $OLDPWD/paris_amd/host/demo/paris_hip_demo 1024 1024 0.2 0.2 0 0 500 500 0.5 720 lcg /dev/null --cycle 48 --no-out > /dev/null 2>&1 )
{ echo "# rocprofv3 --kernel-trace --memory-copy-trace -- paris_hip_demo 1024 1024 ... 720 lcg (PARIS's loop through paris::hip, whole circle): tools/timeline.py"; python tools/timeline.py $out/demo_trace; } > $out/${tag}_demo_timeline_1024.txt 2>&1
# the same loop with the time of every call of one iteration, for the default build (by reference, in-place group filter), the
# snapshotting build and the filter-at-once build; and the device timeline of the 360 x 512^2 -> 512^3 job
bash tools/mirror_split.sh > $out/${tag}_mirror_split.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/$out/demo_trace512 -- $OLDPWD/paris_amd/host/demo/paris_hip_demo 512 512 0.2 0.2 0 0 500 500 1.0 360 lcg /dev/null --cycle 48 --no-out > /dev/null 2>&1 )
{ echo "# rocprofv3 --kernel-trace --memory-copy-trace -- paris_hip_demo 512 512 ... 360 lcg (PARIS's loop through paris::hip, whole circle into 512^3): tools/timeline.py"; python tools/timeline.py $out/demo_trace512; } > $out/${tag}_demo_timeline_512.txt 2>&1
rm -rf $out/demo_trace512
echo "demo done" >&2
rm -rf $out/demo_trace $out/stats $out/prof_fetch $out/prof_write $out/pmc_sq_a $out/pmc_sq_b $out/pmc_w1 $out/quick.json
ls -la $out >&2
