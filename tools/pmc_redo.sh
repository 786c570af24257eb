set -e
out=gpurun_out/final
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 --cpu-budget 0 --cpu-c1 0 --live-traffic 0 --workloads 0 --paris-loop 0 > $out/quick.json
S="--steps 1 --warmup 1 --batch 8 --spread 1 --cpu-budget 0 --cpu-c1 0 --noskip-step 0 --live-traffic 0 --fused-steps 2 --deferred-leg 0"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq_a -- python3 bench.py $S > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq_b -- python3 bench.py $S > /dev/null 2>&1
python tools/pmc_sq.py $out/pmc_sq_a $out/pmc_sq_b $out/quick.json $out/r04_pmc_sq_counters_c3.json "$S"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_w1 -- python3 bench.py $S > /dev/null 2>&1
python tools/pmc_wait.py $out/pmc_w1 $out/pmc_sq_a $out/quick.json > $out/r04_pmc_wave_states_c3.json
rm -rf $out/pmc_sq_a $out/pmc_sq_b $out/pmc_w1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/final/quick.json") if l.startswith("{")][-1])
print("live fused roofline:", d["fused_extension"]["roofline"])
PY
rm $out/quick.json
