mkdir -p gpurun_out
for ov in 0 1; do for fd in 1 0; do
python bench.py --workload c2 --steps 10 --warmup 2 --cpu-budget 0 --cpu-c1 0 --noskip-step 0 --live-traffic 0 --overlap $ov --filter-deferral $fd > gpurun_out/ab_c2_ov${ov}_fd${fd}.json 2>gpurun_out/ab_c2_err.log
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_c2_ov${ov}_fd${fd}.json") if l.startswith("{")][-1])
print("overlap $ov filter_deferral $fd: deferred", round(d["deferred_boundary"]["value"],1), "us/proj", round(d["deferred_boundary"]["us_per_projection"],1), "fused", round(d["fused_extension"]["value"],1), "kernel_ms", round(d["fused_extension"]["kernel_ms_per_launch"],2))
PY
done; done
