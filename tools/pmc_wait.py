#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes of bench.py that carry the SQ wave-state counters (SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY +
SQ_WAIT_ANY + SQ_WAIT_INST_ANY, disjoint) per backprojection kernel: where the waves' time goes.

  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_w1 -- python3 bench.py <ARGS>
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_w2 -- python3 bench.py <ARGS>
  python tools/pmc_wait.py gpurun_out/pmc_w1 gpurun_out/pmc_w2 <bench json of an unprofiled run of the same <ARGS>>

The voxel-updates of a launch come from that bench line (slab size x projections per fused launch -- round 3 normalised by a
hard-wired 16 per launch while the bench ran 48: 78.15 "instructions per update" that were 26.05).

<ARGS> = --steps 1 --warmup 1 --batch 8 --cpu-budget 0 --fused-steps 1"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    dirs = [a for a in sys.argv[1:] if not a.endswith(".json")]
    jsons = [a for a in sys.argv[1:] if a.endswith(".json")]
    if len(jsons) != 1:
        raise SystemExit("usage: pmc_wait.py <counter dirs...> <bench.json>: the voxel-updates per launch are read from the bench line")
    with open(jsons[0]) as f:
        bench = json.loads([l for l in f if l.startswith("{")][-1])
    voxels = 1.0
    for v in bench["config"]["slab_per_gpu"]:
        voxels *= v
    fb = bench["fused_extension"]["projections_per_launch"]
    upd = {"fused": voxels * fb, "tile": voxels}
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                k = "tile" if "bp_tile_kernel" in n else "fused" if "bp_fused_kernel" in n else None
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        c = {n: sum(v) / len(v) for n, v in cs.items()}
        rec = {"counters": c}
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if n in c:
                    rec["share_of_wave_cycles:" + n] = c[n] / w
        if "SQ_INSTS_VALU" in c:
            rec["valu_instructions_per_voxel_update"] = c["SQ_INSTS_VALU"] * 64.0 / upd[k]
        rec["voxel_updates_per_launch"] = upd[k]
        out[k] = rec
    out["normalisation"] = {"slab": bench["config"]["slab_per_gpu"], "projections_per_fused_launch": fb, "bench_line": jsons[0]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
