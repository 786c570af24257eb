#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (SQ counters, --output-format csv) of bench.py into the per-kernel record that bench.py's
fused_extension.roofline reads (VALU instructions per voxel-update) and DESIGN.md quotes (issue utilisation, LDS conflicts).

  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_sq_a -- python3 bench.py <ARGS>
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq_b -- python3 bench.py <ARGS>
  python tools/pmc_sq.py gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b <bench json of an unprofiled run> profiles/rNN_pmc_sq_counters_c3.json "<ARGS>"

<ARGS> = --steps 1 --warmup 1 --batch 8 --cpu-budget 0 --fused-steps 1 (2048^3 volume: counters are device-wide sums per launch).
Kernel times are taken from the unprofiled bench line (counter collection slows the kernels down)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = "bp_tile" if "bp_tile_kernel" in name else "bp_fused" if "bp_fused_kernel" in name else None
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    a_dir, b_dir, bench_json, out_path, args = sys.argv[1:6]
    with open(bench_json) as f:
        bench = json.loads([l for l in f if l.startswith("{")][-1])
    a, b = collect(a_dir), collect(b_dir)
    voxels = 1.0
    for v in bench["config"]["slab_per_gpu"]:
        voxels *= v
    fb = bench["fused_extension"]["projections_per_launch"]
    out = {"command": "rocprofv3 --pmc <4 SQ counters per pass> --output-format csv -- python3 bench.py " + args
           + "  (two passes; values are device-wide sums per launch, averaged over the launches seen)", "kernels": {}}
    for key, label, ms, updates in (
            ("bp_tile", "bp_tile_kernel (1 projection per launch)", bench["config"]["backproject_kernel_ms"], voxels),
            ("bp_fused", "bp_fused_kernel, default shape (%d projections per launch)" % fb, bench["fused_extension"]["kernel_ms_per_launch"], voxels * fb)):
        c = {}
        for src in (a, b):
            for name, vals in src.get(key, {}).items():
                c[name] = sum(vals) / len(vals)
        if not c:
            continue
        cycles = ms * 1e-3 * 2.4e9
        out["kernels"][label] = {
            "counters": c, "kernel_ms_without_counters": ms,
            "derived": {
                "valu_instructions_per_voxel_update": c["SQ_INSTS_VALU"] * 64.0 / updates if "SQ_INSTS_VALU" in c else None,
                "lds_instructions_per_voxel_update": c["SQ_INSTS_LDS"] * 64.0 / updates if "SQ_INSTS_LDS" in c else None,
                # A wave64 VALU instruction occupies its SIMD-32's issue port for 2 cycles (MI355X_MICROARCH.md: "issues each VALU
                # instruction over 2 cycles"): issue-port cycles used / cycles available at the nominal 2.4 GHz. (Round 3 divided
                # SQ_ACTIVE_INST_VALU -- quad-cycles summed over WAVES, several of which are in flight per SIMD -- by the SIMDs'
                # quad-cycles and got 1.35 for the fused kernel: not a utilisation.)
                "valu_issue_utilisation (SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles), 2.4 GHz)":
                    c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles) if "SQ_INSTS_VALU" in c else None,
                "valu_lane_instructions_per_second_T": c["SQ_INSTS_VALU"] * 64.0 / (ms * 1e-3) / 1e12 if "SQ_INSTS_VALU" in c else None,
                "fraction_of_measured_v_mul_f32_issue_rate (58.12 T lane-instr/s, profiles/r01_pkbench.txt)":
                    c["SQ_INSTS_VALU"] * 64.0 / (ms * 1e-3) / 58.12e12 if "SQ_INSTS_VALU" in c else None,
                "valu_active_wave_quad_cycles_per_simd_quad_cycle (SQ_ACTIVE_INST_VALU / (1024 x kernel cycles / 4); summed over waves, may exceed 1)":
                    c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles / 4.0) if "SQ_ACTIVE_INST_VALU" in c else None,
                "lds_busy_fraction (SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles))":
                    c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles) if "SQ_LDS_IDX_ACTIVE" in c else None,
                "lds_bank_conflict_share_of_lds_cycles":
                    c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
            },
        }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
