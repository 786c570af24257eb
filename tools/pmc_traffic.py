#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of bench.py into the per-launch HBM
traffic record bench.py reports as roofline.traffic.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --batch 2 --cpu-budget 0
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 1 --warmup 1 --batch 2 --cpu-budget 0
  python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write "<workload name>" profiles/rNN_pmc_traffic_c3.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950
FETCH_SIZE counts exactly half the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """fingerprint of the backprojection kernel's sources: bench.py flags a traffic record taken with other sources as stale"""
    h = hashlib.sha256()
    for name in ("backproject.hip", "bp_device.h"):
        with open(os.path.join(ROOT, "paris_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def values(d, counter, kernel="bp_tile"):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]


def main():
    fetch_dir, write_dir, workload, out_path = sys.argv[1:5]
    fetch, write = values(fetch_dir, "FETCH_SIZE"), values(write_dir, "WRITE_SIZE")
    launches = [{"FETCH_SIZE_KiB": a, "WRITE_SIZE_KiB": b, "fetch_bytes_corrected": a * 2048, "write_bytes": b * 1024,
                 "hbm_bytes": a * 2048 + b * 1024} for a, b in zip(fetch, write)]
    out = {"workload": workload, "kernel": "bp_tile_kernel, one projection per launch", "launches": launches,
           "traffic_bytes_per_launch": sum(l["hbm_bytes"] for l in launches) / len(launches),
           "kernel_source_sha16": kernel_source_sha16(),
           "note": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of wide streaming reads) + WRITE_SIZE x 1024"}
    # the fused kernel's launches of the same runs (bench.py's fused_extension steps), when there are any: volume read and written
    # once per launch + whatever of the projections' box staging misses the L2
    ff, fw = values(fetch_dir, "FETCH_SIZE", "bp_fused"), values(write_dir, "WRITE_SIZE", "bp_fused")
    if ff and fw:
        out["fused_kernel"] = {"launches": len(ff), "fetch_bytes_corrected_per_launch": sum(ff) / len(ff) * 2048,
                               "write_bytes_per_launch": sum(fw) / len(fw) * 1024,
                               "traffic_bytes_per_launch": sum(ff) / len(ff) * 2048 + sum(fw) / len(fw) * 1024}
    json.dump(out, open(out_path, "w"), indent=1)
    print(out["traffic_bytes_per_launch"])


if __name__ == "__main__":
    main()
