#!/usr/bin/env python3
"""Interleaved same-device A/B of bench.py argument sets (and optionally library builds): R rounds over all configurations in
one GPU call, one line per run. The only comparison that survives the +-3 % spread between the boxes of the pool.

  python tools/ab_args.py [--rounds 2] [--common "--steps 10 --warmup 2"] "<args of config A>" "<args of config B>" ...

A configuration may start with `lib=<path to a libparis_hip.so>` (that build is copied over paris_amd/lib/libparis_hip.so for
the run; the library in place is restored at the end) and / or `env=NAME=VALUE` words (environment of that run)."""
import argparse
import json
import os
import shlex
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "paris_amd", "lib", "libparis_hip.so")

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--common", default="--steps 10 --warmup 2 --cpu-budget 0 --cpu-c1 0 --live-traffic 0")
ap.add_argument("configs", nargs="+")
args = ap.parse_args()

keep = LIB + ".ab_keep"
shutil.copy(LIB, keep)
try:
    for rnd in range(args.rounds):
        for cfg in args.configs:
            words = shlex.split(cfg)
            env = dict(os.environ)
            lib = keep
            while words and (words[0].startswith("lib=") or words[0].startswith("env=")):
                if words[0].startswith("lib="):
                    lib = words[0][4:]
                else:
                    k, v = words[0][4:].split("=", 1)
                    env[k] = v
                words = words[1:]
            shutil.copy(lib, LIB)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + shlex.split(args.common) + words,
                               capture_output=True, text=True, cwd=ROOT, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                print(json.dumps({"cfg": cfg, "error": r.stderr[-400:]}), flush=True)
                continue
            j = json.loads(line[0])
            rf, f = j["roofline"], j.get("fused_extension") or {}
            print(json.dumps({"cfg": cfg, "round": rnd, "value": round(j["value"], 1),
                              "kernel_ms": round(j["config"]["backproject_kernel_ms"], 4), "frac": round(rf["frac"], 4),
                              "frac_without_skip": round(rf.get("frac_without_skip") or 0, 4),
                              "fused": round(f.get("value", 0), 1), "fused_kernel": round(f.get("kernel_GVox_per_s_per_gpu", 0), 1),
                              "deferred": round((j.get("deferred_boundary") or {}).get("value", 0), 1)}), flush=True)
finally:
    shutil.copy(keep, LIB)
    os.remove(keep)
