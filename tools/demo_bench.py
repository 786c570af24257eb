#!/usr/bin/env python3
"""Throughput of the PARIS-style per-projection loop through the C++ backend mirror paris::hip (paris_hip_demo: make /
load / weight / filter / backproject per projection, exactly the reference's call sequence), with the library's deferral on
(default build) and off (paris_hip_demo_immediate).

  python tools/demo_bench.py [n=1024] [n_proj=128]
"""
import os
import subprocess
import sys

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 128
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = "/tmp/paris_demo_bench"
os.makedirs(work, exist_ok=True)
np.random.default_rng(0).random((n_proj, n, n), dtype=np.float32).tofile(os.path.join(work, "in.raw"))
for exe in ("paris_hip_demo", "paris_hip_demo_immediate", "paris_hip_demo", "paris_hip_demo_immediate"):
    r = subprocess.run([os.path.join(root, "paris_amd", "host", "demo", exe), str(n), str(n), "0.2", "0.2", "0", "0", "500", "500",
                        repr(360.0 / n_proj), str(n_proj), os.path.join(work, "in.raw"), os.path.join(work, "out.raw")],
                       capture_output=True, text=True)
    print(exe, "|", " | ".join(r.stdout.strip().splitlines()), r.stderr.strip(), flush=True)
subprocess.run(["rm", "-rf", work])
