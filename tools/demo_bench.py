#!/usr/bin/env python3
"""Throughput of the PARIS-style per-projection loop through the C++ backend mirror paris::hip (paris_hip_demo: make /
load / weight / filter / backproject per projection, exactly the reference's call sequence, src/main.cpp:98-105):
the default build (deferral 48, fused launches on the second stream, uploads on the upload stream), the one-stream build
(paris_hip_demo_serial) and one launch per call (paris_hip_demo_immediate), interleaved.

  python tools/demo_bench.py [n=1024] [n_proj=128] [rounds=2] [exes=demo,serial,immediate]

Frames are the SURVEY 8c LCG noise, 48 distinct ones cycled (throughput does not depend on the data); the volume is the natural
one of an n x n detector and is not read back (--no-out).
"""
import os
import subprocess
import sys
import time

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
names = {"demo": "paris_hip_demo", "serial": "paris_hip_demo_serial", "immediate": "paris_hip_demo_immediate",
         "filter_deferral": "paris_hip_demo_filter_deferral"}
exes = [names[k] for k in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["demo", "serial", "immediate"])]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("## n = %d, %d projections (whole job: every projection of the circle through the loop)" % (n, n_proj), flush=True)
for _ in range(rounds):
    for exe in exes:
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(root, "paris_amd", "host", "demo", exe), str(n), str(n), "0.2", "0.2", "0", "0", "500", "500",
                            repr(360.0 / n_proj), str(n_proj), "lcg", "/dev/null", "--cycle", "48", "--no-out"],
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        # (the process also generates its 48 LCG frames, creates the HIP context and -- set_device(), PARIS_HIP_CTX_WARM -- the ctx's
        # streams and loads the code objects before the loop starts: none of it is in "projection loops")
        print(exe, "|", " | ".join(r.stdout.strip().splitlines()), r.stderr.strip(), "| whole process %.3f s" % wall, flush=True)
