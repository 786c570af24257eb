#!/usr/bin/env python3
"""End-to-end timing of the rebuilt driver on a synthetic HIS data set: writes n_proj 16-bit frames of n x n pixels
into HIS files, runs paris.hip (HIS -> pipelined weight/filter/backproject -> DDBVF) and prints its stage report.

  python tools/e2e_bench.py [n] [n_proj] [frames_per_file] [workdir]
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from his_write import write_his  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_proj = int(sys.argv[2]) if len(sys.argv) > 2 else 720
per_file = int(sys.argv[3]) if len(sys.argv) > 3 else 60
work = sys.argv[4] if len(sys.argv) > 4 else "/tmp/paris_e2e"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.environ.get("PARIS_E2E_EXE") or os.path.join(root, "paris_amd", "host", "demo", "paris.hip")  # another build for an A/B

os.makedirs(os.path.join(work, "in"), exist_ok=True)
rng = np.random.default_rng(0)
t0 = time.time()
for k in range(0, n_proj, per_file):
    fr = rng.integers(0, 60000, size=(min(per_file, n_proj - k), n, n), dtype=np.uint16)
    write_his(os.path.join(work, "in", "scan_%04d.his" % (k // per_file)), fr, 32)
print("wrote %d frames of %dx%d u16 in %.1f s" % (n_proj, n, n, time.time() - t0), flush=True)
with open(os.path.join(work, "geo.ini"), "w") as f:
    f.write("n_row=%d\nn_col=%d\nl_px_row=0.2\nl_px_col=0.2\ndelta_s=0\ndelta_t=0\nd_so=500\nd_od=500\ndelta_phi=%r\n" % (n, n, 360.0 / n_proj))
if os.environ.get("PARIS_E2E_WRITE_ONLY"):  # leave the data set in `work` for a profiler run of paris.hip itself
    print("data set kept in", work)
    sys.exit(0)
t0 = time.time()
r = subprocess.run([exe, "--geometry", os.path.join(work, "geo.ini"), "--input", os.path.join(work, "in"), "--output",
                    os.path.join(work, "out")] + sys.argv[5:], capture_output=True, text=True)
print(r.stdout, r.stderr)
print("paris.hip wall %.2f s (exit %d)" % (time.time() - t0, r.returncode))
size = os.path.getsize(os.path.join(work, "out", "vol.ddbvf")) if r.returncode == 0 else 0
print("output %.2f GiB" % (size / 2 ** 30))
subprocess.run(["rm", "-rf", work])
