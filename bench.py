#!/usr/bin/env python3
"""Benchmark of the FDK hot path (weight -> ramp row filter -> backproject) on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): GVoxel-updates/s = voxels x projections / seconds / 1e9, whole job over all N GPUs.
A step is one pass of the hot path over one batch of synthetic projections: for each projection a device copy of
the raw frame into the work buffer (stands in for the upload), paris::weight, paris::filter, paris::backproject
into the rank's z-slab. Inputs are resident in HBM before the timed region starts. The batch defaults to
ceil(n_proj / steps): the K timed steps are the WHOLE job of the workload (all 1440 projections of config 3, the full
circle), projection index = position in the job; `--batch B` restores a fixed batch (a partial arc, flagged in config).

Workloads (BASELINE.json configs; geometry per SURVEY.md 8d: l_px 0.2 mm, d_so = d_od = 500 mm, no offsets):
  c3 (default)  2048^3 volume, 1440 projections @ 2048x2048 fp32. N = 1: the whole volume on one GPU (config 3,
                the HBM-roofline run); N > 1: the same volume in N z-slabs, one per GPU (config 4 at N = 8):
                total work is fixed, so "scaling" is "strong". No collective on the data path.
  c2            1024^3 volume, 720 projections @ 1024x1024.
  c1            256^3 volume, 360 projections @ 512x512.
  c5            4096^3 grid with the {1024..3072}^3 ROI (2048^3 voxels allocated), 3600 projections, stored as IEEE half
                after filtering; the ROI is split into N z-slabs like c3.

The JSON line also carries
  roofline      for the dominant kernel (backprojection): algorithmic bytes per launch (8 B per voxel-update +
                one pass over the projection) / average launch duration from HIP events recorded on the launch
                stream inside the timed region, against the 8 TB/s HBM3E peak;
  cpu_baseline  the CPU oracle (this repo's restatement of the reference's OpenMP path, "port") timed on this
                host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.md section 3 asks for OMP_PROC_BIND=spread OMP_PLACES=cores on the CPU leg. Measured on the GPU box (a 16-core cgroup
# quota on a 256-CPU host, profiles/r03_ab_cpu_binding.txt) that binding costs the oracle 7.5x (0.19 against 1.46 GVox/s for the
# whole config-1 job), so the leg runs unbound unless the caller's environment binds it; "omp" in the line records what was used

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
HBM_COPY_GBS = 6290.0  # same guide: measured copy rate (tools/membench.hip: 6.67 TB/s linear nontemporal read-modify-write here)
# a slab of up to ~192 MiB stays in the 256 MiB Infinity Cache between launches: its read-modify-write is bounded by the cache, not
# by HBM. Measured in-place update rate of a 64 ... 192 MiB buffer (profiles/r02_rmw_cache_resident.txt): 6.7 ... 7.2 TB/s
CACHE_RESIDENT_BYTES = 192 << 20
CACHE_RMW_GBS = 7000.0

WORKLOADS = {
    "c3": dict(n_row=2048, n_col=2048, n_proj=1440, vol=(2048, 2048, 2048),
               name="2048^3 volume, 1440 projections @ 2048x2048 fp32"),
    "c2": dict(n_row=1024, n_col=1024, n_proj=720, vol=(1024, 1024, 1024),
               name="1024^3 volume, 720 projections @ 1024x1024 fp32"),
    "c1": dict(n_row=512, n_col=512, n_proj=360, vol=(256, 256, 256),
               name="256^3 volume, 360 projections @ 512x512 fp32"),
    # BASELINE config 5: the allocated output is the 2048^3 ROI of the 4096^3 grid; projections are rounded to IEEE
    # half after filtering and backprojected with fp32 interpolation and accumulation
    "c5": dict(n_row=2048, n_col=2048, n_proj=3600, vol=(4096, 4096, 4096), roi=(1024, 3072, 1024, 3072, 1024, 3072),
               f16=True, name="4096^3 grid, 2048^3 ROI, 3600 projections @ 2048x2048 fp16-in/fp32-accum"),
    # the stretch variant of config 5: no ROI, the whole 4096^3 grid -- 256 GiB, so only as the slabs of a larger
    # job: `--workload c5u --as-world 8 --as-rank-base r` is rank r's 4096 x 4096 x 512 slab (32 GiB) of the 8-GPU job
    # (/root/reference/src/main.cpp:124-130: the ROI is optional; src/cuda/subvolume_information.cpp:72-116: the split)
    "c5u": dict(n_row=2048, n_col=2048, n_proj=3600, vol=(4096, 4096, 4096), f16=True,
                name="4096^3 grid un-cropped, 3600 projections @ 2048x2048 fp16-in/fp32-accum"),
}


def geometry(B, w):
    det = B.DetectorGeometry(w["n_row"], w["n_col"], 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / w["n_proj"])
    nat = B.calculate_volume_geometry(det)
    dx, dy, dz = w["vol"]
    import numpy as np
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(w["n_row"]) / np.float32(dx))
    return det, B.VolumeGeometry(dx, dy, dz, l_vx, l_vx, l_vx)


def measured_traffic(w, world):
    """HBM bytes per backprojection launch from the PMC counters (FETCH_SIZE, WRITE_SIZE; separate rocprofv3
    --pmc passes, gfx950 correction applied) of the same workload, as committed under profiles/ by
    tools/pmc_traffic.py. Counters cannot be read from inside this process, so None when no profile matches."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if world == 1 and d.get("workload", "").startswith(w["name"]):
            best = (d["traffic_bytes_per_launch"], os.path.relpath(path, ROOT), d.get("kernel_source_sha16"))
    return best if best else (None, None, None)


def live_counters(argv_tail, launches=16, fused_batch=16):
    """Counters measured NOW, by child runs of this script under `rocprofv3 --pmc`, one counter per pass and nothing else traced
    (/opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE and WRITE_SIZE around `launches` single-projection launches spread over the
    whole circle (HBM bytes per launch: both counters are in KiB, FETCH_SIZE counts half the bytes of wide streaming reads on
    gfx950), and SQ_INSTS_VALU around a few fused launches (wave-level vector instructions; x 64 lanes / voxel-updates of a launch
    = instructions per voxel-update). Must be called before this process touches the GPU: the children are ordinary child
    processes and the profiler starts `python3 bench.py ...` directly. A pass that fails leaves its figure out."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return {}
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = ["python3", os.path.join(ROOT, "bench.py")] + argv_tail + ["--live-traffic", "0", "--steps", "1", "--warmup", "0", "--spread", "1",
                                                                      "--cpu-budget", "0", "--cpu-c1", "0", "--noskip-step", "0", "--workloads", "0", "--paris-loop", "0"]
    single = base + ["--batch", str(launches), "--fused-steps", "0"]
    fused = base + ["--batch", "1", "--fused-steps", "2", "--fused-batch", str(fused_batch), "--deferred-leg", "0"]
    work = tempfile.mkdtemp(prefix="paris_pmc_", dir="/tmp")

    def one_pass(counter, child, kernel):
        d = os.path.join(work, counter)
        r = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        files = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
        if r.returncode != 0 or not files:
            return None
        with open(files[0]) as f:
            return [float(row["Counter_Value"]) for row in csv.DictReader(f)
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter]

    out = {}
    try:
        fetch = one_pass("FETCH_SIZE", single, "bp_tile")
        write = one_pass("WRITE_SIZE", single, "bp_tile") if fetch else None
        if fetch and write and len(fetch) == launches == len(write):
            per_launch = [a * 2048.0 + b * 1024.0 for a, b in zip(fetch, write)]
            out["traffic"] = {"bytes_per_launch": sum(per_launch) / len(per_launch), "min": min(per_launch), "max": max(per_launch),
                              "launches": launches}
        valu = one_pass("SQ_INSTS_VALU", fused, "bp_fused")
        if valu:
            out["fused_valu"] = {"wave_instructions_per_launch": sum(valu) / len(valu), "launches": len(valu)}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        pass
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def other_workloads(common_tail):
    """The other BASELINE configs beside the headline: fresh child runs of this script, one at a time, after the
    headline's measurements are done -- config 1 and 2 as whole jobs, config 4's slab shape (2048 x 2048 x 256, --slices 256) as a
    whole job, config 5 sampled over the circle (360 of its 3600 launches), and one rank's slab of config 5 WITHOUT the ROI crop
    (4096 x 4096 x 512 of the 4096^3 grid, 72 launches over the circle) -- each reduced to the figures the headline reports for
    config 3. Children are ordinary child processes of this one (which keeps its GPU context: two processes on the card)."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    plans = [
        ("c1", ["--workload", "c1", "--steps", "10", "--warmup", "2"]),
        # config 1's step is launch bound (three launches of ~25 us of GPU work from Python): the same steps captured into hipGraphs
        ("c1_hip_graph", ["--workload", "c1", "--steps", "10", "--warmup", "2", "--graph", "1"]),
        ("c2", ["--workload", "c2", "--steps", "10", "--warmup", "2"]),
        ("c4_slab_shape", ["--workload", "c3", "--slices", "256", "--steps", "20", "--warmup", "2"]),
        ("c5_sampled", ["--workload", "c5", "--steps", "10", "--warmup", "1", "--batch", "36", "--spread", "1"]),
        # config 5 without the ROI crop: rank 3's 4096 x 4096 x 512 slab of the 8-GPU job over the whole 4096^3 grid, sampled
        ("c5_uncropped_slab", ["--workload", "c5u", "--as-world", "8", "--as-rank-base", "3", "--steps", "4", "--warmup", "1", "--batch", "18",
                               "--spread", "1", "--fused-steps", "4"]),
    ]
    out = {}
    for name, argv in plans:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv + common_tail
                               + ["--cpu-budget", "0", "--cpu-c1", "0", "--live-traffic", "0", "--workloads", "0", "--paris-loop", "0"],
                               cwd=ROOT, env=env, capture_output=True, text=True, timeout=150)  # (2-10 s each; bounded)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": (r.stderr or "no JSON line")[-300:]}
                continue
            d = json.loads(lines[-1])
        except (OSError, subprocess.SubprocessError, ValueError) as e:
            out[name] = {"error": str(e)[:300]}
            continue
        rf, cfg = d["roofline"], d["config"]
        e = {"workload": cfg["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
             "projections_timed": cfg["projections_timed"], "whole_job": cfg["whole_job"], "dtype": d["dtype"], "slab": cfg["slab_per_gpu"],
             "backproject_kernel_ms": cfg["backproject_kernel_ms"],
             "roofline": {"bound": rf["bound"], "frac": rf["frac"], "achieved": rf["achieved"], "unit": rf["unit"],
                          "frac_without_skip": rf.get("frac_without_skip"), "launches_timed": rf["launches_timed"]},
             "fused": d.get("fused_extension", {}).get("value"), "fused_kernel_ms_per_launch": d.get("fused_extension", {}).get("kernel_ms_per_launch"),
             "deferred": d.get("deferred_boundary", {}).get("value"), "hip_graph": cfg.get("hip_graph", False),
             "wall_seconds_of_the_child_run": time.perf_counter() - t0}
        if "frac_of_cache_resident_rate" in rf:
            e["roofline"]["frac_of_cache_resident_rate"] = rf["frac_of_cache_resident_rate"]
        out[name] = e
    return out


PARIS_LOOP_JOBS = [
    # (key, n, projections, explicit volume or None = the natural volume of an n x n detector)
    ("1440x2048^2->natural_2048x2048x2090", 2048, 1440, None),
    ("720x1024^2->natural_1024x1024x1029", 1024, 720, None),
    ("360x512^2->512^3", 512, 360, None),
    ("360x512^2->256^3_config1", 512, 360, (256, 256, 256, 2.0)),
]


def paris_loop(budget_s=40.0):
    """PARIS's own per-projection loop (/root/reference/src/main.cpp:98-105: make_projection_host / fill / load / weight / filter /
    backproject / free per projection, frames uploaded from pinned host memory: /root/reference/src/loader.cpp:28-33) through the C++
    mirror paris::hip, as child processes of paris_amd/host/demo/paris_hip_demo (built by build()): whole circles, from the first call
    to the end of the GPU work. The volume is neither read back nor written; 48 distinct noise frames are cycled. Each job reports
    the loop's rate, the host's own frame fill, the backend calls' share and the time of every call of one iteration."""
    import subprocess
    exe = os.path.join(ROOT, "paris_amd", "host", "demo", "paris_hip_demo")
    if not os.path.exists(exe):
        return {"error": "paris_hip_demo is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    import numpy as np
    out = {}
    t_all = time.perf_counter()
    for key, n, n_proj, vol in PARIS_LOOP_JOBS:
        if time.perf_counter() - t_all > budget_s:
            out[key] = {"error": "skipped: the leg's time budget was spent"}
            continue
        argv = [exe, str(n), str(n), "0.2", "0.2", "0", "0", "500", "500", repr(360.0 / n_proj), str(n_proj), "lcg", "/dev/null",
                "--cycle", "48", "--no-out", "--json"]
        if vol is not None:
            # the config's own voxel size: l_nat x (n_row / dim_x), as geometry() derives it for the other legs
            from paris_amd import backend as B
            det = B.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / n_proj)
            l_vx = float(np.float32(B.calculate_volume_geometry(det).l_vx_x) * np.float32(vol[3]))
            argv += ["--vol", str(vol[0]), str(vol[1]), str(vol[2]), repr(l_vx)]
        best = None
        t0 = time.perf_counter()
        try:
            for _ in range(1 if n >= 2048 else 2):  # (small jobs twice, the better run: a process start-up may land in the first)
                r = subprocess.run(argv, cwd=ROOT, capture_output=True, text=True, timeout=120)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not lines:
                    best = {"error": (r.stderr or r.stdout or "no JSON line")[-300:]}
                    break
                d = json.loads(lines[-1])
                if best is None or d["value"] > best["value"]:
                    best = d
        except (OSError, subprocess.SubprocessError, ValueError) as e:
            best = {"error": str(e)[:300]}
        if "error" not in best:
            best["host_fill_share"] = best["host_fill_seconds"] / best["seconds"]
            best["backend_call_share"] = best["backend_call_seconds"] / best["seconds"]
            best["wall_seconds_of_the_child_runs"] = time.perf_counter() - t0
        out[key] = best
    return out


def kernel_source_sha16():
    """fingerprint of the backprojection kernel's sources (as tools/pmc_traffic.py records it)"""
    import hashlib
    h = hashlib.sha256()
    for name in ("backproject.hip", "bp_device.h"):
        with open(os.path.join(ROOT, "paris_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def library_sha16():
    """fingerprint of the built library this process loaded"""
    import hashlib

    from paris_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(w, budget_s):
    """Times the oracle's backprojection on this host's cores: the workload's geometry, a slab of `slices`
    central slices, as many projections as fit the time budget."""
    import numpy as np

    from oracle import oracle as O
    cores = O.usable_cores()
    O.lib().po_set_num_threads(cores)
    det = O.DetectorGeometry(w["n_row"], w["n_col"], 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / w["n_proj"])
    nat = O.calculate_volume_geometry(det)
    dx, dy, dz = w["vol"]
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(w["n_row"]) / np.float32(dx))
    vg = O.VolumeGeometry(dx, dy, dz, l_vx, l_vx, l_vx)
    slices = max(1, min(dz, (64 << 20) // (dx * dy)))  # ~64 Mi voxels per projection
    z0 = (dz - slices) // 2
    vol = np.zeros((slices, dy, dx), np.float32)
    fs = O.filter_size(det.n_row)
    k = O.make_filter(fs, det.l_px_row)
    t_w = t_f = t_b = 0.0
    n = 0
    t_start = time.perf_counter()
    while n < w["n_proj"]:
        p = O.lcg_projection(det.n_row, det.n_col, n)
        t0 = time.perf_counter()
        O.weight(p, det)
        t1 = time.perf_counter()
        O.apply_filter(p, k, fs)
        t2 = time.perf_counter()
        s, c, ds, dt = O.backproject_constants(det, n)
        O.backproject(vol, p, z0, det, vg, s, c, ds, dt)
        t3 = time.perf_counter()
        t_w += t1 - t0
        t_f += t2 - t1
        t_b += t3 - t2
        n += 1
        if n >= 2 and time.perf_counter() - t_start > budget_s:
            break
    updates = float(slices) * dx * dy * n
    # whole hot path, scaled to the sample: weighting and filtering run once per projection for all dz slices of the
    # real job, so the sample (slices of dz) is charged that share of them
    share = float(slices) / dz
    t_path = t_b + (t_w + t_f) * share
    return {
        "value": updates / t_path / 1e9, "unit": "GVoxel-updates/s", "cores": cores, "kind": "port",
        "per_core": updates / t_path / 1e9 / cores, "omp": omp_binding(),
        "sample": "%s geometry, %d central slices (z %d..%d) x %d projections: backproject %.2f s + %d/%d of "
                  "weight %.2f s and filter %.2f s" % (w["name"], slices, z0, z0 + slices - 1, n, t_b, slices, dz,
                                                       t_w, t_f),
    }


def omp_binding():
    return {"proc_bind": os.environ.get("OMP_PROC_BIND"), "places": os.environ.get("OMP_PLACES")}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_c1():
    """BASELINE.md section 3 / SURVEY.md 8(d): the WHOLE config-1 job on this host's cores -- 256^3 volume, 360 projections @
    512 x 512 of the analytic 3-D Shepp-Logan phantom (tests/phantom.py, generated outside the timers), weight / filter /
    backproject timed separately around the oracle's calls (the restatement of src/openmp/*.cpp, all usable cores)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import phantom
    from oracle import oracle as O
    cores = O.usable_cores()
    O.lib().po_set_num_threads(cores)
    w = WORKLOADS["c1"]
    n, n_proj = w["n_row"], w["n_proj"]
    det = O.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / n_proj)
    nat = O.calculate_volume_geometry(det)
    dx, dy, dz = w["vol"]
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(dx))
    vg = O.VolumeGeometry(dx, dy, dz, l_vx, l_vx, l_vx)
    radius = 0.45 * dx * l_vx
    fs = O.filter_size(n)
    k = O.make_filter(fs, det.l_px_row)
    vol = np.zeros((dz, dy, dx), np.float32)
    t_w = t_f = t_b = 0.0
    t_gen0 = time.perf_counter()
    # the frames are synthesised up front on a thread pool (numpy releases the GIL inside its loops), outside every timer
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=cores) as pool:
        frames = list(pool.map(lambda i: phantom.projection(n, n, 0.2, 0.2, 500.0, 500.0, i * det.delta_phi, radius), range(n_proj)))
    for i in range(n_proj):
        p = frames[i]
        t0 = time.perf_counter()
        O.weight(p, det)
        t1 = time.perf_counter()
        O.apply_filter(p, k, fs)
        t2 = time.perf_counter()
        s, c, ds, dt = O.backproject_constants(det, i)
        O.backproject(vol, p, 0, det, vg, s, c, ds, dt)
        t3 = time.perf_counter()
        t_w += t1 - t0
        t_f += t2 - t1
        t_b += t3 - t2
    wall = time.perf_counter() - t_gen0
    updates = float(dx) * dy * dz * n_proj
    return {
        "value": updates / t_b / 1e9, "unit": "GVoxel-updates/s", "cores": cores, "kind": "port",
        "per_core": updates / t_b / 1e9 / cores, "cpu_model": cpu_model(), "omp": omp_binding(),
        "backproject_s": t_b, "weight_s": t_w, "filter_s": t_f,
        "whole_path_value": updates / (t_b + t_w + t_f) / 1e9,
        "volume_checksum": float(vol.sum(dtype=np.float64)),
        "sample": "the whole job of BASELINE config 1: %s, analytic Shepp-Logan frames; backproject timed alone "
                  "(steady clock around the %d calls), weight and filter listed; %.1f s wall including frame synthesis"
                  % (w["name"], n_proj, wall),
    }


VALU_PEAK_LANE_INSTR = 58.12e12  # plain fp32 v_mul_f32 lane-instructions/s on this chip (tools/pkbench.hip, profiles/r01_pkbench.txt)


def fused_sq_profile():
    """VALU instructions per voxel-update of the fused kernel from the newest committed SQ counter profile
    (rocprofv3 --pmc SQ_INSTS_VALU ...: counters cannot be read from inside this process)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sq_counters*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for name, k in d.get("kernels", {}).items():
            if name.startswith("bp_fused_kernel"):
                return k["derived"], os.path.relpath(path, ROOT)
    return None, None


def fused_report(fused, fb, f16, voxels_rank, voxels_all, n_row, n_col, per_launch, live_valu=None):
    rate = per_launch / (fused["kernel_ms"] * 1e-3) if fused["kernel_ms"] > 0 else 0.0  # voxel-updates/s of the kernel
    hbm_gbps = ((8.0 * voxels_rank + (2.0 if f16 else 4.0) * n_row * n_col * fb) / (fused["kernel_ms"] * 1e-3) / 1e9
                if fused["kernel_ms"] > 0 else 0.0)
    rep = {
        "what": "the same per-projection copy, then weight + filter of the group's frames (one launch for all of them unless "
                "--filter-batch 0) and one fused launch that adds %d projections "
                "(paris_hip_backproject_batch; bit-identical volume); not the headline because the plugin boundary is one "
                "projection per call" % fb,
        "value": voxels_all * fb * fused["steps"] / fused["seconds"] / 1e9,
        "unit": "GVoxel-updates/s",
        "kernel_ms_per_launch": fused["kernel_ms"],
        "kernel_ms_min_max": [fused["kernel_ms_min"], fused["kernel_ms_max"]],
        "kernel_GVox_per_s_per_gpu": rate / 1e9,
        "algorithmic_bytes_per_update": 8.0 / fb,
        "projections_per_launch": fb,
        "hbm_GBps": hbm_gbps,
        "hbm_frac": hbm_gbps / HBM_PEAK_GBS,
    }
    derived, src = fused_sq_profile()
    if live_valu:  # SQ_INSTS_VALU of this run's own fused launches (a child pass under rocprofv3 --pmc)
        derived = {"valu_instructions_per_voxel_update": live_valu["wave_instructions_per_launch"] * 64.0 / per_launch,
                   "lds_bank_conflict_share_of_lds_cycles": (derived or {}).get("lds_bank_conflict_share_of_lds_cycles")}
        src = "measured in this run (rocprofv3 --pmc SQ_INSTS_VALU around %d fused launches of a child run)" % live_valu["launches"]
    if derived is not None:
        ipu = derived["valu_instructions_per_voxel_update"]
        achieved = rate * ipu  # lane-instructions/s: one lane executes `ipu` vector instructions per voxel-update
        rep["roofline"] = {
            "bound": "valu_issue", "achieved": achieved / 1e12, "peak": VALU_PEAK_LANE_INSTR / 1e12,
            "unit": "T lane-instr/s", "frac": achieved / VALU_PEAK_LANE_INSTR,
            "valu_instructions_per_voxel_update": ipu,
            "lds_bank_conflict_share_of_lds_cycles": derived.get("lds_bank_conflict_share_of_lds_cycles"),
            "source": "%s (SQ_INSTS_VALU per voxel-update) x this run's kernel rate; peak = plain v_mul_f32 issue rate, "
                      "profiles/r01_pkbench.txt" % src,
        }
    else:
        rep["roofline"] = None
    return rep


def octant_stats(kernel_ms, idx_of_launch, n_proj):
    """min / mean / max kernel time per 45-degree octant of the projection angle (phi = idx * 360 / n_proj)"""
    bins = [[] for _ in range(8)]
    for ms, idx in zip(kernel_ms, idx_of_launch):
        bins[min(7, int(8.0 * (idx % n_proj) / n_proj))].append(ms)
    out = []
    for o, b in enumerate(bins):
        if b:
            out.append({"deg": [45 * o, 45 * (o + 1)], "launches": len(b), "min_ms": min(b), "mean_ms": sum(b) / len(b),
                        "max_ms": max(b)})
    return out


def launch_ranks(args, argv, grace=120.0, _cmd=None):
    """`python3 bench.py --gpus N` with N > 1 and no RANK in the environment: the job starts itself. One rank per GPU is started
    through torch.distributed.run as an ordinary CHILD process (this process never touches the GPU: counting devices does not
    initialise it, and nothing is exec'ed), rank 0's one JSON line is relayed to stdout, the launcher's exit status is returned.
    Stands where the reference fans out one thread per device (/root/reference/src/main.cpp:157-167)."""
    import socket
    import subprocess
    n = args.gpus
    if _cmd is None and args.dist_backend == "nccl":
        import torch
        visible = torch.cuda.device_count()
        if visible < n:
            sys.stderr.write("bench.py: --gpus %d over nccl (= RCCL) needs one GPU per rank, %d visible: nothing was run "
                             "(a rehearsal with ranks sharing a card: --dist-backend gloo --device 0)\n" % (n, visible))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes fails without it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--max-restarts", "0",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if _cmd is not None:  # (tests: another child in the launcher's place)
        cmd = list(_cmd)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)  # (stderr is inherited)
    import threading
    seen = {"line": None, "at": None}

    def relay():
        for text in child.stdout:
            if text.startswith("{"):
                seen["line"], seen["at"] = text, time.perf_counter()
            elif text.strip():
                sys.stderr.write(text)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    # the line is the result: ranks that do not come down after it (a hang while the process group is torn down) must not cost the
    # caller its own time limit
    while child.poll() is None:
        if seen["at"] is not None and time.perf_counter() - seen["at"] > grace:
            sys.stderr.write("bench.py: the ranks did not exit within %.0f s of the result line: terminating them\n" % grace)
            child.terminate()
            try:
                child.wait(timeout=15)
            except subprocess.TimeoutExpired:
                child.kill()
            break
        time.sleep(0.2)
    rc = child.wait()
    reader.join(timeout=5)
    line = seen["line"]
    if line is not None and rc != 0 and seen["at"] is not None and time.perf_counter() - seen["at"] > grace:
        rc = 0  # (the measurement was complete; only the teardown was cut short)
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks ended without a JSON line\n")
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--batch", type=int, default=0, help="projections per step; 0 (default) = ceil(n_proj / steps), i.e. the "
                    "timed steps are the whole job")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline (0 disables)")
    ap.add_argument("--vx", type=int, default=0)
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--tz", type=int, default=0)
    ap.add_argument("--lds-bytes", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0, help="backprojection kernel variant (0: default; 5: two-pass, column constants precomputed per projection)")
    ap.add_argument("--order", type=int, default=-1, help="workgroup -> tile order of the backprojection kernel (-1: the library's default)")
    ap.add_argument("--fused-steps", type=int, default=8, help="extra steps with the fused multi-projection kernel, "
                    "reported as fused_extension next to the headline (0 disables); the steps are spread over the circle")
    ap.add_argument("--fused-batch", type=int, default=48, help="projections per fused launch in fused_extension and "
                    "deferred_boundary (2..64; the headline step keeps single-projection launches)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal of the N > 1 path "
                    "with several ranks sharing one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="GPU index for this rank (default: LOCAL_RANK)")
    ap.add_argument("--row-band", type=int, default=1, help="1 (default): each rank uploads, weights and filters only the "
                    "detector rows its z-slab can read (paris_hip_slab_row_band; the whole detector at N = 1); 0: all rows")
    ap.add_argument("--stage-fusion", type=int, default=1, help="1 (default): weight + filter in one launch (paris_hip_set_stage_fusion); "
                    "0: one launch per call")
    ap.add_argument("--filter-shard", type=int, default=0, help="N > 1 (SURVEY f4, second half): 1 = rank r weights and filters "
                    "projections r, r + N, ... once for everybody and the ranks exchange the rows of each other's detector bands per "
                    "group of N projections (paris_amd.sharding.exchange_filtered, RCCL point-to-point batch). Off by default: it "
                    "costs more than it removes (profiles/r02_rank_breakdown_c4_c5.txt)")
    ap.add_argument("--graph", type=int, default=0, help="1: every timed step is captured once into a hipGraph (torch.cuda.graph on "
                    "the bench stream: copy + weight/filter + backproject of all its projections, angles baked in) before the timed "
                    "region and replayed inside it -- for the launch-bound small configurations; kernel times then come from a "
                    "separate eager pass (events cannot be read back from a captured stream)")
    ap.add_argument("--noskip-step", type=int, default=1, help="1 (default): after the timed region, one more step with "
                    "paris_hip_set_backproject_skip_invalid(0), reported as roofline.frac_without_skip")
    ap.add_argument("--cpu-c1", type=int, default=1, help="1 (default, N = 1 only): also time the oracle on the whole BASELINE "
                    "config-1 job (cpu_baseline_c1, a few seconds of CPU work plus ~20 s of frame synthesis)")
    ap.add_argument("--filter-batch", type=int, default=1, help="fused_extension leg: 1 (default) = the group's frames are weighted and "
                    "filtered by one launch (paris_hip_stage_weight_filter_batch), 0 = one launch per frame")
    ap.add_argument("--filter-deferral", type=int, default=1, help="deferred_boundary leg: 1 (default) = the filter() of each projection is "
                    "held back with its weight() and runs on the library's snapshots, one launch per group (paris_hip_set_filter_deferral: "
                    "what paris::hip switches on); 0 = one weight + filter launch per projection")
    ap.add_argument("--overlap", type=int, default=1, help="deferred_boundary leg: 1 (default: what the C++ mirror paris::hip runs) = "
                    "fused launches of deferred calls on the ctx's second stream beside the next group's copies and filters "
                    "(paris_hip_set_backproject_overlap); 0 (the bare library's default) = on the ctx stream. Config 1: 920 -> 1040-1050 "
                    "GVox/s, configs 2-5 within noise (profiles/r04_ab_overlap_c1.txt, r04_ab_overlap_resident_c2.txt)")
    ap.add_argument("--live-traffic", type=int, default=1, help="1 (default; N = 1, config 3 only): roofline.traffic is measured in this "
                    "run -- before the GPU is touched, two child runs of this script under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` add 16 "
                    "projections spread over the circle each (about 20 s per pass); 0: the newest matching record under profiles/")
    ap.add_argument("--spread", type=int, default=0, help="1: with a partial job (--batch), the timed projections are spread evenly "
                    "over the circle instead of taken from its start (counter runs: a few launches that sample every angle)")
    ap.add_argument("--slices", type=int, default=0, help="rehearsal only: cap the volume depth (0 = the workload's)")
    ap.add_argument("--final-gather", choices=["checksums", "slabs", "off"], default="checksums",
                    help="N > 1, after the timed region and timed separately: the job's one collective. checksums (default): "
                    "all-gather of per-slab checksums; slabs: the slabs themselves gathered on rank 0 (4 GiB each at N = 8)")
    ap.add_argument("--deferred-leg", type=int, default=1, help="0: the fused extension without the deferred_boundary leg -- every "
                    "bp_fused_kernel launch of the run then adds exactly --fused-batch projections (counter passes normalise per launch; "
                    "the deferred leg's first groups are launched early, after 8, 16 and 32 calls)")
    ap.add_argument("--workloads", type=int, default=1, help="1 (default; N = 1, the whole config-3 job only): after the headline's "
                    "measurements the other BASELINE configs are run as child processes and summarised under `workloads` "
                    "(config 1, config 2, config 4's slab shape, config 5 sampled)")
    ap.add_argument("--as-world", type=int, default=0, help="rehearsal of a larger job on fewer processes (the GPU pool allows at "
                    "most 6 processes on a card, so the 8-rank job cannot run on one GPU at once): the volume is partitioned as for "
                    "this many ranks and this process takes the slab of rank --as-rank-base + RANK; two runs of 4 processes cover "
                    "the 8 slabs of BASELINE configs 4 / 5 at their full shape")
    ap.add_argument("--as-rank-base", type=int, default=0)
    ap.add_argument("--block-checksums", type=int, default=0, help="N = 1: also report the float64 sum of each of this many z blocks "
                    "of the volume (the reference's split rule), i.e. the slab checksums an N-rank run of the same projections must reproduce")
    ap.add_argument("--dist-timeout", type=float, default=240.0, help="seconds a rank waits in the rendezvous (init_process_group) "
                    "before it gives up: one dead rank ends the run instead of hanging the others until the caller's limit")
    ap.add_argument("--paris-loop", type=int, default=1, help="1 (default; N = 1, the whole config-3 job only): PARIS's own per-projection "
                    "loop through the C++ mirror (paris_hip_demo, child processes) is timed for four jobs and reported as `paris_loop`")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # called the way the N = 1 line is (`python3 bench.py --gpus N ...`): this process becomes the launcher
        sys.exit(launch_ranks(args, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: libraries that print there from native code (RCCL's version banner at
    # communicator creation, the ROCm runtime) are sent to stderr for the life of the process; the line is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # where this invocation's wall time goes, leg by leg (reported as `legs_seconds`: the driver's one command must stay within minutes)
    marks = [("start", time.perf_counter())]

    def mark(name):
        marks.append((name, time.perf_counter()))

    # roofline.traffic, measured live: child processes under the profiler, before this process initialises the GPU
    live = {}
    under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if (args.live_traffic and args.gpus == 1 and args.workload == "c3" and args.slices == 0 and "RANK" not in os.environ
            and not under_profiler):
        tail = []
        for name in ("vx", "unroll", "tz", "lds_bytes", "variant", "order", "row_band", "stage_fusion"):
            tail += ["--" + name.replace("_", "-"), str(getattr(args, name))]
        live = live_counters(tail, fused_batch=max(2, min(64, args.fused_batch)))

    mark("live_counter_passes")
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N > 1 through torch.distributed.run)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible -- the hot path has no CPU fallback")
    from paris_amd import sharding

    n_visible = torch.cuda.device_count()
    dev_index = sharding.device_of_rank(local_rank, n_visible, args.device)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # under torch.distributed.run, also at N = 1
        import torch.distributed as dist
        import datetime
        limit = datetime.timedelta(seconds=max(10.0, args.dist_timeout))
        if args.dist_backend == "nccl":
            if n_visible < world:  # (the launcher checks too; this is the torchrun-started path)
                raise SystemExit("bench.py: %d ranks over nccl (= RCCL) need one GPU each, %d visible" % (world, n_visible))
            dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.dist_backend, timeout=limit)
        # the rendezvous is over: the job's own collectives (barriers around the timed region, the final gather) may wait longer
        # for a rank that is merely slow than for one that never arrived
        try:
            dist.distributed_c10d._set_pg_timeout(datetime.timedelta(seconds=max(600.0, args.dist_timeout)))
        except (AttributeError, RuntimeError, TypeError):
            pass
        # every rank must sit on a GPU of its own (RCCL needs it; a wrapped LOCAL_RANK would silently halve the job's HBM)
        placement = sharding.gather_placement(dist, dev_index, torch.cuda.get_device_properties(dev_index))
        sharding.check_placement(placement, exclusive=(args.dist_backend == "nccl"))

    from paris_amd import backend as B

    w = dict(WORKLOADS[args.workload])
    if args.slices > 0:
        w["vol"] = (w["vol"][0], w["vol"][1], min(args.slices, w["vol"][2]))
        w["name"] += " [rehearsal: %d slices]" % w["vol"][2]
    det, vol_geo = geometry(B, w)
    roi = B.RegionOfInterest(*w["roi"]) if "roi" in w else None
    out_geo = B.apply_roi(vol_geo, *w["roi"]) if roi is not None else vol_geo  # what is allocated: the ROI (src/main.cpp:124-130)
    vworld = args.as_world if args.as_world > 0 else world  # (rehearsal: the partition of a larger job, --as-world)
    vrank = args.as_rank_base + rank if args.as_world > 0 else rank
    if not 0 <= vrank < vworld or (args.as_world > 0 and args.filter_shard):
        raise SystemExit("bench.py: --as-rank-base %d + rank %d outside the %d-rank partition (or --filter-shard with --as-world)"
                         % (args.as_rank_base, rank, vworld))
    info = sharding.make_subvolume_info(out_geo, vworld)  # one z-slab per rank
    z_first, z_count = sharding.slab_of_task(info, vrank)

    # One stream for everything in the step: torch's copies and fills and the library's kernels are ordered on it (a ctx with a
    # private stream would run unordered beside torch's default-stream work). PARIS_BENCH_STREAM picks which stream, for A/B:
    # "side" (default) = an explicit torch stream, "legacy" = the legacy default stream, "private" = the ctx's own stream (unordered
    # with torch's work: timing only).
    mode = os.environ.get("PARIS_BENCH_STREAM", "side")
    if mode == "side":
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
    stream = None if mode == "private" else torch.cuda.current_stream(dev).cuda_stream
    be = B.Backend(dev_index, stream=stream, synchronous=False)
    be.set_backproject_tuning(args.vx, args.unroll, args.tz, args.lds_bytes)
    if args.order >= 0:
        be.set_backproject_order(args.order, -1)
    if args.variant:
        be.set_backproject_variant(args.variant)
    # paris::weight is held back and rides along in the load of the paris::filter call that follows: one launch for the pair
    be.set_stage_fusion(bool(args.stage_fusion))
    be.set_backproject_overlap(False)  # (the headline's one-launch-per-call steps have nothing to overlap; switched on for the deferred leg)

    n_row, n_col, n_proj = w["n_row"], w["n_col"], w["n_proj"]
    batch = args.batch if args.batch > 0 else -(-n_proj // max(1, args.steps))
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345)  # every rank holds the same projection stack (north star: "each GPU holding the full projection stack")
    fb = max(2, min(64, args.fused_batch))
    nb = max(16, fb)  # work slots and distinct raw frames; a step cycles through them (stream order makes the reuse safe)
    raw = torch.rand((nb, n_col, n_row), generator=gen, device=dev, dtype=torch.float32)
    work = torch.empty_like(raw)
    # the slab comes from the library's make_volume_device, as PARIS's does (src/make_volume.cpp:36): zero-filled, and known to the
    # library as a volume nothing but backprojections writes (paris_hip_set_backproject_skip_invalid); torch sees the same memory
    d_vol = be.make_volume_device(out_geo.dim_x, out_geo.dim_y, z_count)

    class _DeviceMemory:
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}

    vol = torch.as_tensor(_DeviceMemory(d_vol.ptr, (z_count, out_geo.dim_y, out_geo.dim_x)), device=dev)
    f16 = bool(w.get("f16"))
    half = torch.empty((n_col, n_row), device=dev, dtype=torch.float16) if f16 else None
    pitch = work.stride(1) * 4
    projs = [be.wrap_projection(work[b].data_ptr(), pitch, n_row, n_col, owner=work) for b in range(nb)]

    # f4: the detector rows this rank's slab can read for any angle; rows outside never reach its voxels
    band_first, band_count = 0, n_col
    if args.row_band:
        band_first, band_count = B.slab_row_band(det, vol_geo, out_geo.dim_x, out_geo.dim_y, z_count, z_first, roi)
    band = slice(band_first, band_first + band_count)

    launched = []  # projection index of every single-projection backprojection call, in call order

    shard = bool(args.filter_shard) and world > 1
    if shard:
        if f16:
            raise SystemExit("bench.py --filter-shard: fp32 workloads only")
        on_device = args.dist_backend == "nccl"
        slabs = [sharding.slab_of_task(info, t) for t in range(world)]
        bands = [B.slab_row_band(det, vol_geo, out_geo.dim_x, out_geo.dim_y, slabs[t][1], slabs[t][0], roi) if args.row_band else (0, n_col)
                 for t in range(world)]
        shard_recv = torch.zeros((world, n_col, n_row), device=dev, dtype=torch.float32)
        recv_projs = [be.wrap_projection(shard_recv[q].data_ptr(), pitch, n_row, n_col, owner=shard_recv) for q in range(world)]

    def sharded_step(first_idx, cnt):
        for j0 in range(0, cnt, world):
            j = j0 + rank  # this rank's projection of the group: weighted and filtered here, once, for every rank
            mine = None
            if j < cnt:
                p = projs[0]
                p.idx = (first_idx + j) % n_proj
                work[0].copy_(raw[j % nb], non_blocking=True)   # the whole frame: every rank needs other rows of it
                B.weight_rows(be, p, det, 0, n_col)
                B.filter_rows(be, p, det, 0, n_col)
                mine = work[0]
            sharding.exchange_filtered(dist, mine, [shard_recv[q] for q in range(world)], bands, rank, world, on_device=on_device)
            for q in range(world):
                if j0 + q < cnt:
                    rp = recv_projs[q]
                    rp.idx = (first_idx + j0 + q) % n_proj
                    B.backproject(be, rp, d_vol, z_first, det, vol_geo, False, roi is not None, roi)
                    launched.append(rp.idx)

    def step(first_idx, count=None, indices=None):
        """one pass of the hot path over `count` consecutive projections from first_idx, or over the given projection indices"""
        if shard:
            return sharded_step(first_idx, batch if count is None else count)
        todo = indices if indices is not None else [first_idx + j for j in range(batch if count is None else count)]
        for j, at in enumerate(todo):
            b = j % nb
            p = projs[b]
            p.idx = at % n_proj
            work[b, band].copy_(raw[b, band], non_blocking=True)              # stands in for the upload
            if f16:
                # config 5: weight + filter in one launch that stores the band as IEEE half (no separate conversion pass)
                B.weight_filter_rows(be, p, det, band_first, band_count, half.data_ptr(), n_row * 2)
                sn, cs = B.stage_angle(det, p.idx)
                be.backproject_f16(half.data_ptr(), n_row * 2, n_row, n_col, d_vol, z_first, det, vol_geo, True, roi, sn, cs, 0.0, 0.0)
            else:
                B.weight_rows(be, p, det, band_first, band_count)             # src/main.cpp:102 (held back: stage fusion)
                B.filter_rows(be, p, det, band_first, band_count)             # :103 (weights in its load)
                B.backproject(be, p, d_vol, z_first, det, vol_geo, False, roi is not None, roi)  # :104
            launched.append(p.idx)

    def barrier():
        if dist is not None:
            if args.dist_backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # warmup steps run the projections just before index 0 (the end of the circle), the timed steps the job from index 0
    for s in range(args.warmup):
        step((s - args.warmup) * batch)
    torch.cuda.synchronize()
    timed_launches = args.steps * batch
    be.backproject_timing_arm(min(65536, max(1, timed_launches)))
    del launched[:]

    graphs = None
    if args.graph:
        if mode != "side":
            raise SystemExit("bench.py --graph needs the explicit bench stream (PARIS_BENCH_STREAM=side)")
        # kernel durations for the roofline: one eager pass over the job with events, outside the timed region
        for s in range(args.steps):
            step(s * batch)
        kernel_ms = be.backproject_timing_collect()
        kernel_idx = launched[-len(kernel_ms):] if kernel_ms else []
        be.backproject_timing_arm(0)  # no event records in the captured stream
        graphs = []
        for s in range(args.steps):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                step(s * batch)
            graphs.append(g)
        torch.cuda.synchronize()
        # the eager pass and the captures each added the job once (a capture enqueues nothing): start the timed job from zero
        vol.zero_()
        torch.cuda.synchronize()

    barrier()
    torch.cuda.synchronize()
    mark("setup_and_warmup")
    t0 = time.perf_counter()
    if graphs is not None:
        for g in graphs:
            g.replay()
    elif args.spread and timed_launches < n_proj:
        for s in range(args.steps):
            step(0, indices=[((s * batch + j) * n_proj) // timed_launches for j in range(batch)])
    else:
        for s in range(args.steps):
            step(s * batch)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    mark("timed_region")

    if graphs is None:
        kernel_ms = be.backproject_timing_collect()
        kernel_idx = launched[-len(kernel_ms):] if kernel_ms else []
    else:
        be.backproject_timing_arm(1)

    # ---- one more step with paris_hip_set_backproject_skip_invalid(0): every tile is read and written, also those no ray
    # reaches -- the kernel's rate on exactly the algorithmic bytes. Its projections are spread over the whole circle.
    noskip_ms = None
    if args.noskip_step and world == 1:  # (N > 1: the slabs' checksums are gathered afterwards and must be the job's)
        spread = [(j * n_proj) // batch for j in range(batch)] if batch <= n_proj else list(range(batch))
        be.set_backproject_skip_invalid(False)
        be.backproject_timing_arm(min(65536, max(1, len(spread))))
        step(0, indices=spread)
        noskip_ms = be.backproject_timing_collect()
        be.set_backproject_skip_invalid(True)
        be.backproject_timing_arm(1)
        del launched[-len(spread):]

    mark("skip_off_step")
    # ---- extension, outside the headline: the same step with ONE fused launch per batch (paris_hip_backproject_batch)
    fused = None
    if args.fused_steps > 0:
        sc = [B.stage_angle(det, b) for b in range(n_proj)]
        stride = work.stride(0) * 4
        halves = torch.empty((fb, n_col, n_row), device=dev, dtype=torch.float16) if f16 else None
        # the fused steps start at evenly spaced angles of the circle (consecutive projections inside a launch)
        starts = [(s * n_proj) // args.fused_steps for s in range(args.fused_steps)]

        def fused_step(first_idx):
            idx = [(first_idx + b) % n_proj for b in range(fb)]
            for b in range(fb):
                work[b, band].copy_(raw[b, band], non_blocking=True)
                if args.filter_batch:
                    continue
                if f16:
                    B.weight_filter_rows(be, projs[b], det, band_first, band_count, halves[b].data_ptr(), n_row * 2)
                else:
                    B.weight_rows(be, projs[b], det, band_first, band_count)
                    B.filter_rows(be, projs[b], det, band_first, band_count)
            if args.filter_batch:
                # the group's frames weighted and filtered by ONE launch (paris_hip_stage_weight_filter_batch; bit-identical)
                B.weight_filter_batch(be, work.data_ptr(), pitch, stride, fb, n_row, n_col, det, band_first, band_count,
                                      halves.data_ptr() if f16 else None, n_row * 2 if f16 else 0, n_row * n_col * 2 if f16 else 0)
            if f16:
                be.backproject_batch_f16(halves.data_ptr(), n_row * 2, n_row * n_col * 2, fb, n_row, n_col, d_vol, z_first, det,
                                         vol_geo, roi is not None, roi, [sc[i][0] for i in idx], [sc[i][1] for i in idx], 0.0, 0.0)
            else:
                be.backproject_batch(work.data_ptr(), pitch, stride, fb, n_row, n_col, d_vol, z_first, det, vol_geo,
                                     roi is not None, roi, [sc[i][0] for i in idx], [sc[i][1] for i in idx], 0.0, 0.0)

        fused_step(n_proj - fb)
        torch.cuda.synchronize()
        be.backproject_timing_arm(args.fused_steps)
        barrier()
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for first in starts:
            fused_step(first)
        torch.cuda.synchronize()
        barrier()
        tf = max_over_ranks(time.perf_counter() - tf0)
        fms = be.backproject_timing_collect()
        fused = {"steps": args.fused_steps, "seconds": tf, "kernel_ms": sum(fms) / max(1, len(fms)),
                 "kernel_ms_min": min(fms) if fms else 0.0, "kernel_ms_max": max(fms) if fms else 0.0}

        if args.deferred_leg:
            # ---- the same per-projection calls as the headline, with the library's deferral switched on: every
            # paris_hip_backproject call snapshots its projection, `batch` of them are added by one fused launch
            be.set_backproject_deferral(fb)
            be.set_filter_deferral(bool(args.filter_deferral) and not f16)
            be.set_backproject_overlap(bool(args.overlap))
            step(n_proj - fb, fb)
            be.flush()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            td0 = time.perf_counter()
            for first in starts:
                step(first, fb)
            td_host = time.perf_counter() - td0  # the host's share: every call has returned, the GPU may still be working
            be.flush()
            torch.cuda.synchronize()
            barrier()
            td = max_over_ranks(time.perf_counter() - td0)
            be.set_filter_deferral(False)
            be.set_backproject_deferral(1)
            be.set_backproject_overlap(False)
            fused["deferred_seconds"] = td
            fused["deferred_host_seconds"] = td_host
            fused["deferred_projections"] = len(starts) * fb

    mark("fused_and_deferred_legs")
    # ---- the job's one collective (north star: "no RCCL collective needed beyond a final gather"), timed on its own
    gather = None
    # (N = 1 under torch.distributed.run: only when the slabs are asked for -- it exercises dist.gather and the float64
    # all_gather_into_tensor on the nccl backend, the two calls a first real multi-GPU run depends on and gloo cannot vouch for)
    if dist is not None and (world > 1 or args.final_gather == "slabs") and args.final_gather != "off":
        on_device = args.dist_backend == "nccl"
        barrier()
        torch.cuda.synchronize()
        tg0 = time.perf_counter()
        try:
            res = sharding.final_gather(dist, vol, info, rank, world, full=(args.final_gather == "slabs"), on_device=on_device,
                                        task_base=vrank - rank)
            torch.cuda.synchronize()
            barrier()
            tg = max_over_ranks(time.perf_counter() - tg0)
            gather = {"mode": args.final_gather, "seconds": tg, "backend": "rccl" if on_device else args.dist_backend,
                      "rccl_ranks_seen": dist.get_world_size(), "slab_checksums": res["checksums"],
                      "checksum_of_checksums": res["checksum_of_checksums"]}
            if "gathered_bytes" in res:  # (0 bytes cross a link at N = 1: rank 0 gathers its own slab)
                gather["gathered_bytes"] = res["gathered_bytes"]
                gather["GBps_into_rank0"] = res["gathered_bytes"] / tg / 1e9
                gather["gathered_matches_checksums"] = res.get("gathered_matches_checksums")
        except RuntimeError as e:  # the headline was measured before this point: report the failure instead of losing the line
            gather = {"mode": args.final_gather, "error": str(e)[:500]}

    voxels_rank = float(z_count) * out_geo.dim_x * out_geo.dim_y
    voxels_all = float(out_geo.dim_z) * out_geo.dim_x * out_geo.dim_y
    if args.as_world > 0:  # only the slabs of the participating ranks are processed
        voxels_all = float(sum(sharding.slab_of_task(info, args.as_rank_base + r)[1] for r in range(world))) * out_geo.dim_x * out_geo.dim_y
    blocks = None
    if args.block_checksums > 0 and world == 1:
        binfo = sharding.make_subvolume_info(out_geo, args.block_checksums)
        torch.cuda.synchronize()
        blocks = []
        for t in range(binfo.num):
            b0, bc = sharding.slab_of_task(binfo, t)
            blocks.append(float(sharding.slab_checksum(vol[b0:b0 + bc])[0].item()))

    # every rank's own kernel statistics, so that a first multi-GPU run explains itself: a slow rank, a rank on another build
    # of the kernel, a rank whose slab reads a wider detector band
    per_rank = None
    if dist is not None:
        ms = kernel_ms or [0.0]
        mine = {"rank": rank, "as_rank": vrank, "device": dev_index, "slab": [z_first, z_count], "detector_row_band": [band_first, band_count],
                "launches": len(kernel_ms), "kernel_ms_min": min(ms), "kernel_ms_mean": sum(ms) / len(ms), "kernel_ms_max": max(ms),
                "kernel_GVox_per_s": voxels_rank / (sum(ms) / len(ms) * 1e-3) / 1e9 if sum(ms) > 0 else 0.0,
                "kernel_ms_without_skip": (sum(noskip_ms) / len(noskip_ms)) if noskip_ms else None,
                "kernel_source_sha16": kernel_source_sha16(), "library_sha16": library_sha16()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    updates_all = voxels_all * batch * args.steps

    if rank == 0:
        traffic, traffic_src, traffic_sha = measured_traffic(w, world)
        traffic_live = False
        if live.get("traffic"):
            lt = live["traffic"]
            traffic, traffic_sha, traffic_live = lt["bytes_per_launch"], kernel_source_sha16(), True
            traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around child runs of "
                           "bench.py adding %d projections spread over the circle; FETCH_SIZE x 2048 (KiB, gfx950 half-count of wide "
                           "streaming reads) + WRITE_SIZE x 1024; per-launch min %.4g max %.4g bytes"
                           % (lt["launches"], lt["min"], lt["max"]))
        avg_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        algo_bytes = 8.0 * voxels_rank + (2.0 if f16 else 4.0) * n_row * n_col  # per launch: RMW of the slab + one projection pass
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        distinct = len(set(kernel_idx))
        octants = octant_stats(kernel_ms, kernel_idx, n_proj)
        out = {
            "metric": "GVoxel-updates/s (voxels x projections / s), FDK hot path weight+filter+backproject",
            "value": updates_all / elapsed / 1e9,
            "unit": "GVoxel-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16-in/f32" if f16 else "f32",
            "data": "synthetic (uniform noise projections generated on device; zero-initialised volume)",
            "config": {
                "workload": w["name"] + (", 1 GPU" if world == 1 else ", %d z-slabs on %d GPUs" % (world, world)),
                "projections_per_step": batch,
                "projections_timed": timed_launches,
                "whole_job": timed_launches == n_proj,
                "angles_covered": {"distinct_projections": distinct, "of": n_proj,
                                   "first_deg": 0.0, "last_deg": 360.0 * ((timed_launches - 1) % n_proj) / n_proj
                                   if timed_launches < n_proj else 360.0 * (n_proj - 1) / n_proj},
                "slab_per_gpu": [out_geo.dim_x, out_geo.dim_y, z_count],
                "detector_row_band_rank0": [band_first, band_count],
                "parallelism": "z-slab per GPU, no collective on the data path",
                "stage_fusion": bool(args.stage_fusion),
                "hip_graph": bool(args.graph),
                "volume_device_address": hex(vol.data_ptr()),
                "filter_shard": shard,
                "backproject_kernel_ms": avg_ms,
                "backproject_kernel_ms_min": min(kernel_ms) if kernel_ms else 0.0,
                "backproject_kernel_ms_max": max(kernel_ms) if kernel_ms else 0.0,
                "backproject_kernel_ms_by_octant": octants,
                "backproject_GVox_per_s_per_gpu": voxels_rank / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_this_run": traffic_live,
                # the counters come from a committed rocprofv3 --pmc run (they cannot be read inside this process): stale when the
                # kernel's sources have changed since
                "traffic_is_of_this_kernel": (traffic_sha == kernel_source_sha16()) if traffic is not None else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                # achieved is priced on the ALGORITHMIC bytes (8 B for every voxel-update the reference performs); measured traffic
                # below them = tiles no ray of the projection reaches, which the library leaves untouched (the reference adds +0
                # there; bit-identical for a library-allocated volume: paris_hip_set_backproject_skip_invalid, DESIGN.md 4.1, profiles/HISTORY.md)
                "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                # the DRAM-side fraction: measured bytes (tiles no ray reaches are not moved) over this run's kernel time
                "frac_dram": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                "kernel": "bp_tile_kernel (one projection per launch, 8 B per voxel-update)",
                "launches_timed": len(kernel_ms),
            },
        }
        if 4.0 * voxels_rank <= CACHE_RESIDENT_BYTES:
            out["roofline"]["slab_is_infinity_cache_resident"] = True
            out["roofline"]["cache_resident_rmw_GBps"] = CACHE_RMW_GBS
            out["roofline"]["frac_of_cache_resident_rate"] = achieved / CACHE_RMW_GBS
            out["roofline"]["note"] = ("the %.0f MiB slab stays in the 256 MiB Infinity Cache between launches: `frac` (against the HBM "
                                       "peak, as the contract asks) is not the binding bound here; the measured in-place update rate of a "
                                       "cache-resident buffer is %.0f GB/s" % (4.0 * voxels_rank / 2 ** 20, CACHE_RMW_GBS))
        if noskip_ms:
            ns = sum(noskip_ms) / len(noskip_ms)
            out["roofline"]["frac_without_skip"] = algo_bytes / (ns * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["kernel_ms_without_skip"] = ns
            out["roofline"]["without_skip_launches"] = len(noskip_ms)
        if octants:
            worst = max(octants, key=lambda o: o["mean_ms"])
            best = min(octants, key=lambda o: o["mean_ms"])
            out["roofline"]["frac_worst_octant"] = algo_bytes / (worst["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["frac_best_octant"] = algo_bytes / (best["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if fused is not None:
            per_launch = voxels_rank * fb
            out["fused_extension"] = fused_report(fused, fb, f16, voxels_rank, voxels_all, n_row, n_col, per_launch, live.get("fused_valu"))
            if fused.get("deferred_seconds"):
                out["deferred_boundary"] = {
                    "what": "the headline's step unchanged -- one paris_hip_backproject call per projection -- with "
                            "paris_hip_set_backproject_deferral(%d): the library snapshots each call's projection and adds %d of "
                            "them per fused launch (the first groups of a sequence after 8, 16 and 32 calls; bit-identical volume)%s; the call sequence of PARIS's per-projection loop with the projections resident on the device"
                            % (fb, fb, ", and paris_hip_set_filter_deferral(1): each call pair weight() + filter() is held back and runs "
                               "on the snapshots, one launch per group" if (args.filter_deferral and not f16) else ""),
                    "filter_deferral": bool(args.filter_deferral and not f16),
                    "second_stream": bool(args.overlap),
                    "value": voxels_all * fused["deferred_projections"] / fused["deferred_seconds"] / 1e9,
                    "unit": "GVoxel-updates/s",
                    # how long the calls themselves took to return (Python + ctypes + HIP enqueue): close to 1 = the leg is bound
                    # by the host loop, not by the GPU
                    "host_enqueue_share": fused["deferred_host_seconds"] / fused["deferred_seconds"],
                    "us_per_projection": fused["deferred_seconds"] / fused["deferred_projections"] * 1e6,
                }
        if gather is not None:
            out["final_gather"] = gather
        if blocks is not None:
            out["block_checksums"] = {"blocks": len(blocks), "sums": blocks,
                                      "what": "float64 sum of every z block of the volume after everything this run added to it"}
        if args.as_world > 0:
            out["config"]["as_world"] = vworld
            out["config"]["as_rank_base"] = args.as_rank_base
        if dist is not None:
            out["config"]["rank_placement"] = placement
            out["config"]["per_rank"] = per_rank
            # what this line does NOT contain: the ranks' frames are resident in HBM before the timed region, as the
            # contract asks -- no host frame fill, no upload. PARIS's own loop (fill + upload per projection, host bound at the slab
            # shape of 8 ranks: 0.55 ms of fill per 16 MiB frame) is measured at N = 1 (`paris_loop`); N such loops on one host would
            # share its memory bandwidth, which a one-GPU box cannot show
            out["config"]["frames"] = "resident in HBM on every rank before the timed region (no host fill, no upload in this line)"
            # the PMC traffic figure is of ONE kernel build: every rank reports the build it ran
            out["roofline"]["kernel_source_sha16_by_rank"] = [r["kernel_source_sha16"] for r in per_rank]
        mark("final_gather_and_report")
        if world == 1 and args.cpu_budget > 0:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_budget)
            out["cpu_baseline"]["cpu_model"] = cpu_model()
            if args.cpu_c1:
                out["cpu_baseline_c1"] = cpu_baseline_c1()
        mark("cpu_baselines")
        if ((args.workloads or args.paris_loop) and world == 1 and dist is None and args.workload == "c3" and args.slices == 0 and args.batch == 0
                and not under_profiler):
            # the headline is complete: free its 32 GiB slab and stacks, then the other configs, one child at a time
            be.free(d_vol)
            del vol, raw, work
            torch.cuda.empty_cache()
            tail = []
            for name in ("vx", "unroll", "tz", "lds_bytes", "variant", "order", "row_band", "stage_fusion", "fused_batch"):
                tail += ["--" + name.replace("_", "-"), str(getattr(args, name))]
            if args.workloads:
                out["workloads"] = other_workloads(tail)
                mark("workloads_children")
            if args.paris_loop:
                out["paris_loop"] = paris_loop()
                mark("paris_loop_children")
        mark("end")
        out["legs_seconds"] = {b[0]: round(b[1] - a[1], 3) for a, b in zip(marks, marks[1:]) if b[0] != "end" or b[1] - a[1] > 0.05}
        out["legs_seconds"]["total"] = round(marks[-1][1] - marks[0][1], 3)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    be.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
