"""bench.py's output contract on a small workload (N = 1), and a rehearsal of its N = 2 path with both ranks sharing
the one GPU of the test box over gloo (the real multi-GPU run uses RCCL, one rank per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text  # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "3", "--warmup", "1",
                        "--cpu-budget", "2", "--fused-steps", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    d = last_json_line(r.stdout)
    for k in REQUIRED + ["cpu_baseline", "cpu_baseline_c1", "fused_extension"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "GVoxel-updates/s" and d["value"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # default batch: the timed steps are the whole job (360 projections of config 1 in 3 steps), the full circle
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches_timed"] == 360
    cfg = d["config"]
    assert cfg["projections_per_step"] == 120 and cfg["projections_timed"] == 360 and cfg["whole_job"] is True
    assert cfg["angles_covered"]["distinct_projections"] == 360 and cfg["angles_covered"]["of"] == 360
    octs = cfg["backproject_kernel_ms_by_octant"]
    assert len(octs) == 8 and sum(o["launches"] for o in octs) == 360
    assert all(o["min_ms"] <= o["mean_ms"] <= o["max_ms"] for o in octs)
    fr = d["fused_extension"]["roofline"]
    assert fr is None or (fr["bound"] == "valu_issue" and abs(fr["frac"] - fr["achieved"] / fr["peak"]) < 1e-12)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["cpu_model"]
    # the thread binding the leg ran under is on the line (unbound unless the caller's environment binds it)
    assert cb["omp"] == {"proc_bind": os.environ.get("OMP_PROC_BIND"), "places": os.environ.get("OMP_PLACES")}
    assert abs(cb["per_core"] * cb["cores"] - cb["value"]) < 1e-9 * cb["value"]
    # the skip-off leg: one more step (120 launches here) with every tile read and written; the DRAM-side fraction needs a
    # committed PMC profile of this workload (none for config 1: null)
    assert rf["without_skip_launches"] == 120 and rf["frac_without_skip"] > 0 and rf["kernel_ms_without_skip"] > 0
    assert "frac_dram" in rf and (rf["frac_dram"] is None) == (rf["traffic"] is None)
    # the whole config-1 job on the host's cores, backprojection timed alone
    c1 = d["cpu_baseline_c1"]
    assert c1["kind"] == "port" and c1["cores"] >= 1 and c1["value"] > 0 and c1["backproject_s"] > 0 and c1["cpu_model"]
    assert abs(c1["value"] - 256.0 ** 3 * 360 / c1["backproject_s"] / 1e9) < 1e-9 * c1["value"]
    # where the invocation's wall time went, leg by leg
    legs = d["legs_seconds"]
    assert legs["total"] > legs["timed_region"] > 0 and "cpu_baselines" in legs and "setup_and_warmup" in legs
    assert abs(sum(v for k, v in legs.items() if k != "total") - legs["total"]) < 0.2


def test_two_rank_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "c1", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--device", "0", "--fused-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["slab_per_gpu"] == [256, 256, 128]  # 256 slices split over two ranks
    assert "cpu_baseline" not in d                          # rank 0 at N = 1 only
    fg = d["final_gather"]                                  # the job's one collective, here over gloo
    assert fg["rccl_ranks_seen"] == 2 and fg["mode"] == "checksums" and len(fg["slab_checksums"]) == 2
    assert abs(fg["checksum_of_checksums"] - sum(fg["slab_checksums"])) <= 1e-9 * max(1.0, abs(fg["checksum_of_checksums"]))
    assert [e["rank"] for e in d["config"]["rank_placement"]] == [0, 1]
    # every rank reports its own kernel statistics and the kernel build it ran
    pr = d["config"]["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1] and [e["slab"] for e in pr] == [[0, 128], [128, 128]]
    assert all(e["launches"] == 360 and 0 < e["kernel_ms_min"] <= e["kernel_ms_mean"] <= e["kernel_ms_max"] for e in pr)
    assert len(set(d["roofline"]["kernel_source_sha16_by_rank"])) == 1 and len(set(e["library_sha16"] for e in pr)) == 1


def test_four_rank_rehearsal_with_a_ragged_split():
    """Four ranks on the one GPU (gloo), a volume whose 256 slices split evenly but whose 7-projection batch does not divide by
    the rank count: every rank reports its own slab and statistics, the sharded filter and the plain run agree."""
    def run(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload",
                            "c1", "--steps", "1", "--warmup", "0", "--batch", "7", "--dist-backend", "gloo", "--device", "0",
                            "--fused-steps", "0"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return last_json_line(r.stdout)
    plain, sharded = run([]), run(["--filter-shard", "1"])
    pr = plain["config"]["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1, 2, 3] and [e["slab"] for e in pr] == [[64 * r, 64] for r in range(4)]
    assert plain["n_gpus"] == 4 and plain["config"]["slab_per_gpu"] == [256, 256, 64]
    assert plain["final_gather"]["rccl_ranks_seen"] == 4 and len(plain["final_gather"]["slab_checksums"]) == 4
    assert sharded["final_gather"]["slab_checksums"] == plain["final_gather"]["slab_checksums"]


def test_two_rank_rehearsal_gathers_slabs():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "c1", "--steps", "1", "--warmup", "0", "--batch", "4", "--dist-backend", "gloo", "--device", "0",
                        "--fused-steps", "0", "--final-gather", "slabs"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json_line(r.stdout)
    fg = d["final_gather"]
    assert fg["mode"] == "slabs" and fg["gathered_matches_checksums"] is True
    assert fg["gathered_bytes"] == 4.0 * 256 * 256 * 128 and d["config"]["whole_job"] is False


def test_two_rank_rehearsal_filter_sharding():
    """f4, second half, through bench.py itself: two ranks on the one GPU (gloo), each filters every other projection, bands
    are exchanged, the slab checksums equal those of the unsharded two-rank run (same projections, same volume)."""
    def run(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                            "c1", "--steps", "1", "--warmup", "0", "--batch", "7", "--dist-backend", "gloo", "--device", "0",
                            "--fused-steps", "0"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return last_json_line(r.stdout)
    plain, sharded = run([]), run(["--filter-shard", "1"])
    assert sharded["config"]["filter_shard"] is True and plain["config"]["filter_shard"] is False
    # every rank holds the same synthetic stack, band rows filtered alone equal the same rows of a whole-frame filter bit for bit
    # and the backprojection is bit-exact: the two runs must produce identical slabs
    assert sharded["final_gather"]["slab_checksums"] == plain["final_gather"]["slab_checksums"]
    assert len(sharded["final_gather"]["slab_checksums"]) == 2 and sharded["value"] > 0
    assert sharded["roofline"]["launches_timed"] == 7


def test_one_rank_under_torchrun_with_rccl_prints_only_the_json_line():
    """The driver launches N > 1 through torch.distributed.run with the nccl (= RCCL) backend; RCCL prints a version banner on
    stdout from native code when the communicator is created. bench.py must keep stdout to its ONE JSON line (everything
    else goes to stderr). One rank on the one GPU exercises the same code: process group, placement gathering, barrier,
    all-reduce of the elapsed time."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload",
                        "c1", "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--fused-steps", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["rank_placement"][0]["rank"] == 0 and "final_gather" not in d
    # the same job without the launcher gives the same figure (config 1 is launch bound: the per-launch kernel
    # time within 10 %, the whole-step value within 25 % -- the spread of two runs on one box)
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--steps", "2", "--warmup", "1",
                            "--cpu-budget", "0", "--cpu-c1", "0", "--fused-steps", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")})
    assert plain.returncode == 0, plain.stderr[-2000:]
    q = last_json_line(plain.stdout)
    assert abs(q["config"]["backproject_kernel_ms"] / d["config"]["backproject_kernel_ms"] - 1.0) < 0.10
    assert abs(q["value"] / d["value"] - 1.0) < 0.25
    # the job's one collective on the nccl (= RCCL) backend, which the gloo rehearsals cannot vouch for --
    # dist.gather of the slab and all_gather_into_tensor of float64 checksums, at N = 1 (rank 0 gathers its own 64 MiB slab)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload",
                        "c1", "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--fused-steps", "0", "--final-gather", "slabs"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    fg = json.loads(lines[0])["final_gather"]
    assert "error" not in fg, fg
    assert fg["backend"] == "rccl" and fg["mode"] == "slabs" and fg["rccl_ranks_seen"] == 1
    assert fg["gathered_matches_checksums"] is True and len(fg["slab_checksums"]) == 1 and fg["slab_checksums"][0] != 0.0


def test_plain_command_line_starts_its_own_ranks():
    """`python3 bench.py --gpus N` with N > 1 and no RANK in the environment -- the way the driver calls the
    N = 1 line -- runs the job itself: torch.distributed.run as a child process, rank 0's one JSON line relayed. Two gloo ranks on
    the one GPU here; on a multi-GPU node the same command line runs over RCCL, one rank per GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                        "--workload", "c1", "--steps", "2", "--warmup", "1", "--fused-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # stdout carries the one line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["final_gather"]["rccl_ranks_seen"] == 2
    assert [e["rank"] for e in d["config"]["per_rank"]] == [0, 1]


def test_plain_command_line_over_rccl_needs_a_gpu_per_rank():
    """... and over nccl (= RCCL) with fewer GPUs than ranks it ends at once with one clear line and a non-zero status: no
    rank is started, no two ranks silently share a card."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs one GPU per rank" in r.stderr and time.perf_counter() - t0 < 120.0
    # the torchrun-started path says the same (every rank, before the rendezvous)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "need one GPU each" in r.stderr


def test_config3_line_measures_its_traffic_in_the_run():
    """The driver's own invocation is config 3 at N = 1: before it touches the GPU, bench.py runs child passes of itself under
    `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: one counter per pass) and reports the HBM bytes per launch and the
    fused kernel's instructions per voxel-update measured there. A short job here (4 timed projections); the counter passes are
    the real ones."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--batch", "4", "--spread", "1",
                        "--cpu-budget", "0", "--cpu-c1", "0", "--fused-steps", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json_line(r.stdout)
    rf = d["roofline"]
    assert rf["traffic_measured_in_this_run"] is True and rf["traffic_is_of_this_kernel"] is True
    # tiles no ray reaches are not moved: somewhat under the algorithmic bytes, never above them by more than counter noise
    assert 0.85 < rf["traffic_over_algorithmic"] < 1.01
    assert abs(rf["frac_dram"] - rf["traffic"] / (d["config"]["backproject_kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert d["config"]["whole_job"] is False and rf["launches_timed"] == 4
    fr = d["fused_extension"]["roofline"]
    assert fr["source"].startswith("measured in this run") and 15.0 < fr["valu_instructions_per_voxel_update"] < 40.0


def _torchrun(n, extra, timeout=1500, plain=False):
    """N ranks of bench.py: started by torch.distributed.run (the driver's way for N > 1), or -- plain=True -- by bench.py's own
    launcher from the plain command line (`python bench.py --gpus N ...`, no RANK in the environment)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if plain:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra, capture_output=True, text=True,
                           timeout=timeout, cwd=ROOT, env=env)
    else:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra,
                           capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return last_json_line(r.stdout)


@pytest.mark.parametrize("workload,gather", [("c3", "slabs"), ("c5", "checksums")])
def test_eight_slab_partition_at_full_shape(workload, gather):
    """the exact partition of BASELINE configs 4 / 5 -- eight z-slabs of 2048 x 2048 x 256 (config 5: of the
    2048^3 ROI of the 4096^3 grid, half-precision projections) -- on the one GPU. The pool admits at most six processes on a card
    (gpurun's process guard), so the eight ranks' work runs as two waves of four gloo ranks (--as-world 8 --as-rank-base 0 / 4):
    every process is ONE rank of the 8-rank job -- its slab, its slab offset, its detector row band, the full 4 GiB slab in HBM
    next to the projection stack -- and the job's one collective runs among the four (config 3: the four 4 GiB slabs gathered on
    rank 0 and re-checksummed). The eight slab checksums must be the float64 sums of the eight 256-slice blocks of an N = 1 run
    of the same 16 projections (bit-identical slabs sum to identical doubles)."""
    common = ["--workload", workload, "--steps", "2", "--warmup", "0", "--batch", "8", "--fused-steps", "1", "--cpu-budget", "0",
              "--cpu-c1", "0", "--noskip-step", "0", "--live-traffic", "0"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--block-checksums", "8"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    want = last_json_line(one.stdout)["block_checksums"]["sums"]
    assert len(want) == 8 and all(abs(v) > 0 for v in want)
    got, bands = [], []
    for base in (0, 4):
        # (the second wave through bench.py's own launcher: the plain command line at full shape)
        d = _torchrun(4, common + ["--dist-backend", "gloo", "--device", "0", "--final-gather", gather, "--as-world", "8",
                                   "--as-rank-base", str(base)], plain=(base == 4))
        pr = d["config"]["per_rank"]
        assert [e["as_rank"] for e in pr] == [base + r for r in range(4)]
        assert [e["slab"] for e in pr] == [[256 * (base + r), 256] for r in range(4)]
        assert d["config"]["slab_per_gpu"] == [2048, 2048, 256] and d["config"]["as_world"] == 8
        assert all(e["launches"] == 16 for e in pr)
        fg = d["final_gather"]
        assert fg["rccl_ranks_seen"] == 4 and fg["mode"] == gather and len(fg["slab_checksums"]) == 4
        if gather == "slabs":
            assert fg["gathered_matches_checksums"] is True and fg["gathered_bytes"] == 3.0 * 4.0 * 2048 * 2048 * 256
        got += fg["slab_checksums"]
        bands += [tuple(e["detector_row_band"]) for e in pr]
    assert got == want
    # the slabs see different detector bands, higher slabs higher rows, mirrored about the mid-plane; config 3's outermost slabs
    # reach the detector's first and last row (config 5's ROI is the middle of the grid: its bands stay inside)
    assert all(bands[t][0] <= bands[t + 1][0] and sum(bands[t]) <= sum(bands[t + 1]) for t in range(7)) and all(0 < b[1] < 2048 for b in bands)
    assert all(bands[t][1] == bands[7 - t][1] and bands[t][0] + bands[t][1] == 2048 - bands[7 - t][0] for t in range(4))
    if workload == "c3":
        assert bands[0][0] == 0 and bands[7][0] + bands[7][1] == 2048


def test_default_run_reports_every_baseline_config():
    """the driver's invocation (config 3, whole job, N = 1) ends with a `workloads` object -- config 1 and 2 as
    whole jobs, config 4's slab shape as a whole job, config 5 sampled over the circle -- each with the headline's figures, from
    fresh child runs -- and with `paris_loop`, PARIS's per-projection loop through paris::hip for four jobs. (CPU legs and the live
    counter passes are switched off here: they have tests of their own.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--cpu-budget", "0", "--cpu-c1", "0",
                        "--live-traffic", "0", "--fused-steps", "1", "--noskip-step", "0"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json_line(r.stdout)
    assert d["config"]["whole_job"] is True and d["config"]["projections_timed"] == 1440
    wl = d["workloads"]
    assert sorted(wl) == ["c1", "c1_hip_graph", "c2", "c4_slab_shape", "c5_sampled", "c5_uncropped_slab"]
    for name, e in wl.items():
        assert "error" not in e, (name, e)
        for key in ("value", "backproject_kernel_ms", "fused", "deferred", "ms_per_step", "projections_timed", "whole_job"):
            assert e[key] is not None, (name, key)
        assert e["value"] > 0 and e["fused"] > e["value"] and 0.3 < e["roofline"]["frac"] < 1.0
        assert e["roofline"]["frac_without_skip"] is not None
    assert wl["c1_hip_graph"]["whole_job"] and wl["c1_hip_graph"]["value"] > 0.9 * wl["c1"]["value"]
    assert wl["c1"]["whole_job"] and wl["c2"]["whole_job"] and wl["c4_slab_shape"]["whole_job"] and not wl["c5_sampled"]["whole_job"]
    assert wl["c1"]["projections_timed"] == 360 and wl["c2"]["projections_timed"] == 720 and wl["c4_slab_shape"]["projections_timed"] == 1440
    assert wl["c5_sampled"]["projections_timed"] == 360 and wl["c5_sampled"]["dtype"] == "f16-in/f32"
    # one rank's slab of config 5 without the ROI crop -- 4096 x 4096 x 512 of the 4096^3 grid, 32 GiB
    cu = wl["c5_uncropped_slab"]
    assert cu["slab"] == [4096, 4096, 512] and cu["dtype"] == "f16-in/f32" and cu["projections_timed"] == 72 and not cu["whole_job"]
    assert "frac_of_cache_resident_rate" in wl["c1"]["roofline"]
    assert "2048, 2048, 256" in str(wl["c4_slab_shape"]["workload"]) or "256 slices" in wl["c4_slab_shape"]["workload"]
    # PARIS's own per-projection loop through the C++ mirror (paris_hip_demo child processes), whole circles
    pl = d["paris_loop"]
    assert len(pl) == 4 and any("config1" in k for k in pl) and any("2048^2" in k for k in pl)
    for key, e in pl.items():
        assert "error" not in e, (key, e)
        assert e["value"] > 0 and e["seconds"] > 0 and e["unit"] == "GVoxel-updates/s" and e["deferral"] == 48
        assert e["by_reference"] == 1 and e["filter_deferral"] == 2   # the mirror's defaults since round 5
        assert 0.0 <= e["host_fill_share"] < 1.0 and 0.0 < e["backend_call_share"] < 1.0 and "second" in e["streams"]
        assert abs(e["value"] - float(e["volume"][0]) * e["volume"][1] * e["volume"][2] * e["projections"] / e["seconds"] / 1e9) < 1e-3 * e["value"]
        assert sorted(e["us_per_projection"]) == ["backproject", "filter", "frame_fill", "free_device", "free_host", "load",
                                                   "make_projection_host", "weight"]
    assert [e["volume"] for k, e in pl.items() if "config1" in k] == [[256, 256, 256]]
