"""bench.py's own launcher for N > 1, as far as a box without a GPU can show it: the placement refusal over
nccl, and that the gloo rehearsal path starts the ranks as child processes and relays their failure (no GPU here)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _gpus_visible():
    import torch
    return torch.cuda.device_count()


def test_nccl_with_fewer_gpus_than_ranks_is_refused_at_once():
    n = _gpus_visible() + 2
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "c1"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=ENV)
    assert r.returncode == 2 and r.stdout == ""
    assert "needs one GPU per rank" in r.stderr and "nothing was run" in r.stderr
    assert time.perf_counter() - t0 < 120.0


def test_launcher_relays_the_ranks_failure():
    if _gpus_visible() > 0:
        import pytest
        pytest.skip("a GPU is visible: the ranks would run (tests/test_gpu_bench_contract.py covers that)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                        "--workload", "c1"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=ENV)
    assert r.returncode != 0 and r.stdout == ""
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr and "--max-restarts 0" in r.stderr
    assert "no GPU visible" in r.stderr  # each rank's own message: the hot path has no CPU fallback


def test_launcher_relays_one_line_and_does_not_wait_for_ranks_that_hang_after_it(capsys):
    """The launcher's own logic with stand-in children: (1) a child that prints noise and one JSON line and exits -- the line alone
    reaches stdout, status 0; (2) a child that prints its line and then never exits (ranks stuck while the process group is torn
    down): terminated after the grace period, the line is kept, status 0; (3) a child that dies without a line: its status."""
    import argparse
    import sys as _sys
    _sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(gpus=2, dist_backend="gloo")
    py = sys.executable
    rc = bench.launch_ranks(args, [], _cmd=[py, "-c", "print('NCCL version banner'); print('{\"n_gpus\": 2}'); print('bye')"])
    out = capsys.readouterr()
    assert rc == 0 and out.out == '{"n_gpus": 2}\n' and "NCCL version banner" in out.err
    t0 = time.perf_counter()
    rc = bench.launch_ranks(args, [], grace=2.0, _cmd=[py, "-c", "import time; print('{\"n_gpus\": 2}', flush=True); time.sleep(600)"])
    out = capsys.readouterr()
    assert rc == 0 and out.out == '{"n_gpus": 2}\n' and "terminating" in out.err and time.perf_counter() - t0 < 60.0
    rc = bench.launch_ranks(args, [], _cmd=[py, "-c", "import sys; sys.exit(7)"])
    out = capsys.readouterr()
    assert rc == 7 and out.out == ""
