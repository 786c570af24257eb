"""The C++ host mirror (paris_amd/host: namespace paris::hip + stage wrappers) driven like PARIS's per-device loop
(src/main.cpp:79-109) by paris_amd/host/demo/paris_hip_demo, compared with the golden volume."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "paris_amd", "host", "demo", "paris_hip_demo")
KAT_ARGS = ["64", "48", "0.2", "0.25", "1.5", "-0.75", "100", "200", "45", "8"]


def run_demo(tmp_path, in_spec, extra, exe=DEMO):
    out = tmp_path / "vol.raw"
    if not os.path.exists(exe):
        pytest.fail("%s missing: run __graft_entry__.build()" % exe)
    r = subprocess.run([exe] + KAT_ARGS + [in_spec, str(out)] + extra, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    dims = [int(x) for x in r.stdout.split()[1:4]]
    return np.fromfile(out, np.float32).reshape(dims[2], dims[1], dims[0])


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "kat.npz"))


@pytest.mark.parametrize("extra,crop", [
    ([], (slice(None), slice(None), slice(None))),
    (["--slabs", "3"], (slice(None), slice(None), slice(None))),          # 20 + 20 + 21 slices, offsets 0/20/40
    (["--roi", "8", "40", "4", "36", "10", "30"], (slice(10, 30), slice(4, 36), slice(8, 40))),
    (["--roi", "8", "40", "4", "36", "10", "30", "--slabs", "2"], (slice(10, 30), slice(4, 36), slice(8, 40))),
])
def test_cpp_loop_backprojection_bit_exact(tmp_path, gold, extra, crop):
    f = tmp_path / "filtered.raw"
    gold["filtered"].astype(np.float32).tofile(f)
    got = run_demo(tmp_path, str(f), ["--no-weight", "--no-filter"] + extra)
    want = gold["volume"][crop]
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cpp_loop_one_launch_per_call(tmp_path, gold):
    """The same loop built with PARIS_HIP_BACKPROJECT_DEFERRAL=1 (paris_hip_demo_immediate): asynchronous calls, recycled
    buffers, one backprojection launch per projection."""
    f = tmp_path / "filtered.raw"
    gold["filtered"].astype(np.float32).tofile(f)
    got = run_demo(tmp_path, str(f), ["--no-weight", "--no-filter", "--slabs", "3"], DEMO + "_immediate")
    assert np.array_equal(got.view(np.uint32), gold["volume"].view(np.uint32))


@pytest.mark.parametrize("exe", [DEMO, DEMO + "_immediate"])
def test_cpp_loop_many_projections_rotating_buffers(tmp_path, oracle, exe):
    """40 projections of 200 x 160 through the asynchronous loop: the deferral ring is launched after 8, then 16 more calls (8 + 16 + 16 at
    read-back), released host / device projection buffers rotate through the library's pools while earlier uploads and
    launches are still in flight. Bit-identical to the oracle's backprojection of the same frames."""
    n_row, n_col, n_proj = 200, 160, 40
    args = [str(n_row), str(n_col), "0.2", "0.2", "0.5", "-0.25", "300", "200", "9", str(n_proj)]
    frames = np.stack([oracle.lcg_projection(n_row, n_col, i) - np.float32(0.5) for i in range(n_proj)])
    f = tmp_path / "in.raw"
    frames.tofile(f)
    out = tmp_path / "vol.raw"
    r = subprocess.run([exe] + args + [str(f), str(out), "--no-weight", "--no-filter", "--slabs", "2"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    dims = [int(x) for x in r.stdout.split()[1:4]]
    got = np.fromfile(out, np.float32).reshape(dims[2], dims[1], dims[0])
    det = oracle.DetectorGeometry(n_row, n_col, 0.2, 0.2, 0.5, -0.25, 300, 200, 9)
    vg = oracle.calculate_volume_geometry(det)
    assert (vg.dim_x, vg.dim_y, vg.dim_z) == tuple(dims)
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    for i in range(n_proj):
        s, c, ds, dt = oracle.backproject_constants(det, i)
        oracle.backproject(want, frames[i], 0, det, vg, s, c, ds, dt)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cpp_loop_full_pipeline(tmp_path, gold):
    got = run_demo(tmp_path, "lcg", ["--slabs", "2"])
    want = gold["volume"]
    assert np.max(np.abs(got - want)) <= 1e-5 * np.abs(want).max()


def test_cpp_loop_with_filter_deferral_is_bit_identical(tmp_path):
    """PARIS_HIP_FILTER_DEFERRAL=1 (paris_hip_demo_filter_deferral): through the C++ mirror's unchanged load / weight / filter /
    backproject loop the filter() of every projection is held back with its weight() and runs on the library's snapshots, a group
    per launch. A 512-pixel detector row (1024-point filter: the fused weight + filter kernel), 60 projections (one full group of
    48 and a partial one), two slabs: the volume equals the default build's bit for bit."""
    args = ["512", "24", "0.2", "0.2", "0.5", "-0.25", "300", "200", "6", "60"]
    vols = []
    for exe in (DEMO, DEMO + "_filter_deferral"):
        if not os.path.exists(exe):
            pytest.fail("%s missing: run __graft_entry__.build()" % exe)
        out = tmp_path / (os.path.basename(exe) + ".raw")
        r = subprocess.run([exe] + args + ["lcg", str(out), "--slabs", "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        vols.append(np.fromfile(out, np.float32))
    assert vols[0].size > 0 and np.abs(vols[0]).max() > 0
    assert np.array_equal(vols[0].view(np.uint32), vols[1].view(np.uint32))


def test_cpp_loop_second_stream_equals_one_stream_and_one_launch_per_call(tmp_path):
    """paris::hip now runs a full group's fused launch on the ctx's second stream and copy_h2d(projection) on
    the upload stream, and released host / device projection buffers come back from the pools as soon as THEIR last user has
    finished. PARIS's unchanged loop over 150 projections of 320 x 256 (1024-point fused weight + filter; three full groups of 48
    and a partial one), two slabs: the default build, the one-stream build (paris_hip_demo_serial) and one launch per call
    (paris_hip_demo_immediate) write the same volume bit for bit."""
    args = ["320", "256", "0.2", "0.2", "0.5", "-0.25", "300", "200", "2.4", "150"]
    vols = []
    # (round 5: the default build takes its projections BY REFERENCE, filters them in place a group at a time and validates
    # asynchronously; paris_hip_demo_snapshots is round 4's form -- a snapshot per call --, paris_hip_demo_filter_at_once references
    # with a filter launch per projection)
    for exe in (DEMO, DEMO + "_serial", DEMO + "_immediate", DEMO + "_snapshots", DEMO + "_filter_at_once"):
        if not os.path.exists(exe):
            pytest.fail("%s missing: run __graft_entry__.build()" % exe)
        out = tmp_path / (os.path.basename(exe) + ".raw")
        r = subprocess.run([exe] + args + ["lcg", str(out), "--slabs", "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        vols.append(np.fromfile(out, np.float32))
        os.unlink(out)
    assert vols[0].size > 0 and np.abs(vols[0]).max() > 0
    for other in vols[1:]:
        assert np.array_equal(vols[0].view(np.uint32), other.view(np.uint32))


def test_cpp_loop_with_the_host_ahead_of_the_device(tmp_path):
    """The pools' blocking path: small frames (512 x 512: the host supplies one in ~70 us) into a large volume (1024 x 1024 x 512:
    the device needs ~270 us per projection), so the host runs ahead until every buffer of the rotation is parked and busy and
    paris_hip_malloc_* waits for the oldest one -- its H2D copy for a pinned buffer, its snapshot for a device buffer -- while
    fused launches run on the second stream and the ring's halves alternate. 130 projections (8 + 16 + 32 + 48 + 26). The
    volume equals the one-launch-per-call build's bit for bit."""
    args = ["512", "512", "0.2", "0.2", "0", "0", "500", "500", "2.769", "130"]
    vols = []
    for exe in (DEMO, DEMO + "_immediate"):
        if not os.path.exists(exe):
            pytest.fail("%s missing: run __graft_entry__.build()" % exe)
        out = tmp_path / (os.path.basename(exe) + ".raw")
        r = subprocess.run([exe] + args + ["lcg", str(out), "--vol", "1024", "1024", "512", "0.05"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        vols.append(np.fromfile(out, np.float32))
        os.unlink(out)
    assert vols[0].size == 1024 * 1024 * 512 and np.abs(vols[0]).max() > 0
    assert np.array_equal(vols[0].view(np.uint32), vols[1].view(np.uint32))


def test_flush_rules_at_config3_size():
    """the deferred boundary (48 backproject() calls per fused launch; a sequence's first groups after 8, 16, 32 calls) and the held-back weight() through the
    C++ mirror paris::hip on the 2048^2 detector / 2048^3 grid of BASELINE config 3, with every observer that must flush
    interleaved (copy_d2h in the middle of a group, calls alternating between two slabs, make / free of other buffers,
    synchronize, a projection read between weight and filter): every read-back equals the one-launch-per-call run bit for bit;
    round 3: once more with paris_hip_set_backproject_overlap(1) (fused launches on a second stream that every observer joins)."""
    exe = os.path.join(ROOT, "paris_amd", "host", "demo", "paris_hip_flush_rules")
    if not os.path.exists(exe):
        pytest.fail("%s missing: run __graft_entry__.build()" % exe)
    r = subprocess.run([exe, "64", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # four read-backs of the deferred run + three of the run with the fused launches on the ctx's second stream
    assert "flush rules ok" in r.stdout and r.stdout.count("equal bit for bit") == 7 and r.stdout.count("second stream") == 3


def test_cpp_loop_device_memory_is_bounded_by_the_rotation(tmp_path):
    """ADVICE r04 / deferral by reference: however many projections PARIS's loop pushes through paris::hip, what the device holds
    afterwards is the library's bounded rotation of projection buffers (two groups of 48 and a few, here 4 MiB frames) and its
    tables -- the same after 300 projections as after 2400 (paris_hip_demo --json reports the bytes in use once the loops are over
    and the volume has been released)."""
    import json
    seen = []
    for n_proj in (300, 2400):
        r = subprocess.run([DEMO, "1024", "1024", "0.2", "0.2", "0", "0", "500", "500", "0.15", str(n_proj), "lcg", "/dev/null", "--cycle", "16",
                            "--no-out", "--json", "--vol", "512", "512", "256", "0.2"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["by_reference"] == 1 and d["projections"] == n_proj
        seen.append(d["device_bytes_in_use_after_the_loops"])
    frame = 1024 * 1024 * 4
    assert abs(seen[1] - seen[0]) <= 16 * frame, seen          # no growth with the number of projections
    assert seen[1] <= (2 << 30), seen                          # runtime + tables + at most 104 parked frames of 4 MiB
