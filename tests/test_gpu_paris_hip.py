"""End to end through the rebuilt driver (paris_amd/host/demo/paris.hip = paris_amd/host/paris/reconstruct.h): HIS
files in a directory -> pipelined load/weight/filter/backproject per slab task -> one DDBVF file, compared with the
oracle's pipeline on the same frames (filter tolerance)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import formats as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "paris_amd", "host", "demo", "paris.hip")
KAT = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
TOL = 1e-5


def write_dataset(oracle, d, files=(3, 5), number_type=128):
    """8 LCG frames split over two HIS files plus a stray non-HIS file."""
    d.mkdir()
    i = 0
    for k, n in enumerate(files):
        fr = np.stack([oracle.lcg_projection(64, 48, i + j) for j in range(n)])
        (d / ("proj_%03d.his" % k)).write_bytes(F.his_file_bytes(fr, number_type, 32))
        i += n
    (d / "README.txt").write_text("not a projection")
    geo = d.parent / "geo.ini"
    geo.write_text("\n".join("%s = %s" % kv for kv in zip(
        ("n_row", "n_col", "l_px_row", "l_px_col", "delta_s", "delta_t", "d_so", "d_od", "delta_phi"), KAT)) + "\n")
    return geo


def run(args):
    if not os.path.exists(EXE):
        pytest.fail("%s missing: run __graft_entry__.build()" % EXE)
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    return r.stdout


def oracle_volume(oracle, idxs, phis=None, roi=None, dims=None):
    det = oracle.DetectorGeometry(*KAT)
    vg = oracle.calculate_volume_geometry(det)
    fs = oracle.filter_size(det.n_row)
    k = oracle.make_filter(fs, det.l_px_row)
    vol = np.zeros(dims or (vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    for n, i in enumerate(idxs):
        p = oracle.lcg_projection(64, 48, i)
        oracle.weight(p, det)
        oracle.apply_filter(p, k, fs)
        s, c, ds, dt = oracle.backproject_constants(det, i, phis is not None, phis[n] if phis is not None else 0.0)
        oracle.backproject(vol, p, 0, det, vg, s, c, ds, dt, roi)
    return vol


def assert_close(got, want):
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= TOL * np.abs(want).max()


@pytest.mark.parametrize("slabs", [1, 3])
def test_his_to_ddbvf(tmp_path, oracle, slabs):
    geo = write_dataset(oracle, tmp_path / "in")
    out = run(["--geometry", geo, "--input", tmp_path / "in", "--output", tmp_path / "out", "--name", "kat", "--slabs", slabs])
    assert "skipped invalid file" in out and ("%d projections" % (8 * slabs)) in out  # every task re-reads the set
    head, vol = F.ddbvf_read(str(tmp_path / "out" / "kat.ddbvf"))
    assert head == F.ddbvf_header_bytes(67, 67, 61)
    assert_close(vol, oracle_volume(oracle, range(8)))


@pytest.mark.parametrize("extra", [["--drain-chunk-kib", 40], ["--drain-chunk-kib", 1, "--slabs", 2], ["--no-row-band", "--slabs", 4],
                                   ["--batch", 1], ["--batch", 3, "--slabs", 2], ["--batch", 32], ["--batch", 64], ["--f16"],
                                   ["--slabs", 5], ["--slabs", 5, "--one-volume"], ["--slabs", 2, "--drain-chunk-kib", 1],
                                   ["--no-read-ahead"], ["--no-read-ahead", "--batch", 3, "--slabs", 3], ["--batch", 2, "--slabs", 3]])
def test_chunked_drain_and_row_band_switch(tmp_path, oracle, extra):
    """The volume goes to the file through two pinned chunks of whole slices (67 x 67 floats = 17.5 KiB per slice: 40 KiB =
    2 slices per chunk, 1 KiB = 1 slice); the detector row band (f4) can be switched off; frames are backprojected in
    groups of --batch per fused launch (default 32; 1 = one launch per projection; 3 leaves a partial last group of the 8
    frames). Same volume every way; --f16 rounds the filtered frames to half and is held to a looser bound."""
    geo = write_dataset(oracle, tmp_path / "in")
    out = run(["--geometry", geo, "--input", tmp_path / "in", "--output", tmp_path / "out", "--name", "kat"] + extra)
    # several slabs on one device: the drain thread writes slab k while slab k + 1 is reconstructed in a second volume buffer
    # (unless --one-volume keeps the reference's one buffer per device)
    if "--slabs" in extra:
        assert ("two volume buffers" in out) == ("--one-volume" not in extra), out
    head, vol = F.ddbvf_read(str(tmp_path / "out" / "kat.ddbvf"))
    assert head == F.ddbvf_header_bytes(67, 67, 61)
    want = oracle_volume(oracle, range(8))
    if "--f16" in extra:
        assert np.max(np.abs(vol - want)) <= 2e-3 * np.abs(want).max()
    else:
        assert_close(vol, want)


def test_one_thread_per_device_fan_out(tmp_path, oracle):
    """src/main.cpp:157-167: one host thread per device, all draining one task queue and writing through one sink.
    PARIS_HIP_VIRTUAL_DEVICES=3 gives the one GPU of the test box three device handles (three threads, three ctxs)."""
    geo = write_dataset(oracle, tmp_path / "in")
    env = dict(os.environ, PARIS_HIP_VIRTUAL_DEVICES="3")
    r = subprocess.run([EXE, "--geometry", geo, "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out"), "--name", "kat",
                        "--slabs", "7", "--share-frames", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("device ")]
    assert len(lines) == 3 and sum(int(l.split(":")[1].split()[0]) for l in lines) == 7  # 7 tasks over 3 threads
    # the device threads share one read-once frame source per pass over the queue (7 tasks on 3 devices: 3 passes): every one
    # of the 7 x 8 frame requests is answered, from memory when another thread of the pass has read the frame already (how many
    # depends on how the three threads interleave; tests/test_io_formats.py pins the read-once property with controlled threads)
    shared = [l for l in r.stdout.splitlines() if l.startswith("shared frame source:")]
    assert len(shared) == 1
    read, requested = int(shared[0].split()[3]), int(shared[0].split()[10])
    assert requested == 7 * 8 and 3 * 8 <= read <= 7 * 8
    head, vol = F.ddbvf_read(str(tmp_path / "out" / "kat.ddbvf"))
    assert head == F.ddbvf_header_bytes(67, 67, 61)
    assert_close(vol, oracle_volume(oracle, range(8)))
    # a stream per device thread instead (what the driver picks by itself when the slabs of a pass need little more than one
    # detector's worth of rows between them): same volume; left to itself the driver reports which of the two it took
    for extra in (["--share-frames", "0"], []):
        r = subprocess.run([EXE, "--geometry", geo, "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out1"), "--name", "kat",
                            "--slabs", "7"] + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        own = [l for l in r.stdout.splitlines() if l.startswith("frame source: one stream per device thread")]
        shared = [l for l in r.stdout.splitlines() if l.startswith("shared frame source:")]
        assert len(own) + len(shared) == 1 and (len(own) == 1 or not extra)
        _, vol = F.ddbvf_read(str(tmp_path / "out1" / "kat.ddbvf"))
        assert_close(vol, oracle_volume(oracle, range(8)))
    # the memory-driven split sees three devices: at least one slab per device (src/cuda/subvolume_information.cpp:79)
    r = subprocess.run([EXE, "--geometry", geo, "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out2")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "(3 slabs)" in r.stdout, r.stdout + r.stderr
    _, vol = F.ddbvf_read(str(tmp_path / "out2" / "vol.ddbvf"))
    assert_close(vol, oracle_volume(oracle, range(8)))


def test_roi_quality_and_angles(tmp_path, oracle):
    geo = write_dataset(oracle, tmp_path / "in", files=(8,))
    ang = tmp_path / "angles.txt"
    angles = [3.0, 50.5, 91.25, 140.0, 185.5, 230.0, 270.75, 300.0]
    ang.write_text("\n".join(str(a) for a in angles))
    run(["--geometry", geo, "--input", tmp_path / "in", "--output", tmp_path / "o2", "--quality", 2, "--angles", ang,
         "--roi", "--roi-x1", 8, "--roi-x2", 40, "--roi-y1", 4, "--roi-y2", 36, "--roi-z1", 10, "--roi-z2", 30, "--slabs", 2])
    _, vol = F.ddbvf_read(str(tmp_path / "o2" / "vol.ddbvf"))
    roi = oracle.RegionOfInterest(8, 40, 4, 36, 10, 30)
    want = oracle_volume(oracle, [0, 2, 4, 6], [angles[i] for i in (0, 2, 4, 6)], roi, (20, 32, 32))
    assert_close(vol, want)


def test_ushort_frames_and_memory_driven_split(tmp_path, oracle):
    """16-bit detector frames (the usual HIS payload) and the default slab planning (no --slabs)."""
    d = tmp_path / "in"
    d.mkdir()
    fr = np.stack([(oracle.lcg_projection(64, 48, i) * 60000).astype(np.uint16) for i in range(4)])
    (d / "scan.his").write_bytes(F.his_file_bytes(fr, 4, 32))
    geo = tmp_path / "geo.ini"
    geo.write_text("\n".join("%s = %s" % kv for kv in zip(
        ("n_row", "n_col", "l_px_row", "l_px_col", "delta_s", "delta_t", "d_so", "d_od", "delta_phi"), KAT)) + "\n")
    out = run(["--geometry", geo, "--input", d, "--output", tmp_path / "o3", "--devices", 1])
    assert "(1 slab)" in out
    _, vol = F.ddbvf_read(str(tmp_path / "o3" / "vol.ddbvf"))
    det = oracle.DetectorGeometry(*KAT)
    vg = oracle.calculate_volume_geometry(det)
    want = oracle.reconstruct(det, vg, 4, projections=[f.astype(np.float32) for f in fr])
    assert_close(vol, want)


def test_missing_geometry_key_fails_like_the_reference(tmp_path):
    geo = tmp_path / "geo.ini"
    geo.write_text("n_row = 64\n")
    r = subprocess.run([EXE, "--geometry", str(geo), "--input", str(tmp_path), "--output", str(tmp_path / "o")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "required but missing" in r.stderr


def _file_digest(path):
    """xxh3-128 of a (large) file, read in 64 MiB pieces"""
    import xxhash
    h = xxhash.xxh3_128()
    with open(path, "rb") as f:
        while True:
            b = f.read(64 << 20)
            if not b:
                break
            h.update(b)
    return h.hexdigest(), os.path.getsize(path)


def test_eight_device_threads_at_the_config4_partition(tmp_path, oracle):
    """the driver's 8-device job at full width -- a 2048 x 2048 HIS set (16-bit frames, 16 projections), the
    natural 2048 x 2048 x 2090 volume, PARIS_HIP_VIRTUAL_DEVICES=8 on the one GPU: the memory planner hands every device one slab
    (8 slabs of 261 slices, the last one with the remainder: src/cuda/subvolume_information.cpp:112-116), eight feed / device /
    drain thread triples (src/main.cpp:157-167) reconstruct them side by side in one process and one sink assembles the 35 GB
    DDBVF file. Byte for byte the file of the one-device run of the same set (which is checked against the oracle on a crop).
    Falls back to a 2048 x 1024 detector (17 GB file) when the scratch disk is short."""
    import shutil
    n_row, n_col = 2048, 2048
    free = shutil.disk_usage(tmp_path).free  # (one volume file exists at a time: each is digested and removed)
    if free < 45 * 2 ** 30:
        n_col = 1024
    if free < 25 * 2 ** 30:
        pytest.skip("scratch disk too small for a full-width volume (%.0f GiB free)" % (free / 2 ** 30))
    n_proj = 16
    d = tmp_path / "in"
    d.mkdir()
    frames = [(oracle.lcg_projection(n_row, n_col, i) * 60000).astype(np.uint16) for i in range(n_proj)]
    for k in range(2):
        (d / ("scan_%d.his" % k)).write_bytes(F.his_file_bytes(np.stack(frames[8 * k:8 * k + 8]), 4, 32))
    g = (n_row, n_col, 0.2, 0.2, 0.0, 0.0, 500, 500, 360.0 / n_proj)
    geo = tmp_path / "geo.ini"
    geo.write_text("\n".join("%s = %s" % kv for kv in zip(
        ("n_row", "n_col", "l_px_row", "l_px_col", "delta_s", "delta_t", "d_so", "d_od", "delta_phi"), g)) + "\n")
    det = oracle.DetectorGeometry(*g)
    vg = oracle.calculate_volume_geometry(det)

    def reconstruct(devices, out_dir, slabs):
        env = dict(os.environ, PARIS_HIP_VIRTUAL_DEVICES=str(devices)) if devices > 1 else dict(os.environ)
        extra = ["--slabs", str(slabs)] if slabs else []  # (left to itself the driver pipelines several slabs per device)
        r = subprocess.run([EXE, "--geometry", str(geo), "--input", str(d), "--output", str(out_dir)] + extra, capture_output=True,
                           text=True, timeout=1500, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout, os.path.join(str(out_dir), "vol.ddbvf")

    out1, f1 = reconstruct(1, tmp_path / "o1", 1)
    assert "(1 slab)" in out1
    # the one-device file against the oracle: 12 slices around the mid-plane and 6 at the top of the volume
    head = F.ddbvf_header_bytes(vg.dim_x, vg.dim_y, vg.dim_z)
    plane = vg.dim_x * vg.dim_y
    fs = oracle.filter_size(n_row)
    k = oracle.make_filter(fs, det.l_px_row)
    filtered = []
    for i in range(n_proj):
        p = frames[i].astype(np.float32)
        oracle.weight(p, det)
        oracle.apply_filter(p, k, fs)
        filtered.append(p)
    with open(f1, "rb") as fh:
        assert fh.read(len(head)) == head
        for z0, cnt in ((vg.dim_z // 2 - 6, 12), (vg.dim_z - 6, 6)):
            fh.seek(len(head) + 4 * plane * z0)
            got = np.frombuffer(fh.read(4 * plane * cnt), np.float32).reshape(cnt, vg.dim_y, vg.dim_x)
            want = np.zeros((cnt, vg.dim_y, vg.dim_x), np.float32)
            for i in range(n_proj):
                s, c, ds, dt = oracle.backproject_constants(det, i)
                oracle.backproject(want, filtered[i], z0, det, vg, s, c, ds, dt)
            assert np.max(np.abs(got - want)) <= TOL * np.abs(want).max()
    d1 = _file_digest(f1)
    os.unlink(f1)
    assert d1[1] == len(head) + 4 * plane * vg.dim_z
    # the exact partition of BASELINE config 4 -- one slab per device -- and the driver's own plan (the memory planner sees eight
    # devices and pipelines several slabs per device: a second volume buffer per device, drained while the next slab is computed)
    for slabs, tag in ((8, "o8"), (0, "o8p")):
        out8, f8 = reconstruct(8, tmp_path / tag, slabs)
        n_slabs = int(out8.split("(")[1].split()[0])
        assert (n_slabs == 8) if slabs else (n_slabs >= 8 and n_slabs % 8 == 0), out8[:300]
        lines = [l for l in out8.splitlines() if l.startswith("device ")]
        assert len(lines) == 8 and sum(int(l.split(":")[1].split()[0]) for l in lines) == n_slabs  # eight device threads share the slab tasks
        d8 = _file_digest(f8)
        os.unlink(f8)
        assert d8 == d1
