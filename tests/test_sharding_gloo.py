"""The N > 1 path on CPU: world_size-2 gloo ranks each reconstruct their z-slab (paris_amd.sharding, the rule bench.py
uses) with the oracle standing in for the GPU kernels, then gather on rank 0 and compare with the single-rank
volume. Checks the partition (remainder on the last slab, offsets) and the barrier / max-over-ranks timing idiom."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, dim_z_override, result_path):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from paris_amd import backend as B
    from paris_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    det, odet = B.DetectorGeometry(*g), O.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    ovg = O.calculate_volume_geometry(odet)
    info = sharding.make_subvolume_info(vg, world)
    z_first, z_count = sharding.slab_of_task(info, rank)

    dist.barrier()
    t0 = time.perf_counter()
    slab = O.reconstruct(odet, ovg, 4, v_dims=(z_count, vg.dim_y, vg.dim_x), v_offset=z_first)
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)

    # final gather (the only cross-rank step the path has): ragged slabs -> padded gather on rank 0
    pad = info.geo.dim_z + info.geo.remainder
    mine = torch.zeros((pad, vg.dim_y, vg.dim_x))
    mine[:z_count] = torch.from_numpy(slab)
    parts = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank == 0:
        full = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
        for r in range(world):
            zf, zc = sharding.slab_of_task(info, r)
            full[zf:zf + zc] = parts[r][:zc].numpy()
        want = O.reconstruct(odet, ovg, 4)
        np.save(result_path, np.array([float(np.array_equal(full, want)), float(elapsed.item())]))
    dist.destroy_process_group()


def _gather_worker(rank, world, port, result_path):
    """bench.py's own final_gather / placement code (paris_amd.sharding) over gloo, slabs from the oracle"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from paris_amd import backend as B
    from paris_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    det, odet = B.DetectorGeometry(*g), O.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    ovg = O.calculate_volume_geometry(odet)
    info = sharding.make_subvolume_info(vg, world)
    z_first, z_count = sharding.slab_of_task(info, rank)
    slab = torch.from_numpy(O.reconstruct(odet, ovg, 3, v_dims=(z_count, vg.dim_y, vg.dim_x), v_offset=z_first))

    placement = sharding.gather_placement(dist, sharding.device_of_rank(rank, 0, forced=0))
    shared = sharding.check_placement(placement, exclusive=False)  # a rehearsal: all ranks on "device 0" of one host
    try:
        sharding.check_placement(placement, exclusive=True)
        exclusive_raises = False
    except RuntimeError:
        exclusive_raises = True

    light = sharding.final_gather(dist, slab, info, rank, world, full=False, on_device=False)
    full = sharding.final_gather(dist, slab, info, rank, world, full=True, on_device=False)
    if rank == 0:
        want = O.reconstruct(odet, ovg, 3)
        parts, counts = full["volume"]
        got = np.concatenate([parts[t][:counts[t]].numpy() for t in range(world)])
        sums_ok = all(abs(light["checksums"][t] - float(want[sharding.slab_of_task(info, t)[0]:][:counts[t]].sum(dtype=np.float64)))
                      <= 1e-9 * max(1.0, abs(light["checksums"][t])) for t in range(world))
        np.save(result_path, np.array([
            float(np.array_equal(got, want)), float(sums_ok), float(full["gathered_matches_checksums"]),
            float(abs(light["checksum_of_checksums"] - want.sum(dtype=np.float64)) <= 1e-9 * abs(want.sum(dtype=np.float64))),
            float(shared), float(exclusive_raises), float(len(placement)),
            float(full["gathered_bytes"] == 4.0 * vg.dim_x * vg.dim_y * (vg.dim_z - counts[0]))]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_final_gather_and_placement(tmp_path, world):
    import torch.multiprocessing as mp
    result = str(tmp_path / "g.npy")
    mp.spawn(_gather_worker, args=(world, _free_port(), result), nprocs=world, join=True)
    same, sums_ok, matches, total_ok, shared, exclusive_raises, n, bytes_ok = np.load(result)
    assert same == 1.0 and sums_ok == 1.0 and matches == 1.0 and total_ok == 1.0 and bytes_ok == 1.0
    assert shared == 1.0 and exclusive_raises == 1.0 and n == world  # every rank reported, all on one (shared) device


def _wave_worker(rank, world, port, as_world, base, result_path):
    """bench.py --as-world / --as-rank-base: `world` processes play ranks base .. base + world - 1 of an as_world-rank partition"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from paris_amd import backend as B
    from paris_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    det, odet = B.DetectorGeometry(*g), O.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    ovg = O.calculate_volume_geometry(odet)
    info = sharding.make_subvolume_info(vg, as_world)
    z_first, z_count = sharding.slab_of_task(info, base + rank)
    slab = torch.from_numpy(O.reconstruct(odet, ovg, 3, v_dims=(z_count, vg.dim_y, vg.dim_x), v_offset=z_first))
    full = sharding.final_gather(dist, slab, info, rank, world, full=True, on_device=False, task_base=base)
    if rank == 0:
        want = O.reconstruct(odet, ovg, 3)
        parts, counts = full["volume"]
        got = np.concatenate([parts[t][:counts[t]].numpy() for t in range(world)])
        first = sharding.slab_of_task(info, base)[0]
        np.save(result_path, np.array([float(np.array_equal(got, want[first:first + sum(counts)])), float(full["gathered_matches_checksums"]),
                                       float(counts == [sharding.slab_of_task(info, base + t)[1] for t in range(world)])] + full["checksums"]))
    dist.destroy_process_group()


def test_an_eight_rank_partition_in_two_waves_of_four(tmp_path):
    """The rehearsal of the 8-GPU job on a pool that admits fewer processes per card (bench.py --as-world 8 --as-rank-base 0 / 4,
    tests/test_gpu_bench_contract.py): four processes play ranks 0-3, then 4-7, of the 8-rank partition (61 slices: 7 per rank,
    the last one 12); each wave's gather assembles its half, and the eight checksums are those of the single-rank volume's blocks."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from paris_amd import sharding
    from paris_amd import backend as B
    sums = []
    for base in (0, 4):
        result = str(tmp_path / ("w%d.npy" % base))
        mp.spawn(_wave_worker, args=(4, _free_port(), 8, base, result), nprocs=4, join=True)
        r = np.load(result)
        assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0
        sums += list(r[3:])
    g = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    odet = O.DetectorGeometry(*g)
    ovg = O.calculate_volume_geometry(odet)
    want = O.reconstruct(odet, ovg, 3)
    info = sharding.make_subvolume_info(B.calculate_volume_geometry(B.DetectorGeometry(*g)), 8)
    blocks = [float(want[z0:z0 + zc].sum(dtype=np.float64)) for z0, zc in (sharding.slab_of_task(info, t) for t in range(8))]
    assert len(sums) == 8 and all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(sums, blocks))


def test_device_of_rank():
    sys.path.insert(0, ROOT)
    from paris_amd import sharding
    assert [sharding.device_of_rank(r, 8) for r in range(8)] == list(range(8))
    assert [sharding.device_of_rank(r, 1) for r in range(8)] == [0] * 8  # one visible device per rank
    assert sharding.device_of_rank(5, 4) == 1 and sharding.device_of_rank(3, 8, forced=0) == 0
    with pytest.raises(RuntimeError):
        sharding.device_of_rank(0, 0)
    a = {"rank": 0, "host": "n0", "device": 0, "visible": "0"}
    b = {"rank": 1, "host": "n0", "device": 0, "visible": "1"}
    assert sharding.check_placement([a, b]) == 2                       # same index, different visible mask: two GPUs
    with pytest.raises(RuntimeError):
        sharding.check_placement([a, dict(a, rank=1)])
    assert sharding.check_placement([dict(a, uuid="x"), dict(b, uuid="y")]) == 2


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_sharding_matches_single_rank(tmp_path, world):
    import torch.multiprocessing as mp
    result = str(tmp_path / "r.npy")
    mp.spawn(_worker, args=(world, _free_port(), None, result), nprocs=world, join=True)
    ok, elapsed = np.load(result)
    assert ok == 1.0 and elapsed > 0


def test_partition_rule():
    sys.path.insert(0, ROOT)
    from paris_amd import backend as B
    from paris_amd import sharding
    vg = B.VolumeGeometry(8, 8, 61, 1, 1, 1)
    info = sharding.make_subvolume_info(vg, 8)
    assert (info.num, info.geo.dim_z, info.geo.remainder) == (8, 7, 5)  # src/cuda/subvolume_information.cpp:112-116
    slabs = [sharding.slab_of_task(info, t) for t in range(8)]
    assert slabs[0] == (0, 7) and slabs[7] == (49, 12)                   # last slab takes the remainder
    assert sum(c for _, c in slabs) == 61
    assert sharding.tasks_of_rank(info, 1, 2) == [1, 3, 5, 7]
    one = sharding.make_subvolume_info(vg, 1)
    assert sharding.slab_of_task(one, 0) == (0, 61)
    with pytest.raises(ValueError):
        sharding.slab_of_task(info, 8)
    # BASELINE config 4: 2048 slices on 8 GPUs = 8 x 256, no remainder; a 2050-slice grid puts its 2 extra slices on rank 7
    c4 = sharding.make_subvolume_info(B.VolumeGeometry(2048, 2048, 2048, 1, 1, 1), 8)
    assert [sharding.slab_of_task(c4, t) for t in range(8)] == [(256 * t, 256) for t in range(8)] and c4.geo.remainder == 0
    odd = sharding.make_subvolume_info(B.VolumeGeometry(2048, 2048, 2050, 1, 1, 1), 8)
    slabs = [sharding.slab_of_task(odd, t) for t in range(8)]
    assert slabs[:7] == [(256 * t, 256) for t in range(7)] and slabs[7] == (1792, 258) and odd.geo.remainder == 2
    assert [sharding.tasks_of_rank(odd, r, 8) for r in range(8)] == [[r] for r in range(8)]


def _filter_shard_worker(rank, world, port, n_proj, result_path):
    """f4, second half: rank r weights + filters projections r, r + N, ... only; per group of N the ranks exchange the
    detector rows of each other's bands (paris_amd.sharding.exchange_filtered, bench.py --filter-shard); every rank then
    backprojects all frames into its slab from buffers whose rows outside its band are NaN. Oracle arithmetic, gloo."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from paris_amd import backend as B
    from paris_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    # n_proj is not a multiple of the world size: the last group is short
    det, odet = B.DetectorGeometry(*g), O.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    ovg = O.calculate_volume_geometry(odet)
    info = sharding.make_subvolume_info(vg, world)
    slabs = [sharding.slab_of_task(info, t) for t in range(world)]
    bands = [B.slab_row_band(det, vg, vg.dim_x, vg.dim_y, slabs[t][1], slabs[t][0]) for t in range(world)]
    z_first, z_count = slabs[rank]
    fs = O.filter_size(det.n_row)
    k = O.make_filter(fs, det.l_px_row)
    slab = np.zeros((z_count, vg.dim_y, vg.dim_x), np.float32)
    recv = [torch.full((det.n_col, det.n_row), float("nan")) for _ in range(world)]
    filtered_here = 0
    for j0 in range(0, n_proj, world):
        j = j0 + rank
        mine = None
        if j < n_proj:
            assert sharding.owner_of_projection(j, world) == rank
            p = O.lcg_projection(det.n_row, det.n_col, j)
            O.weight(p, odet)
            O.apply_filter(p, k, fs)
            mine = torch.from_numpy(p)
            filtered_here += 1
        for r in recv:
            r.fill_(float("nan"))
        sharding.exchange_filtered(dist, mine, recv, bands, rank, world, on_device=True)
        for q in range(world):
            if j0 + q < n_proj:
                s, c, ds, dt = O.backproject_constants(odet, j0 + q)
                O.backproject(slab, recv[q].numpy(), z_first, odet, ovg, s, c, ds, dt)
    res = sharding.final_gather(dist, torch.from_numpy(slab), info, rank, world, full=True, on_device=False)
    counts = torch.tensor([filtered_here])
    dist.all_reduce(counts)
    if rank == 0:
        want = O.reconstruct(odet, ovg, n_proj)
        parts, depth = res["volume"]
        got = np.concatenate([parts[t][:depth[t]].numpy() for t in range(world)])
        np.save(result_path, np.array([float(np.array_equal(got, want)), float(np.isfinite(got).all()), float(counts.item())]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_proj", [(2, 7), (3, 7), (8, 11)])
def test_filter_sharding_matches_unsharded(tmp_path, world, n_proj):
    """world 8 (the node the north star names): 61 slices -> 7 per rank + 5 more on rank 7; 11 projections = one full group of 8
    and a short one of 3, in which ranks 3..7 have nothing to filter but still receive"""
    import torch.multiprocessing as mp
    result = str(tmp_path / "f.npy")
    mp.spawn(_filter_shard_worker, args=(world, _free_port(), n_proj, result), nprocs=world, join=True)
    same, finite, filtered_total = np.load(result)
    assert same == 1.0 and finite == 1.0   # bit-equal to the unsharded run; no NaN row was ever read
    assert filtered_total == n_proj         # every projection was weighted and filtered exactly once across the ranks
