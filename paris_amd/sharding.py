"""z-slab decomposition of the output volume across devices / ranks.

The reference splits dim_z into `num` equal slabs and gives the remainder to the last one
(src/cuda/subvolume_information.cpp:112-116, src/make_volume.cpp:32-34); slab `id` starts at global slice
id * dim_z (src/main.cpp:96). Every voxel is independent, so ranks need no collective on the data path.
"""
from ._lib import SubvolumeGeometry, SubvolumeInfo


def make_subvolume_info(vol_geo, num):
    """subvolume_info for a fixed slab count (the memory-driven count is paris_hip_make_subvolume_information)."""
    num = max(1, min(int(num), int(vol_geo.dim_z)))
    geo = SubvolumeGeometry(vol_geo.dim_x, vol_geo.dim_y, vol_geo.dim_z // num, vol_geo.dim_z % num)
    return SubvolumeInfo(geo, num)


def slab_of_task(info, task_id):
    """(first global slice, slice count) of task `task_id` (src/main.cpp:92-96)."""
    if not 0 <= task_id < info.num:
        raise ValueError("task id %d outside [0, %d)" % (task_id, info.num))
    last = (info.num - task_id) <= 1
    count = info.geo.dim_z + (info.geo.remainder if last else 0)
    return task_id * info.geo.dim_z, count


def tasks_of_rank(info, rank, world_size):
    """Round-robin assignment of slab tasks to ranks; with num == world_size each rank owns exactly one slab."""
    return [t for t in range(info.num) if t % world_size == rank]


# ---- rank placement and the job's one collective (bench.py --gpus N, tests/test_sharding_gloo.py) -------------------

def device_of_rank(local_rank, n_visible, forced=-1):
    """GPU index of a rank: `forced` when given (rehearsals), LOCAL_RANK when that many devices are visible, else
    LOCAL_RANK modulo the visible count -- a launcher may hand every rank its own single visible device
    (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES), then every rank's index is 0."""
    if forced >= 0:
        return int(forced)
    if n_visible <= 0:
        raise RuntimeError("no GPU visible to local rank %d" % local_rank)
    return local_rank if local_rank < n_visible else local_rank % n_visible


def gather_placement(dist, dev_index, props=None):
    """All-gathers (host, visible-devices mask, device index, PCI bus id / uuid when the runtime exposes them) of every rank."""
    import os
    import socket
    mine = {
        "rank": dist.get_rank(), "host": socket.gethostname(), "device": int(dev_index),
        "visible": os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES")
                   or os.environ.get("CUDA_VISIBLE_DEVICES") or "",
    }
    for attr in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        try:
            v = getattr(props, attr, None) if props is not None else None
        except Exception:  # a runtime that declares the property but cannot answer: fall back to (host, mask, index)
            v = None
        if v is not None:
            mine[attr] = str(v)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return out


def physical_key(entry):
    """what identifies the physical GPU of a rank as far as the gathered record can tell"""
    if "uuid" in entry:
        return (entry["host"], "uuid", entry["uuid"])
    if "pci_bus_id" in entry:
        return (entry["host"], "pci", entry.get("pci_domain_id", ""), entry["pci_bus_id"], entry.get("pci_device_id", ""))
    return (entry["host"], "index", entry["visible"], entry["device"])


def check_placement(placement, exclusive=True):
    """Raises when two ranks sit on the same physical GPU (exclusive=False: gloo rehearsals share one on purpose)."""
    seen = {}
    for e in placement:
        k = physical_key(e)
        if k in seen and exclusive:
            raise RuntimeError("ranks %d and %d are both placed on GPU %r: one rank per GPU is required"
                               % (seen[k], e["rank"], k))
        seen.setdefault(k, e["rank"])
    return len(seen)


def slab_checksum(slab):
    """(sum, sum of squares) of a slab in float64 -- torch tensor on any device, or a numpy array"""
    import torch
    t = slab if isinstance(slab, torch.Tensor) else torch.from_numpy(slab)
    s = torch.sum(t, dtype=torch.float64)
    q = torch.linalg.vector_norm(t.reshape(-1), ord=2, dtype=torch.float64) ** 2  # accumulates in float64, no fp64 copy
    return torch.stack([s, q])


def final_gather(dist, slab, info, rank, world, full=False, on_device=True, dst=0, task_base=0):
    """The only collective of the job (BASELINE north star: "no RCCL collective needed beyond a final gather"; the
    reference writes each slab from its own thread, src/sink.cpp:72-82). Always: all-gather of the per-slab checksums
    (every rank learns the checksum of checksums). full=True: the slabs themselves are gathered on rank `dst` in task order
    (slabs are padded to the largest slab for the collective; the last one carries the remainder, src/make_volume.cpp:32-34)
    and the assembled volume's checksum is compared with the gathered checksums.
    on_device: tensors stay on the GPU (nccl = RCCL); otherwise they go through host memory (gloo).
    task_base: rank r holds the slab of task task_base + r (rehearsals of a larger partition on fewer processes)."""
    import torch
    dev = slab.device if on_device else torch.device("cpu")
    mine = slab_checksum(slab).to(dev)
    sums = torch.empty(world * 2, dtype=torch.float64, device=dev)  # flat: the concatenated form every backend accepts
    dist.all_gather_into_tensor(sums, mine)
    sums_h = sums.cpu().reshape(world, 2)
    res = {"checksums": [float(v) for v in sums_h[:, 0]], "checksum_of_checksums": float(sums_h[:, 0].sum()),
           "sumsq": [float(v) for v in sums_h[:, 1]]}
    if not full:
        return res
    counts = [slab_of_task(info, task_base + t)[1] for t in range(world)]
    zmax = max(counts)
    plane = slab.shape[1] * slab.shape[2]
    send = slab if on_device else slab.cpu()
    if send.shape[0] != zmax:
        padded = torch.zeros((zmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        padded[:send.shape[0]] = send
        send = padded
    send = send.contiguous()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    res["gathered_bytes"] = float(sum(counts) - counts[dst]) * plane * 4.0  # bytes that crossed a link into dst
    if rank == dst:
        ok = True
        for t in range(world):
            got = float(torch.sum(recv[t][:counts[t]], dtype=torch.float64))
            want = res["checksums"][t]
            ok = ok and abs(got - want) <= 1e-9 * max(1.0, abs(want))
        res["gathered_matches_checksums"] = bool(ok)
        res["volume"] = recv, counts  # slabs in task order (padded) and their true depths
    return res


# ---- f4, second half: filter sharding (motivated by src/main.cpp:93-105: every device weights and filters
# every projection again). Rank r weights and filters projections r, r + N, r + 2N, ... once, for everybody; the ranks
# then exchange, per group of N projections, exactly the detector rows each of them needs (its slab's row band): an
# all-to-all of band-cropped frames over RCCL. Off by default: with the row band on, weighting + filtering is ~1 % of a rank's
# step and the exchange costs more than it removes (profiles/r02_rank_breakdown_c4_c5.txt).

def owner_of_projection(position, world):
    """rank that weights and filters the projection at `position` of the job (round-robin)"""
    return position % world


def exchange_filtered(dist, mine, recv, bands, rank, world, on_device=True):
    """One group of `world` projections: `mine` is this rank's filtered frame (n_col x n_row, all rows filtered; None when the
    group is short and this rank has none), `recv[q]` a full-size frame buffer that receives, in the rows of this rank's band
    bands[rank] = (first, count), the same rows of rank q's frame; rows outside the band are not touched. The exchange is
    an all-to-all of band-cropped frames written as one batch of point-to-point sends and receives (what RCCL's all-to-all
    is underneath; unlike all_to_all the batch also runs on gloo): rank r sends to q the rows of q's band, (N - 1) x band
    bytes in and out per rank and group, nothing larger. on_device=False (rehearsal with several ranks on one GPU over gloo):
    the tensors go through host memory."""
    import torch
    my_first, my_count = bands[rank]
    if mine is None:
        mine = torch.zeros_like(recv[0])
    src = mine if on_device else mine.detach().cpu()
    landing = {}
    ops = []
    for q in range(world):
        q_first, q_count = bands[q]
        if q == rank:
            recv[q][my_first:my_first + my_count].copy_(mine[my_first:my_first + my_count])
            continue
        if q_count:
            ops.append(dist.P2POp(dist.isend, src[q_first:q_first + q_count], q))      # a row range: a contiguous view
        if my_count:
            dst = recv[q][my_first:my_first + my_count]
            if not on_device:
                dst = landing[q] = torch.empty(dst.shape, dtype=dst.dtype)
            ops.append(dist.P2POp(dist.irecv, dst, q))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for q, host in landing.items():
        recv[q][my_first:my_first + my_count].copy_(host)
