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
