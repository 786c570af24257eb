"""EVERY voxel of a full-size volume against the CPU oracle.

At the size of BASELINE configs 3, 2 and 5 the other tests compare the HIP path with the oracle on crops and with itself
elsewhere. Here one whole projection -- and the same projection as the first of a fused batch of three -- is compared bit for
bit over the whole volume: the oracle restates /root/reference/src/openmp/backprojection.cpp:86-153 and computes the volume slab
by slab through its own offset / ROI path (256-slice slabs, 64-bit indices), each slab is uploaded and compared on the device
(int32 views: the sign of every zero included). The HIP side runs its default path: a library-allocated volume (tiles no ray
reaches are skipped), the default tile order, depth and nesting for that shape.
"""
import numpy as np
import pytest

from paris_amd import backend as B

pytestmark = pytest.mark.gpu


def device_view(torch, v, dev):
    class _Mem:
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Mem(v.ptr, (v.dim_z, v.dim_y, v.dim_x)), device=dev)


CASES = {
    # name: (detector n, projections of the circle, grid edge, roi or None, half-precision input, projection indices: the first one lies
    #        in the slowest octant of the per-octant table (117.5 deg), the other two in a fast one and in the second slowest)
    "config3": (2048, 1440, 2048, None, False, (470, 100, 1010)),
    "config2": (1024, 720, 1024, None, False, (235, 50, 505)),
    "config5": (2048, 3600, 4096, (1024, 3072, 1024, 3072, 1024, 3072), True, (1175, 250, 2525)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_whole_volume_against_the_oracle(oracle, case):
    import torch
    n, n_proj, grid, roi_t, f16, idxs = CASES[case]
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / n_proj)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(grid))
    vg, ovg = B.VolumeGeometry(grid, grid, grid, l_vx, l_vx, l_vx), oracle.VolumeGeometry(grid, grid, grid, l_vx, l_vx, l_vx)
    roi = B.RegionOfInterest(*roi_t) if roi_t else None
    oroi = oracle.RegionOfInterest(*roi_t) if roi_t else None
    out = B.apply_roi(vg, *roi_t) if roi_t else vg
    dx, dy, dz = out.dim_x, out.dim_y, out.dim_z
    assert (dx, dy, dz) == ((2048, 2048, 2048) if case != "config2" else (1024, 1024, 1024))
    free, _ = torch.cuda.mem_get_info(0)
    if free < 2 * 4 * dx * dy * dz + (8 << 30):
        pytest.skip("needs %d GiB of free HBM" % ((2 * 4 * dx * dy * dz >> 30) + 8))
    dev = torch.device("cuda", 0)
    frames = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    if f16:  # config 5: the projections are stored as IEEE half; the oracle reads the same values widened back
        halves = [f.astype(np.float16) for f in frames]
        frames = [h.astype(np.float32) for h in halves]
    sc = [B.stage_angle(det, i) for i in idxs]
    slab = 256
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        v_one = abe.make_volume_device(dx, dy, dz)
        v_three = abe.make_volume_device(dx, dy, dz)
        t_one, t_three = device_view(torch, v_one, dev), device_view(torch, v_three, dev)
        if f16:
            stack = torch.from_numpy(np.stack(halves)).to(dev)
            abe.backproject_f16(stack[0].data_ptr(), n * 2, n, n, v_one, 0, det, vg, True, roi, sc[0][0], sc[0][1], 0.0, 0.0)
            abe.backproject_batch_f16(stack.data_ptr(), n * 2, n * n * 2, 3, n, n, v_three, 0, det, vg, True, roi,
                                      [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        else:
            stack = torch.from_numpy(np.stack(frames)).to(dev)
            p0 = abe.wrap_projection(stack[0].data_ptr(), n * 4, n, n, idx=idxs[0], owner=stack)
            B.backproject(abe, p0, v_one, 0, det, vg, False, roi is not None, roi)
            abe.backproject_batch(stack.data_ptr(), n * 4, n * n * 4, 3, n, n, v_three, 0, det, vg, roi is not None, roi,
                                  [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        abe.synchronize()
        # the oracle's slab lives in pinned memory (the upload of 1 GiB takes ~40 ms instead of ~300)
        pinned = torch.empty((slab, dy, dx), dtype=torch.float32, pin_memory=True)
        want = pinned.numpy()
        on_dev = torch.empty((slab, dy, dx), dtype=torch.float32, device=dev)
        consts = [oracle.backproject_constants(odet, i) for i in idxs]
        nonzero = 0
        for z0 in range(0, dz, slab):
            want[...] = 0.0
            s, c, ds, dt = consts[0]
            oracle.backproject(want, frames[0], z0, odet, ovg, s, c, ds, dt, oroi)     # v_offset = z0 (+ roi.z1 inside the oracle)
            on_dev.copy_(pinned)
            assert torch.equal(on_dev.view(torch.int32), t_one[z0:z0 + slab].view(torch.int32)), \
                "%s: one projection, slices %d..%d differ from the oracle" % (case, z0, z0 + slab - 1)
            for j in (1, 2):
                s, c, ds, dt = consts[j]
                oracle.backproject(want, frames[j], z0, odet, ovg, s, c, ds, dt, oroi)
            on_dev.copy_(pinned)
            torch.cuda.synchronize()
            assert torch.equal(on_dev.view(torch.int32), t_three[z0:z0 + slab].view(torch.int32)), \
                "%s: fused batch of 3, slices %d..%d differ from the oracle" % (case, z0, z0 + slab - 1)
            nonzero += int(torch.count_nonzero(on_dev))
        # something was compared: most of the volume lies in the field of view (config 5's ROI: all of it)
        assert nonzero > 0.5 * dx * dy * dz
        del t_one, t_three
        abe.free(v_one)
        abe.free(v_three)


def test_config5_uncropped_slab(oracle):
    """BASELINE config 5 WITHOUT the ROI crop (SURVEY 8d's stretch variant; the reference makes the ROI optional,
    /root/reference/src/main.cpp:124-130, and splits by memory, src/cuda/subvolume_information.cpp:72-116): one rank's slab of the
    8-GPU job, 4096 x 4096 x 512 voxels (32 GiB) of the 4096^3 grid at v_offset 1536, half-precision projections. Two single
    launches and the same two as one fused batch must agree on the device bit for bit; the oracle pins crops at a corner of the
    plane, at its centre, and on the slab's first and last slice (its own offset path on the half-rounded projections)."""
    import torch
    n, grid, nz, v_offset = 2048, 4096, 512, 1536
    free, _ = torch.cuda.mem_get_info(0)
    if free < 2 * 4 * grid * grid * nz + (8 << 30):
        pytest.skip("needs 72 GiB of free HBM")
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 3600)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l_vx = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(grid))
    vg, ovg = B.VolumeGeometry(grid, grid, grid, l_vx, l_vx, l_vx), oracle.VolumeGeometry(grid, grid, grid, l_vx, l_vx, l_vx)
    idxs = (1175, 3001)
    halves = [(oracle.lcg_projection(n, n, i) - np.float32(0.5)).astype(np.float16) for i in idxs]
    frames = [h.astype(np.float32) for h in halves]
    sc = [B.stage_angle(det, i) for i in idxs]
    dev = torch.device("cuda", 0)
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        stack = torch.from_numpy(np.stack(halves)).to(dev)
        v_a = abe.make_volume_device(grid, grid, nz)
        v_b = abe.make_volume_device(grid, grid, nz)
        t_a, t_b = device_view(torch, v_a, dev), device_view(torch, v_b, dev)
        for j in range(2):
            abe.backproject_f16(stack[j].data_ptr(), n * 2, n, n, v_a, v_offset, det, vg, False, None, sc[j][0], sc[j][1], 0.0, 0.0)
        abe.backproject_batch_f16(stack.data_ptr(), n * 2, n * n * 2, 2, n, n, v_b, v_offset, det, vg, False, None,
                                  [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        abe.synchronize()
        assert torch.equal(t_a.view(torch.int32), t_b.view(torch.int32))
        # the grid's corners lie outside the field of view, its middle inside
        assert float(t_a[:, :64, :64].abs().max()) == 0.0 and float(t_a[:, 2016:2080, 2016:2080].abs().min()) > 0.0
        consts = [oracle.backproject_constants(odet, i) for i in idxs]
        for (x1, y1, z1) in ((0, 0, 0), (4032, 4032, nz - 8), (2016, 2016, 0), (2016, 2016, nz - 8), (700, 3300, 250), (3900, 2000, 100)):
            crop = t_a[z1:z1 + 8, y1:y1 + 64, x1:x1 + 64].cpu().numpy()
            oroi = oracle.RegionOfInterest(x1, x1 + 64, y1, y1 + 64, 0, grid)
            want = np.zeros((8, 64, 64), np.float32)
            for f, (s, c, ds, dt) in zip(frames, consts):
                oracle.backproject(want, f, v_offset + z1, odet, ovg, s, c, ds, dt, oroi)
            assert np.array_equal(crop.view(np.uint32), want.view(np.uint32)), (x1, y1, z1)
        del t_a, t_b
        abe.free(v_a)
        abe.free(v_b)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PARIS_FUZZ_PLANES", "4"))))
def test_random_geometries_on_large_planes(oracle, seed):
    """the paths a launch takes only on planes beyond 1024^2 -- 16-slice tiles with ONE slice in flight at four
    workgroups per CU and the deep nesting of the dealt order from 257 slices (odd seeds: 257 .. 700 slices), 8-slice tiles up to
    256 slices (even seeds: 30 .. 256) -- on seeded random geometries (detector size and pitch, offsets, cone, anisotropic voxels,
    slab offset; every third seed an unaligned row length: narrower lanes), the default path with the skip on, single launches and
    the fused batch, EVERY voxel against the oracle. PARIS_FUZZ_PLANES=n draws n seeds (the round's fuzz campaign: 40)."""
    import torch
    rng = np.random.default_rng(77000 + seed)
    n_row, n_col = int(rng.integers(300, 900)), int(rng.integers(200, 700))
    l_r, l_c = float(rng.choice([0.1, 0.2, 0.127, 0.4])), float(rng.choice([0.1, 0.2, 0.25]))
    d_so, d_od = float(rng.uniform(150, 600)), float(rng.uniform(50, 600))
    g = (n_row, n_col, l_r, l_c, float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), d_so, d_od, float(rng.uniform(0.5, 40)))
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    dx, dy = int(rng.integers(1028, 1500)), int(rng.integers(1026, 1400))
    dx = dx if seed % 3 == 2 else (dx + 3) // 4 * 4                      # (unaligned rows every third seed)
    dz = int(rng.integers(257, 700)) if seed % 2 else int(rng.integers(30, 257))
    assert dx * dy > (1 << 20)
    full_z = dz + int(rng.integers(0, 60))
    v_offset = full_z - dz
    scale = [float(nat.l_vx_x * n_row / dx * rng.uniform(0.7, 1.3)), float(nat.l_vx_x * n_row / dy * rng.uniform(0.7, 1.3)),
             float(nat.l_vx_z * n_col / full_z * rng.uniform(0.6, 1.2))]
    half_diag = 0.5 * np.hypot(dx * scale[0], dy * scale[1])
    if half_diag > 0.8 * d_so:
        scale[0] *= 0.8 * d_so / half_diag
        scale[1] *= 0.8 * d_so / half_diag
    vg, ovg = B.VolumeGeometry(dx, dy, full_z, *scale), oracle.VolumeGeometry(dx, dy, full_z, *scale)
    angles = [float(rng.uniform(0, 360)) for _ in range(3)]
    projs = [oracle.lcg_projection(n_row, n_col, 50 * seed + i) - np.float32(0.5) for i in range(3)]
    want = np.zeros((dz, dy, dx), np.float32)
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i, True, angles[i])
        oracle.backproject(want, p, v_offset, odet, ovg, s, c, ds, dt, None)
    dev = torch.device("cuda", 0)
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        stack = torch.from_numpy(np.stack(projs)).to(dev)
        v_a = abe.make_volume_device(dx, dy, dz)
        v_b = abe.make_volume_device(dx, dy, dz)
        for i in range(3):
            p = abe.wrap_projection(stack[i].data_ptr(), n_row * 4, n_row, n_col, idx=i, phi=angles[i], owner=stack)
            B.backproject(abe, p, v_a, v_offset, det, vg, True, False, None)
        sc = [B.stage_angle(det, i, True, angles[i]) for i in range(3)]
        abe.backproject_batch(stack.data_ptr(), n_row * 4, n_row * n_col * 4, 3, n_row, n_col, v_b, v_offset, det, vg, False, None,
                              [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
        abe.synchronize()
        w = torch.from_numpy(want).to(dev)
        assert torch.equal(device_view(torch, v_a, dev).view(torch.int32), w.view(torch.int32)), (seed, "single launches", dx, dy, dz)
        assert torch.equal(device_view(torch, v_b, dev).view(torch.int32), w.view(torch.int32)), (seed, "fused batch", dx, dy, dz)
        assert int(torch.count_nonzero(w)) > 0
        abe.free(v_a)
        abe.free(v_b)
