"""Deferral BY REFERENCE (paris_hip_set_backproject_references): a deferred backproject() of a whole
paris_hip_malloc_projection buffer takes no snapshot -- the group's fused launch reads the caller's buffer itself, paris_hip_free of
it returns at once and the buffer is recycled behind the launch (one event per group), and every other call that touches such a
buffer first launches the pending group. Whatever the caller does through the API, the volume and every buffer it reads back
must equal, bit for bit, what the snapshotting library gives -- which the other tests pin to the oracle
(/root/reference/src/openmp/backprojection.cpp:86-153; loop: src/main.cpp:98-105, buffers: src/loader.cpp:28-33,
src/cuda/memory.cpp:42-44).
"""
import ctypes as C

import numpy as np
import pytest

from paris_amd import backend as B

pytestmark = pytest.mark.gpu

KAT = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def to_host(be, d_p):
    h = be.make_projection_host(d_p.dim_x, d_p.dim_y)
    be.copy_d2h(d_p, h)
    return h.buf.copy()


def volume_to_host(be, d_v):
    h = be.make_volume_host(d_v.dim_x, d_v.dim_y, d_v.dim_z)
    be.copy_d2h(d_v, h)
    return h.buf.copy()


def upload(be, d_p, frame):
    h = C.c_void_p()
    n = frame.shape[1] * frame.shape[0] * 4
    assert be._L.paris_hip_malloc_host(be._ctx, n, C.byref(h)) == 0
    C.memmove(h.value, np.ascontiguousarray(frame).ctypes.data, n)
    B._lib.check(be._L.paris_hip_upload_projection(be._ctx, d_p.ptr, d_p.pitch, h.value, frame.shape[1] * 4, frame.shape[1], frame.shape[0]), "upload")
    assert be._L.paris_hip_free_host(be._ctx, h) == 0


@pytest.mark.parametrize("depth,overlap", [(3, False), (5, True), (48, True)])
def test_paris_loop_by_reference_equals_the_oracle(oracle, depth, overlap):
    """PARIS's loop with a buffer per projection, freed right after backproject(): the oracle's filtered frames go in, the volume
    must be the oracle's bit for bit; no deferral ring is ever allocated; the buffers rotate through the pool."""
    import torch
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    n_proj = 150
    fs = oracle.filter_size(64)
    k = oracle.make_filter(fs, odet.l_px_row)
    frames = [oracle.apply_filter(oracle.weight(oracle.lcg_projection(64, 48, i % 8), odet), k, fs) for i in range(8)]
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    det.delta_phi = odet.delta_phi = 360.0 / n_proj
    for i in range(n_proj):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, frames[i % 8], 0, odet, ovg, s, c, ds, dt)
    free0, _ = torch.cuda.mem_get_info(0)
    with B.Backend(0, synchronous=False) as abe:
        abe.set_backproject_deferral(depth)
        abe.set_backproject_overlap(overlap)
        abe.set_backproject_references(True)
        d_v = abe.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
        seen = set()
        for i in range(n_proj):
            d_p = abe.make_projection_device(64, 48)
            seen.add(d_p.ptr)
            upload(abe, d_p, frames[i % 8])
            d_p.idx = i
            B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
            abe.free(d_p)
        got = volume_to_host(abe, d_v)
        assert len(seen) < n_proj  # buffers came back from the pool
        # the rotation is bounded: at most two groups and a few more parked, one group pending
        assert len(seen) <= 3 * depth + 16
    assert np.array_equal(bits(got), bits(want))


def run_loop(oracle, refs, script, depth=4, filter_deferral=False, overlap=True, n_proj=14):
    """the loop over raw frames -- upload, weight, filter, backproject, free -- with `script(i, abe, d_p, ...)` hooks; returns the
    volume and whatever the hooks read back"""
    g = (512, 40, 0.2, 0.2, 1.25, -0.5, 300, 200, 7.0)   # 512 pixels per row: the fused weight + filter kernel
    det = B.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(96, 80, 24, nat.l_vx_x * 4.0, nat.l_vx_x * 4.5, nat.l_vx_z * 1.2)
    raws = [oracle.lcg_projection(512, 40, 40 + i) for i in range(n_proj)]
    seen = {}
    with B.Backend(0, synchronous=False) as abe:
        abe.set_stage_fusion(True)
        abe.set_backproject_deferral(depth)
        abe.set_filter_deferral(filter_deferral)
        abe.set_backproject_overlap(overlap)
        abe.set_backproject_references(refs)
        d_v = abe.make_volume_device(96, 80, 24)
        for i, raw in enumerate(raws):
            d_p = abe.make_projection_device(512, 40)
            upload(abe, d_p, raw)
            d_p.idx = i
            B.weight(abe, d_p, det)
            B.filter(abe, d_p, det)
            B.backproject(abe, d_p, d_v, 3, det, vg, False, False, None)
            keep = script(i, abe, d_p, d_v, det, vg, raws, seen)
            if not keep:
                abe.free(d_p)
        seen["volume"] = volume_to_host(abe, d_v)
    return seen


def scripts():
    def nothing(i, abe, d_p, d_v, det, vg, raws, seen):
        return False

    def filter_again(i, abe, d_p, d_v, det, vg, raws, seen):
        # a second weight + filter + backproject of the SAME buffer: the first backprojection must have read the once-filtered pixels
        if i in (1, 6, 7):
            B.weight(abe, d_p, det)
            B.filter(abe, d_p, det)
            B.backproject(abe, d_p, d_v, 3, det, vg, False, False, None)
        return False

    def read_back(i, abe, d_p, d_v, det, vg, raws, seen):
        # copy_d2h of a projection whose backprojection (and, with filter deferral, whose filter) is still pending: filtered pixels
        if i in (2, 9):
            seen["p%d" % i] = to_host(abe, d_p)
        return False

    def upload_again(i, abe, d_p, d_v, det, vg, raws, seen):
        # the buffer is refilled and used again before it is freed: the pending group must have read the first frame
        if i in (0, 5, 10):
            upload(abe, d_p, raws[(i + 3) % len(raws)])
            B.weight(abe, d_p, det)
            B.filter(abe, d_p, det)
            B.backproject(abe, d_p, d_v, 3, det, vg, False, False, None)
        return False

    def twice_unchanged(i, abe, d_p, d_v, det, vg, raws, seen):
        # the same buffer backprojected twice with nothing in between: two references to one buffer in one group
        if i in (3, 4):
            B.backproject(abe, d_p, d_v, 3, det, vg, False, False, None)
        return False

    def kept_alive(i, abe, d_p, d_v, det, vg, raws, seen):
        # buffers the caller keeps beyond the group's launch, then reads back and frees much later
        if i in (1, 2):
            seen.setdefault("kept", []).append(d_p)
            return True
        if i == 12:
            for j, q in enumerate(seen.pop("kept")):
                seen["kept%d" % j] = to_host(abe, q)
                abe.free(q)
        return False

    def other_volume(i, abe, d_p, d_v, det, vg, raws, seen):
        # a call with another slab offset in between: the pending group (by reference) is flushed first
        if i == 5:
            if "v2" not in seen:
                seen["v2"] = abe.make_volume_device(96, 80, 24)
            B.backproject(abe, d_p, seen["v2"], 7, det, vg, False, False, None)
        if i == 13:
            v2 = seen.pop("v2")
            seen["volume2"] = volume_to_host(abe, v2)
        return False

    return {f.__name__: f for f in (nothing, filter_again, read_back, upload_again, twice_unchanged, kept_alive, other_volume)}


@pytest.mark.parametrize("filter_deferral", [False, True])
@pytest.mark.parametrize("name", sorted(scripts()))
def test_every_touch_of_a_referenced_buffer_sees_what_a_snapshot_would(oracle, name, filter_deferral):
    script = scripts()[name]
    # the yardstick: snapshots, no filter deferral (pinned to the oracle by tests/test_gpu_parity.py)
    want = run_loop(oracle, False, script, filter_deferral=False)
    for depth, overlap in ((4, True), (3, False), (48, True)):
        got = run_loop(oracle, True, script, depth=depth, filter_deferral=filter_deferral, overlap=overlap)
        assert sorted(got) == sorted(want)
        for key in want:
            assert np.array_equal(bits(got[key]), bits(want[key])), (name, key, depth, overlap)
    assert np.abs(want["volume"]).max() > 0


def test_a_group_may_mix_references_and_snapshots(oracle):
    """projections in memory the library did not allocate (a torch tensor), row-band pointers and whole library buffers in one
    group: the first two are snapshotted into the ring, the third are read in place; same volume as one launch per call"""
    import torch
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    fs = oracle.filter_size(64)
    k = oracle.make_filter(fs, odet.l_px_row)
    frames = [oracle.apply_filter(oracle.weight(oracle.lcg_projection(64, 48, i), odet), k, fs) for i in range(9)]
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    for i, f in enumerate(frames):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, f, 0, odet, ovg, s, c, ds, dt)
    dev = torch.device("cuda", 0)
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        abe.set_backproject_deferral(4)
        abe.set_backproject_references(True)
        d_v = abe.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
        foreign = torch.from_numpy(np.stack(frames)).to(dev)
        for i, f in enumerate(frames):
            if i % 3 == 0:
                p = abe.wrap_projection(foreign[i].data_ptr(), 64 * 4, 64, 48, idx=i, owner=foreign)
                B.backproject(abe, p, d_v, 0, det, vg, False, False, None)
            else:
                d_p = abe.make_projection_device(64, 48)
                upload(abe, d_p, f)
                d_p.idx = i
                B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
                abe.free(d_p)
        got = volume_to_host(abe, d_v)
    assert np.array_equal(bits(got), bits(want))


def test_destroying_a_ctx_with_references_pending(oracle):
    """buffers freed while a group that is never launched refers to them (the caller lets go of everything and closes): no launch
    into a volume that is gone, nothing leaks into the next ctx, the next ctx works"""
    import torch
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    frame = oracle.lcg_projection(64, 48, 1)
    free_before, _ = torch.cuda.mem_get_info(0)
    for own_volume in (True, False):
        abe = B.Backend(0, synchronous=False)
        abe.set_backproject_deferral(48)
        abe.set_backproject_references(True)
        if own_volume:
            d_v = abe.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
        else:
            t = torch.zeros((vg.dim_z, vg.dim_y, vg.dim_x), dtype=torch.float32, device="cuda:0")
            d_v = abe.wrap_volume(t.data_ptr(), vg.dim_x, vg.dim_y, vg.dim_z, owner=t)
        for i in range(5):
            d_p = abe.make_projection_device(64, 48)
            upload(abe, d_p, frame)
            B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
            abe.free(d_p)
        abe.close()
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info(0)
    assert free_after >= free_before - (64 << 20)


def test_parked_buffers_are_bounded_over_all_sizes(oracle):
    """a driver that moves through many buffer sizes (one per row band) must not leave every size's rotation parked:
    the pools keep at most ~6 GiB per ctx, idle buffers of other sizes go back to the runtime, and what the ctx may hold for one
    detector size is what paris_hip_projection_reserve_bytes reports."""
    import torch
    with B.Backend(0, synchronous=False) as abe:
        abe.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        for size in range(24):                       # 24 sizes x 8 buffers x ~256 MiB = 48 GiB if nothing were trimmed
            rows = 4096 + 16 * size
            bufs = [abe.make_projection_device(16384, rows) for _ in range(8)]
            for b in bufs:
                abe.free(b)
        abe.synchronize()
        free1, _ = torch.cuda.mem_get_info(0)
        assert free0 - free1 <= (7 << 30), "parked: %.1f GiB" % ((free0 - free1) / 2 ** 30)
        # the reserve a driver plans with: rotation + the pending group (references) or the ring's two halves (snapshots)
        abe.set_backproject_deferral(48)
        frame = ((2048 * 4 + 255) // 256 * 256) * 2048
        snap = abe.projection_reserve_bytes(2048, 2048)
        abe.set_backproject_references(True)
        refs = abe.projection_reserve_bytes(2048, 2048)
        assert snap == 56 * frame + 96 * frame and refs == 104 * frame + 48 * frame


def test_asynchronous_validation_changes_no_bit(oracle):
    """paris_hip_set_async_validation: the validators of the hand-expanded IEEE sequences are launched and not waited for; the
    launches made meanwhile use the compiler's forms. A geometry nobody has validated in this process (its own pixel pitches and
    distances), PARIS's loop by reference, raw frames through weight + filter + backproject: the same bits as the blocking default,
    and the backprojection alone equals the oracle's bit for bit."""
    g = (96, 40, 0.2173, 0.2671, 0.75, -1.25, 137.5, 91.25, 3.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    n_proj = 40
    fs = oracle.filter_size(96)
    k = oracle.make_filter(fs, odet.l_px_row)
    filtered = [oracle.apply_filter(oracle.weight(oracle.lcg_projection(96, 40, i), odet), k, fs) for i in range(n_proj)]
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    for i, f in enumerate(filtered):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, f, 0, odet, ovg, s, c, ds, dt)

    def run(asynchronous, raw):
        with B.Backend(0, synchronous=False) as abe:
            abe.set_paris_loop_defaults(depth=4)      # what paris::hip's set_device() switches on
            abe.set_async_validation(asynchronous)
            d_v = abe.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
            for i in range(n_proj):
                d_p = abe.make_projection_device(96, 40)
                upload(abe, d_p, oracle.lcg_projection(96, 40, i) if raw else filtered[i])
                d_p.idx = i
                if raw:
                    B.weight(abe, d_p, det)
                    B.filter(abe, d_p, det)
                B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
                abe.free(d_p)
            vol = volume_to_host(abe, d_v)
            # afterwards the answers are there (the queries wait for a validator that is still running)
            assert abe.fast_division_is_exact(det.l_px_col) in (True, False)
            return vol

    got_async = run(True, False)          # first in this process: the validators really are outstanding during the first launches
    assert np.array_equal(bits(got_async), bits(want))
    assert np.array_equal(bits(run(False, False)), bits(want))
    assert np.array_equal(bits(run(True, True)), bits(run(False, True)))


@pytest.mark.parametrize("depth", [3, 48])
def test_half_precision_projections_by_reference(oracle, depth):
    """BASELINE config 5's storage through the same mechanism: the half-precision copy of each projection lives in a buffer of the
    projection allocator, is backprojected (deferred, by reference) and freed at once, like the fp32 frame it came from. Equal to
    the oracle fed the half-rounded frames, bit for bit; one frame is converted again into the SAME half buffer while the pending
    group still refers to it (the conversion's destination is guarded: the group runs first)."""
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    n_proj = 20
    frames = [(oracle.lcg_projection(64, 48, i) - np.float32(0.5)) * np.float32(3.0) for i in range(n_proj)]
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    order = []
    for i in range(n_proj):
        order.append(i)
        if i == 7:
            order.append(8)   # (projection 8's pixels, at angle 7's successor: the re-converted buffer below)
    with B.Backend(0, synchronous=False) as abe:
        abe.set_backproject_deferral(depth)
        abe.set_backproject_references(True)
        d_v = abe.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
        for i in range(n_proj):
            d_p = abe.make_projection_device(64, 48)
            upload(abe, d_p, frames[i])
            h_ptr, h_pitch = abe.convert_projection_f16(d_p)
            s, c = B.stage_angle(det, i)
            abe.backproject_f16(h_ptr, h_pitch, 64, 48, d_v, 0, det, vg, False, None, s, c, det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
            so, co, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want, frames[i].astype(np.float16).astype(np.float32), 0, odet, ovg, so, co, ds, dt)
            if i == 7:
                # the same half buffer refilled from another frame and backprojected again before it is freed
                upload(abe, d_p, frames[8])
                B._lib.check(abe._L.paris_hip_convert_projection_f16(abe._ctx, d_p.ptr, d_p.pitch, h_ptr, h_pitch, 64, 48), "convert")
                abe.backproject_f16(h_ptr, h_pitch, 64, 48, d_v, 0, det, vg, False, None, s, c, det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
                oracle.backproject(want, frames[8].astype(np.float16).astype(np.float32), 0, odet, ovg, so, co, ds, dt)
            abe.free(h_ptr)
            abe.free(d_p)
        got = volume_to_host(abe, d_v)
    assert np.array_equal(bits(got), bits(want))


def test_out_of_memory_drains_the_pools_and_retries(oracle):
    """buffers parked in the projection pool are memory too. With the device nearly full (a torch tensor holds all but
    ~3 GiB) and ~4 GiB parked in the pool, a 5 GiB volume -- then a projection buffer of a new size -- can only be allocated after the
    pool has given its buffers back: hipErrorOutOfMemory drains it and the allocation is tried once more."""
    import torch
    with B.Backend(0, synchronous=False) as abe:
        bufs = [abe.make_projection_device(16384, 4096 + 16 * (i // 8)) for i in range(16)]   # 16 x ~256 MiB, two size classes
        for b in bufs:
            abe.free(b)                                                                       # parked: ~4 GiB
        abe.synchronize()
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info(0)
        hog = torch.empty(int(free - (3 << 30)), dtype=torch.uint8, device="cuda:0")          # leaves ~3 GiB outside the pool
        free_now, _ = torch.cuda.mem_get_info(0)
        assert free_now < (4 << 30)
        d_v = abe.make_volume_device(1024, 1024, 1280)                                        # 5 GiB: needs the parked memory
        assert d_v.ptr
        abe.free(d_v)
        # the pool is empty now: fill it again and ask for a projection buffer of another size that does not fit beside it
        bufs = [abe.make_projection_device(16384, 4096) for _ in range(8)]
        for b in bufs[1:]:
            abe.free(b)
        abe.synchronize()
        big = abe.make_projection_device(16384, 4096 * 12)                                    # 3 GiB
        assert big.ptr
        abe.free(big)
        abe.free(bufs[0])
        del hog
    torch.cuda.empty_cache()
