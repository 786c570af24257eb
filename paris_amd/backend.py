"""Host-side mirror of the PARIS backend surface for the MI355X HIP backend.

The reference's backend is a C++ namespace with 14 free functions plus buffer types
(src/generic/backend.h:54-88, src/openmp/backend.h:42-89, src/cuda/backend.h:50-103). This module exposes
the same names, argument meaning and error behaviour over the C ABI of include/paris_hip.h, so the parity
tests read like a PARIS driver loop (src/main.cpp:98-105). The C++ equivalent is paris_amd/host/.

All numeric work happens in libparis_hip.so on the GPU; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (DetectorGeometry, ParisHipError, RegionOfInterest, SubvolumeGeometry, SubvolumeInfo,
                   VolumeGeometry, check)

__all__ = ["DetectorGeometry", "VolumeGeometry", "SubvolumeGeometry", "RegionOfInterest", "SubvolumeInfo",
           "ParisHipError", "Projection", "Volume", "FilterBuffer", "Backend", "get_devices", "set_device",
           "calculate_volume_geometry", "apply_roi", "filter_size", "load", "make_volume", "weight", "filter",
           "backproject"]


class Projection:
    """paris::projection<Buffer, Metadata> (src/projection.h:31-46). `buf` is a numpy array (host) or a device
    address (device); `pitch` is the row stride in bytes."""

    def __init__(self, buf, dim_x, dim_y, idx=0, phi=0.0, pitch=None, on_device=False, owner=None):
        self.buf = buf
        self.dim_x = dim_x
        self.dim_y = dim_y
        self.idx = idx
        self.phi = phi
        self.pitch = pitch if pitch is not None else dim_x * 4
        self.on_device = on_device
        self._owner = owner  # Backend that must free buf, or an object that keeps it alive

    @property
    def ptr(self):
        return self.buf if self.on_device else self.buf.ctypes.data


class Volume:
    """paris::volume<Buffer> (src/volume.h:31-45); x fastest, then y, then z."""

    def __init__(self, buf, dim_x, dim_y, dim_z, off=0, on_device=False, owner=None):
        self.buf = buf
        self.dim_x = dim_x
        self.dim_y = dim_y
        self.dim_z = dim_z
        self.off = off
        self.on_device = on_device
        self._owner = owner

    @property
    def ptr(self):
        return self.buf if self.on_device else self.buf.ctypes.data


class FilterBuffer:
    """backend::filter_buffer_type: K = tau * |rFFT(r)|, size/2+1 floats on the device."""

    def __init__(self, ptr, size, backend):
        self.ptr = ptr
        self.size = size
        self._backend = backend


class Backend:
    """One device + stream: the state the reference keeps in thread_local statics after set_device
    (src/main.cpp:87). synchronous=True reproduces the reference's blocking calls. stream: None = a private stream of
    the ctx, a hipStream_t handle = that stream, 0 = the legacy default stream."""

    def __init__(self, device=0, stream=None, synchronous=True):
        self._L = _lib.load()
        ctx = C.c_void_p()
        flags = _lib.CTX_SYNCHRONOUS if synchronous else _lib.CTX_DEFAULT
        if stream is not None and stream == 0:
            # the caller's stream is the legacy default stream (what torch.cuda.current_stream().cuda_stream returns unless a
            # stream was set): enqueue there too, so that the caller's own work and these kernels stay ordered
            flags |= _lib.CTX_LEGACY_STREAM
        check(self._L.paris_hip_ctx_create(device, C.c_void_p(stream) if stream else None, flags, C.byref(ctx)),
              "paris_hip_ctx_create")
        self._ctx = ctx
        self.device = device
        self._owned = set()
        self._wrapped = {}  # device address of a wrapped volume -> weak reference to its owner (None: wrapped without one)

    # ---- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if self._ctx is not None:
            # projections still deferred into a wrapped (caller-owned) volume: run them while the caller's memory is certainly
            # still there -- paris_hip_ctx_destroy itself only runs pending work into volumes the library allocated
            if self._pending_volume_alive():
                self._L.paris_hip_flush(self._ctx)
            for p in list(self._owned):
                self._L.paris_hip_free(self._ctx, C.c_void_p(p))
            self._owned.clear()
            self._L.paris_hip_ctx_destroy(self._ctx)
            self._ctx = None

    def _pending_volume_alive(self):
        """True when projections are pending into a wrapped volume whose owner (wrap_volume(owner=...)) is still alive. Decided
        for THAT volume alone (one owner-less wrap used to switch the close-time flush off for every volume). A
        volume wrapped without an owner is never flushed into at close: nothing says its memory still belongs to the caller --
        callers of owner-less wraps call flush() before they let go of the memory."""
        n, ptr = C.c_uint32(0), C.c_void_p()
        if self._L.paris_hip_pending_backprojections(self._ctx, C.byref(n), C.byref(ptr)) != 0 or n.value == 0 or not ptr.value:
            return False
        for base, (ref, nbytes) in self._wrapped.items():
            if base <= ptr.value < base + nbytes:
                return ref is not None and ref() is not None
        return False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self._L.paris_hip_ctx_synchronize(self._ctx), "paris_hip_ctx_synchronize")

    @property
    def stream(self):
        """the hipStream_t handle the ctx enqueues on (0 = the legacy default stream)"""
        return self._L.paris_hip_ctx_stream(self._ctx) or 0

    def free(self, obj):
        ptr = obj.ptr if hasattr(obj, "ptr") else obj
        if ptr in self._owned:
            check(self._L.paris_hip_free(self._ctx, C.c_void_p(ptr)), "paris_hip_free")
            self._owned.discard(ptr)

    # ---- memory (src/openmp/memory.cpp:33-79, src/cuda/memory.cpp:33-102) -------------------------------
    def make_projection_host(self, dim_x, dim_y):
        return Projection(np.empty((dim_y, dim_x), np.float32), dim_x, dim_y)

    def make_projection_device(self, dim_x, dim_y):
        ptr, pitch = C.c_void_p(), C.c_size_t()
        check(self._L.paris_hip_malloc_projection(self._ctx, dim_x, dim_y, C.byref(ptr), C.byref(pitch)),
              "paris_hip_malloc_projection")
        self._owned.add(ptr.value)
        return Projection(ptr.value, dim_x, dim_y, pitch=pitch.value, on_device=True, owner=self)

    def make_volume_host(self, dim_x, dim_y, dim_z):
        return Volume(np.zeros((dim_z, dim_y, dim_x), np.float32), dim_x, dim_y, dim_z)

    def make_volume_device(self, dim_x, dim_y, dim_z):
        ptr = C.c_void_p()
        check(self._L.paris_hip_malloc_volume(self._ctx, dim_x, dim_y, dim_z, C.byref(ptr)),
              "paris_hip_malloc_volume")
        self._owned.add(ptr.value)
        return Volume(ptr.value, dim_x, dim_y, dim_z, on_device=True, owner=self)

    def wrap_projection(self, ptr, pitch, dim_x, dim_y, idx=0, phi=0.0, owner=None):
        """Adopts device memory allocated elsewhere (e.g. a torch tensor's data_ptr())."""
        return Projection(ptr, dim_x, dim_y, idx, phi, pitch=pitch, on_device=True, owner=owner)

    def wrap_volume(self, ptr, dim_x, dim_y, dim_z, off=0, owner=None, clean=None):
        """Adopts device memory allocated elsewhere as a volume. The library knows nothing about its contents, so every
        addition is performed (paris_hip_set_backproject_skip_invalid does not apply) unless the caller vouches with clean=True
        that it holds no -0 right now (e.g. torch.zeros) and will call volume_mark_dirty before writing anything but zeros into
        it; clean=False withdraws an earlier promise for the range; clean="scan" lets the device look for -0 once
        (paris_hip_volume_scan_clean) and skip when there is none, under the same duty."""
        v = Volume(ptr, dim_x, dim_y, dim_z, off, on_device=True, owner=owner)
        import weakref
        for base in [b for b, (ref, _) in self._wrapped.items() if ref is not None and ref() is None]:
            del self._wrapped[base]  # owners that have been collected: their wraps are history
        try:
            ref = weakref.ref(owner) if owner is not None else None
        except TypeError:
            ref = None
        self._wrapped[ptr] = (ref, 4 * dim_x * dim_y * dim_z)
        if clean is True:
            self.volume_mark_clean(v)
        elif clean is False:
            self.volume_mark_dirty(v)
        elif clean == "scan":
            self.volume_scan_clean(v)
        elif clean is not None:
            raise ValueError("clean must be None, True, False or 'scan'")
        return v

    def volume_mark_dirty(self, v):
        """the caller wrote the volume itself and may have stored a -0 (paris_hip_volume_mark_dirty)"""
        check(self._L.paris_hip_volume_mark_dirty(self._ctx, v.ptr, 4 * v.dim_x * v.dim_y * v.dim_z), "paris_hip_volume_mark_dirty")

    def volume_mark_clean(self, v):
        """the caller vouches that the volume holds no -0 right now (paris_hip_volume_mark_clean)"""
        check(self._L.paris_hip_volume_mark_clean(self._ctx, v.ptr, 4 * v.dim_x * v.dim_y * v.dim_z), "paris_hip_volume_mark_clean")

    def volume_scan_clean(self, v):
        """reads the volume once on the device; lists it as clean when it holds no -0; returns the number of -0 found"""
        n = C.c_uint64(0)
        check(self._L.paris_hip_volume_scan_clean(self._ctx, v.ptr, 4 * v.dim_x * v.dim_y * v.dim_z, C.byref(n)),
              "paris_hip_volume_scan_clean")
        return int(n.value)

    def memset_volume(self, v):
        check(self._L.paris_hip_memset_volume(self._ctx, v.ptr, v.dim_x, v.dim_y, v.dim_z), "paris_hip_memset_volume")

    def copy_h2d(self, h, d):
        """copy_h2d for projections (carries idx, phi: src/openmp/memory.cpp:60-62) and volumes (off: :73)."""
        if isinstance(h, Projection):
            assert h.buf.dtype == np.float32 and h.dim_x == d.dim_x and h.dim_y == d.dim_y
            check(self._L.paris_hip_memcpy_projection_h2d(self._ctx, d.ptr, d.pitch, h.ptr, h.buf.strides[0],
                                                          h.dim_x, h.dim_y), "paris_hip_memcpy_projection_h2d")
            d.idx, d.phi = h.idx, h.phi
        else:
            assert h.buf.dtype == np.float32 and (h.dim_x, h.dim_y, h.dim_z) == (d.dim_x, d.dim_y, d.dim_z)
            check(self._L.paris_hip_memcpy_volume_h2d(self._ctx, d.ptr, h.ptr, h.dim_x, h.dim_y, h.dim_z),
                  "paris_hip_memcpy_volume_h2d")
            d.off = h.off
        self.synchronize()

    def upload(self, h, d):
        """copy_h2d for a projection on the ctx's upload stream (paris_hip_upload_projection): returns without waiting;
        kernels enqueued afterwards are ordered behind the transfer, kernels already queued overlap it."""
        assert h.buf.dtype == np.float32 and h.dim_x == d.dim_x and h.dim_y == d.dim_y
        check(self._L.paris_hip_upload_projection(self._ctx, d.ptr, d.pitch, h.ptr, h.buf.strides[0],
                                                  h.dim_x, h.dim_y), "paris_hip_upload_projection")
        d.idx, d.phi = h.idx, h.phi

    def copy_d2h(self, d, h):
        if isinstance(d, Projection):
            assert h.buf.dtype == np.float32 and h.dim_x == d.dim_x and h.dim_y == d.dim_y
            check(self._L.paris_hip_memcpy_projection_d2h(self._ctx, h.ptr, h.buf.strides[0], d.ptr, d.pitch,
                                                          d.dim_x, d.dim_y), "paris_hip_memcpy_projection_d2h")
            h.idx, h.phi = d.idx, d.phi
        else:
            assert h.buf.dtype == np.float32 and (h.dim_x, h.dim_y, h.dim_z) == (d.dim_x, d.dim_y, d.dim_z)
            check(self._L.paris_hip_memcpy_volume_d2h(self._ctx, h.ptr, d.ptr, d.dim_x, d.dim_y, d.dim_z),
                  "paris_hip_memcpy_volume_d2h")
            h.off = d.off
        self.synchronize()

    def make_subvolume_information(self, vol_geo, det_geo, n_devices=0):
        """src/cuda/subvolume_information.cpp:63-118"""
        out = SubvolumeInfo()
        check(self._L.paris_hip_make_subvolume_information(C.byref(vol_geo), C.byref(det_geo), n_devices,
                                                           C.byref(out)), "paris_hip_make_subvolume_information")
        return out

    # ---- the three numeric stages --------------------------------------------------------------------------
    def weight(self, p, h_min, v_min, d_sd, l_px_row, l_px_col):
        """backend::weight (src/openmp/weighting.cpp:32-57)"""
        check(self._L.paris_hip_weight(self._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, h_min, v_min, d_sd,
                                       l_px_row, l_px_col), "paris_hip_weight")

    def make_filter(self, size, tau, window=0):
        """backend::make_filter (src/openmp/filtering.cpp:139-165); window 1 = Shepp-Logan (extension)"""
        ptr = C.c_void_p()
        check(self._L.paris_hip_make_filter_windowed(self._ctx, size, tau, window, C.byref(ptr)), "paris_hip_make_filter")
        self._owned.add(ptr.value)
        return FilterBuffer(ptr.value, size, self)

    def filter_to_host(self, k):
        n = k.size // 2 + 1
        out = np.empty((1, n), np.float32)
        check(self._L.paris_hip_memcpy_projection_d2h(self._ctx, out.ctypes.data, n * 4, k.ptr, n * 4, n, 1),
              "paris_hip_memcpy_projection_d2h")
        self.synchronize()
        return out[0]

    def apply_filter(self, p, k, filter_size, n_col):
        """backend::apply_filter (src/openmp/filtering.cpp:167-219)"""
        check(self._L.paris_hip_apply_filter(self._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, k.ptr, filter_size,
                                             n_col), "paris_hip_apply_filter")

    def set_filter_variant(self, variant):
        check(self._L.paris_hip_set_filter_variant(self._ctx, variant), "paris_hip_set_filter_variant")

    def set_backproject_skip_invalid(self, enable=True):
        """paris_hip_set_backproject_skip_invalid: tiles no ray of a projection reaches are left untouched (library-allocated
        volumes only; default on)"""
        check(self._L.paris_hip_set_backproject_skip_invalid(self._ctx, int(bool(enable))), "paris_hip_set_backproject_skip_invalid")

    def set_stage_fusion(self, enable=True):
        """weight() is held back and rides along in the load of the apply_filter() that follows (one launch)"""
        check(self._L.paris_hip_set_stage_fusion(self._ctx, int(bool(enable))), "paris_hip_set_stage_fusion")

    def weight_filter_rows(self, p, row_first, row_count, h_min, v_min, d_sd, l_px_row, l_px_col, k, filter_size, half_ptr=None,
                           half_pitch=0):
        """weighting + row filter of a band of rows in one launch (extension); half_ptr: store IEEE half there instead"""
        check(self._L.paris_hip_weight_filter_rows(self._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, row_first, row_count, h_min, v_min,
                                                   d_sd, l_px_row, l_px_col, k.ptr, filter_size, half_ptr, half_pitch),
              "paris_hip_weight_filter_rows")

    def backproject(self, p, v, v_offset, det_geo, vol_geo, enable_roi, roi, sin, cos, delta_s, delta_t):
        """backend::backproject (src/openmp/backprojection.cpp:156-199)"""
        r = roi if roi is not None else RegionOfInterest()
        check(self._L.paris_hip_backproject(self._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, v.ptr, v.dim_x, v.dim_y,
                                            v.dim_z, v_offset, C.byref(det_geo), C.byref(vol_geo),
                                            int(bool(enable_roi)), C.byref(r), sin, cos, delta_s, delta_t),
              "paris_hip_backproject")

    def convert_projection_f16(self, p):
        """fp32 device projection -> new device buffer of IEEE half pixels (round to nearest even); returns
        (device address, pitch in bytes). Free with Backend.free(address)."""
        pitch = (p.dim_x * 2 + 255) // 256 * 256
        d, dp = C.c_void_p(), C.c_size_t()
        # the projection allocator hands out rows of `pitch` bytes (pitch / 4 fp32 slots, already a 256 B multiple)
        check(self._L.paris_hip_malloc_projection(self._ctx, pitch // 4, p.dim_y, C.byref(d), C.byref(dp)),
              "paris_hip_malloc_projection")
        self._owned.add(d.value)
        check(self._L.paris_hip_convert_projection_f16(self._ctx, p.ptr, p.pitch, d.value, dp.value, p.dim_x, p.dim_y),
              "paris_hip_convert_projection_f16")
        return d.value, dp.value

    def backproject_f16(self, p_ptr, p_pitch, p_dim_x, p_dim_y, v, v_offset, det_geo, vol_geo, enable_roi, roi, sin, cos,
                        delta_s, delta_t):
        r = roi if roi is not None else RegionOfInterest()
        check(self._L.paris_hip_backproject_f16(self._ctx, p_ptr, p_pitch, p_dim_x, p_dim_y, v.ptr, v.dim_x, v.dim_y,
                                                v.dim_z, v_offset, C.byref(det_geo), C.byref(vol_geo),
                                                int(bool(enable_roi)), C.byref(r), sin, cos, delta_s, delta_t),
              "paris_hip_backproject_f16")

    def backproject_batch(self, p_ptr, p_pitch, p_stride, n_proj, p_dim_x, p_dim_y, v, v_offset, det_geo, vol_geo,
                          enable_roi, roi, sins, coss, delta_s, delta_t):
        r = roi if roi is not None else RegionOfInterest()
        s = (C.c_float * n_proj)(*sins)
        c = (C.c_float * n_proj)(*coss)
        check(self._L.paris_hip_backproject_batch(self._ctx, p_ptr, p_pitch, p_stride, n_proj, p_dim_x, p_dim_y,
                                                  v.ptr, v.dim_x, v.dim_y, v.dim_z, v_offset, C.byref(det_geo),
                                                  C.byref(vol_geo), int(bool(enable_roi)), C.byref(r), s, c,
                                                  delta_s, delta_t), "paris_hip_backproject_batch")

    def backproject_batch_f16(self, p_ptr, p_pitch, p_stride, n_proj, p_dim_x, p_dim_y, v, v_offset, det_geo, vol_geo,
                              enable_roi, roi, sins, coss, delta_s, delta_t):
        r = roi if roi is not None else RegionOfInterest()
        s = (C.c_float * n_proj)(*sins)
        c = (C.c_float * n_proj)(*coss)
        check(self._L.paris_hip_backproject_batch_f16(self._ctx, p_ptr, p_pitch, p_stride, n_proj, p_dim_x, p_dim_y,
                                                      v.ptr, v.dim_x, v.dim_y, v.dim_z, v_offset, C.byref(det_geo),
                                                      C.byref(vol_geo), int(bool(enable_roi)), C.byref(r), s, c,
                                                      delta_s, delta_t), "paris_hip_backproject_batch_f16")

    # ---- diagnostics ------------------------------------------------------------------------------------------
    def last_backproject_ms(self):
        ms = C.c_float()
        check(self._L.paris_hip_last_backproject_ms(self._ctx, C.byref(ms)), "paris_hip_last_backproject_ms")
        return ms.value

    def backproject_timing_arm(self, capacity):
        check(self._L.paris_hip_backproject_timing_arm(self._ctx, capacity), "paris_hip_backproject_timing_arm")

    def backproject_timing_collect(self, max_n=65536):
        ms = (C.c_float * max_n)()
        n = C.c_uint32()
        check(self._L.paris_hip_backproject_timing_collect(self._ctx, ms, max_n, C.byref(n)),
              "paris_hip_backproject_timing_collect")
        return list(ms[:n.value])

    def set_backproject_variant(self, variant):
        check(self._L.paris_hip_set_backproject_variant(self._ctx, variant), "paris_hip_set_backproject_variant")

    def set_backproject_order(self, order=-1, nontemporal=-1):
        check(self._L.paris_hip_set_backproject_order(self._ctx, order, nontemporal), "paris_hip_set_backproject_order")

    def fast_division_is_exact(self, divisor):
        ok = C.c_int()
        check(self._L.paris_hip_fast_division_is_exact(self._ctx, divisor, C.byref(ok)), "paris_hip_fast_division_is_exact")
        return bool(ok.value)

    def set_backproject_vector_staging(self, enable=True):
        check(self._L.paris_hip_set_backproject_vector_staging(self._ctx, int(bool(enable))),
              "paris_hip_set_backproject_vector_staging")

    def set_backproject_fast_division(self, enable=True):
        check(self._L.paris_hip_set_backproject_fast_division(self._ctx, int(bool(enable))),
              "paris_hip_set_backproject_fast_division")

    def set_backproject_slice_shape(self, waves=0, row_groups=0):
        check(self._L.paris_hip_set_backproject_slice_shape(self._ctx, waves, row_groups),
              "paris_hip_set_backproject_slice_shape")

    def set_filter_window(self, window):
        """window of the K that the filter stage wrapper builds: 0 = the reference's ramp, 1 = Shepp-Logan"""
        check(self._L.paris_hip_set_filter_window(self._ctx, window), "paris_hip_set_filter_window")

    def set_filter_deferral(self, enable):
        """paris_hip_set_filter_deferral: with stage fusion and a deferral depth > 1, the filter() that follows a weight() is held back
        too and runs on the library's snapshot, a group per launch, if the next call backprojects that projection (whose buffer then
        keeps its unfiltered pixels); bit-identical volume"""
        check(self._L.paris_hip_set_filter_deferral(self._ctx, 2 if enable == 2 else int(bool(enable))), "paris_hip_set_filter_deferral")

    def set_backproject_deferral(self, depth):
        """depth > 1: backproject() calls are snapshotted and added by one fused launch per `depth` calls (bit-identical)"""
        check(self._L.paris_hip_set_backproject_deferral(self._ctx, depth), "paris_hip_set_backproject_deferral")

    def set_backproject_overlap(self, enable=True):
        """deferred fused launches run on a second stream beside the caller's next calls (default off: measured slower)"""
        check(self._L.paris_hip_set_backproject_overlap(self._ctx, int(bool(enable))), "paris_hip_set_backproject_overlap")

    def set_backproject_references(self, enable=True):
        """deferral by reference: a deferred backproject() of a whole make_projection_device buffer takes no snapshot, the group's
        fused launch reads the buffer itself; free() of it returns at once and the buffer is recycled behind the launch; any other
        call that touches it launches the pending group first (bit-identical; paris::hip switches it on)"""
        check(self._L.paris_hip_set_backproject_references(self._ctx, int(bool(enable))), "paris_hip_set_backproject_references")

    def set_async_validation(self, enable=True):
        """validators of the hand-expanded IEEE sequences are launched and not waited for; the compiler's forms serve until they
        have answered (same bits)"""
        check(self._L.paris_hip_set_async_validation(self._ctx, int(bool(enable))), "paris_hip_set_async_validation")

    def projection_reserve_bytes(self, dim_x, dim_y):
        """what this ctx may keep allocated for projections of that size beside the volume (buffer rotation, pending group or ring)"""
        n = C.c_size_t(0)
        check(self._L.paris_hip_projection_reserve_bytes(self._ctx, dim_x, dim_y, C.byref(n)), "paris_hip_projection_reserve_bytes")
        return int(n.value)

    def set_paris_loop_defaults(self, depth=48):
        """The settings the C++ mirror paris::hip makes in set_device() (paris_amd/host/paris/hip/backend.h) for PARIS's own loop --
        a buffer allocated, filled, weighted, filtered, backprojected and freed per projection (src/main.cpp:98-105): stage fusion,
        `depth` backprojections per fused launch on the second stream, projections by reference, the filter in place a group at a
        time, validators beside the first projections. Every result stays bit-identical; needs an asynchronous backend."""
        self.set_stage_fusion(True)
        self.set_backproject_deferral(depth)
        self.set_backproject_overlap(True)
        self.set_backproject_references(True)
        self.set_filter_deferral(2)
        self.set_async_validation(True)

    def flush(self):
        check(self._L.paris_hip_flush(self._ctx), "paris_hip_flush")

    def set_backproject_tuning(self, vx=0, unroll=0, tz=0, lds_bytes=0):
        check(self._L.paris_hip_set_backproject_tuning(self._ctx, vx, unroll, tz, lds_bytes),
              "paris_hip_set_backproject_tuning")


# ---- device management (src/cuda/device.cpp:31-47, src/openmp/backend.h:87-89) --------------------------------

def get_devices():
    n = C.c_int()
    check(_lib.load().paris_hip_device_count(C.byref(n)), "paris_hip_device_count")
    return list(range(n.value))


def set_device(device, stream=None, synchronous=True):
    """Binds the calling thread's work to `device`; returns the Backend that carries the per-device state."""
    return Backend(device, stream, synchronous)


# ---- geometry (src/geometry.cpp) -----------------------------------------------------------------------------

def calculate_volume_geometry(det_geo):
    out = VolumeGeometry()
    check(_lib.load().paris_hip_calculate_volume_geometry(C.byref(det_geo), C.byref(out)),
          "paris_hip_calculate_volume_geometry")
    return out


def apply_roi(vol_geo, x1, x2, y1, y2, z1, z2):
    out = VolumeGeometry()
    roi = RegionOfInterest(x1, x2, y1, y2, z1, z2)
    check(_lib.load().paris_hip_apply_roi(C.byref(vol_geo), C.byref(roi), C.byref(out)), "paris_hip_apply_roi")
    return out


def filter_size(n_row):
    return int(_lib.load().paris_hip_filter_size(n_row))


# ---- stage wrappers (src/loader.cpp, make_volume.cpp, weighting.cpp, filtering.cpp, backprojection.cpp) --------

def load(backend, p):
    """paris::load (src/loader.cpp:28-33)"""
    d_p = backend.make_projection_device(p.dim_x, p.dim_y)
    backend.copy_h2d(p, d_p)
    return d_p


def make_volume(backend, subvol_geo, last):
    """paris::make_volume (src/make_volume.cpp:30-37)"""
    dim_z = subvol_geo.dim_z
    if last:
        dim_z += subvol_geo.remainder
    return backend.make_volume_device(subvol_geo.dim_x, subvol_geo.dim_y, dim_z)


def weight(backend, p, det_geo):
    """paris::weight (src/weighting.cpp:32-45)"""
    check(backend._L.paris_hip_stage_weight(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, C.byref(det_geo)),
          "paris_hip_stage_weight")


def filter(backend, p, det_geo):  # noqa: A001 - the reference's name
    """paris::filter (src/filtering.cpp:32-45)"""
    check(backend._L.paris_hip_stage_filter(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, C.byref(det_geo)),
          "paris_hip_stage_filter")


def slab_row_band(det_geo, vol_geo, v_dim_x, v_dim_y, v_dim_z, v_offset=0, roi=None):
    """f4: (first row, row count) of the detector band the slab can read (paris_hip_slab_row_band)."""
    r = roi if roi is not None else RegionOfInterest()
    first, count = C.c_uint32(), C.c_uint32()
    check(_lib.load().paris_hip_slab_row_band(C.byref(det_geo), C.byref(vol_geo), v_dim_x, v_dim_y, v_dim_z, v_offset,
                                              int(roi is not None), C.byref(r), C.byref(first), C.byref(count)),
          "paris_hip_slab_row_band")
    return first.value, count.value


def weight_rows(backend, p, det_geo, row_first, row_count):
    """paris::weight on the rows of a band only (paris_hip_stage_weight_rows)"""
    check(backend._L.paris_hip_stage_weight_rows(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, row_first, row_count,
                                                 C.byref(det_geo)), "paris_hip_stage_weight_rows")


def filter_rows(backend, p, det_geo, row_first, row_count):
    """paris::filter on the rows of a band only (paris_hip_stage_filter_rows)"""
    check(backend._L.paris_hip_stage_filter_rows(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, row_first, row_count,
                                                 C.byref(det_geo)), "paris_hip_stage_filter_rows")


def weight_filter_rows(backend, p, det_geo, row_first, row_count, half_ptr=None, half_pitch=0):
    """paris::weight + paris::filter of a row band in one launch (paris_hip_stage_weight_filter_rows); half_ptr: store the
    filtered band as IEEE half there instead of fp32 in place"""
    check(backend._L.paris_hip_stage_weight_filter_rows(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, row_first, row_count,
                                                        C.byref(det_geo), half_ptr, half_pitch), "paris_hip_stage_weight_filter_rows")


def weight_filter_batch(backend, ptr, pitch, frame_stride, n_frames, dim_x, dim_y, det_geo, row_first, row_count, half_ptr=None,
                        half_pitch=0, half_frame_stride=0):
    """paris::weight + paris::filter of a row band for a group of n_frames projections frame_stride bytes apart, in ONE launch
    (paris_hip_stage_weight_filter_batch); bit-identical to the per-frame calls"""
    check(backend._L.paris_hip_stage_weight_filter_batch(backend._ctx, ptr, pitch, frame_stride, n_frames, dim_x, dim_y, row_first,
                                                         row_count, C.byref(det_geo), half_ptr, half_pitch, half_frame_stride),
          "paris_hip_stage_weight_filter_batch")


def backproject(backend, p, v, v_offset, det_geo, vol_geo, enable_angles, enable_roi, roi):
    """paris::backproject (src/backprojection.cpp:37-69)"""
    r = roi if roi is not None else RegionOfInterest()
    check(backend._L.paris_hip_stage_backproject(backend._ctx, p.ptr, p.pitch, p.dim_x, p.dim_y, p.idx, p.phi,
                                                 v.ptr, v.dim_x, v.dim_y, v.dim_z, v_offset, C.byref(det_geo),
                                                 C.byref(vol_geo), int(bool(enable_angles)),
                                                 int(bool(enable_roi)), C.byref(r)),
          "paris_hip_stage_backproject")


def stage_angle(det_geo, idx, enable_angles=False, phi=0.0):
    s, c = C.c_float(), C.c_float()
    check(_lib.load().paris_hip_stage_angle(C.byref(det_geo), idx, int(bool(enable_angles)), phi, C.byref(s),
                                            C.byref(c)), "paris_hip_stage_angle")
    return s.value, c.value
