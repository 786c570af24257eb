"""Loader and ctypes signatures of paris_amd/lib/libparis_hip.so (the C ABI of include/paris_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the backend raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# the product library; PARIS_HIP_LIBRARY names another build of the same C ABI -- the experiments build of `make EXPERIMENTS=1`
# (lib/libparis_hip_experiments.so: the kernels, tile orders and switches that lost their A/B runs) for tools/ and the variant tests
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "lib", "libparis_hip_experiments.so")
LIB_PATH = os.environ.get("PARIS_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libparis_hip.so")

SUCCESS = 0
ERROR_INVALID_ARGUMENT = 10001
ERROR_NO_DEVICE = 10002
ERROR_UNSUPPORTED = 10003
CTX_DEFAULT = 0
CTX_SYNCHRONOUS = 1
CTX_LEGACY_STREAM = 2
CTX_WARM = 4


class DetectorGeometry(C.Structure):
    """paris::detector_geometry (src/geometry.h:30-46)"""
    _fields_ = [("n_row", C.c_uint32), ("n_col", C.c_uint32),
                ("l_px_row", C.c_float), ("l_px_col", C.c_float),
                ("delta_s", C.c_float), ("delta_t", C.c_float),
                ("d_so", C.c_float), ("d_od", C.c_float),
                ("delta_phi", C.c_float)]


class VolumeGeometry(C.Structure):
    """paris::volume_geometry (src/geometry.h:48-57)"""
    _fields_ = [("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
                ("l_vx_x", C.c_float), ("l_vx_y", C.c_float), ("l_vx_z", C.c_float)]


class SubvolumeGeometry(C.Structure):
    """paris::subvolume_geometry (src/geometry.h:59-69)"""
    _fields_ = [("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
                ("remainder", C.c_uint32)]


class RegionOfInterest(C.Structure):
    """paris::region_of_interest (src/region_of_interest.h:30-38)"""
    _fields_ = [("x1", C.c_uint32), ("x2", C.c_uint32), ("y1", C.c_uint32),
                ("y2", C.c_uint32), ("z1", C.c_uint32), ("z2", C.c_uint32)]


class SubvolumeInfo(C.Structure):
    """paris::subvolume_info (src/subvolume_information.h:30-34)"""
    _fields_ = [("geo", SubvolumeGeometry), ("num", C.c_int)]


_vp = C.c_void_p
_u32 = C.c_uint32
_f = C.c_float
_sz = C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/paris_hip.h declares
SIGNATURES = {
    "paris_hip_device_count": (C.c_int, [_P(C.c_int)]),
    "paris_hip_ctx_create": (C.c_int, [C.c_int, _vp, C.c_uint, _P(_vp)]),
    "paris_hip_ctx_destroy": (C.c_int, [_vp]),
    "paris_hip_ctx_synchronize": (C.c_int, [_vp]),
    "paris_hip_ctx_stream": (_vp, [_vp]),
    "paris_hip_fence_create": (C.c_int, [_vp, _P(_vp)]),
    "paris_hip_fence_record": (C.c_int, [_vp, _vp]),
    "paris_hip_fence_wait": (C.c_int, [_vp, _vp]),
    "paris_hip_fence_destroy": (C.c_int, [_vp, _vp]),
    "paris_hip_malloc_projection": (C.c_int, [_vp, _u32, _u32, _P(_vp), _P(_sz)]),
    "paris_hip_malloc_volume": (C.c_int, [_vp, _u32, _u32, _u32, _P(_vp)]),
    "paris_hip_free": (C.c_int, [_vp, _vp]),
    "paris_hip_malloc_host": (C.c_int, [_vp, _sz, _P(_vp)]),
    "paris_hip_free_host": (C.c_int, [_vp, _vp]),
    "paris_hip_memcpy_projection_h2d": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u32, _u32]),
    "paris_hip_upload_projection": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u32, _u32]),
    "paris_hip_memcpy_projection_d2h": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u32, _u32]),
    "paris_hip_memcpy_volume_h2d": (C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32]),
    "paris_hip_memcpy_volume_d2h": (C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32]),
    "paris_hip_memset_volume": (C.c_int, [_vp, _vp, _u32, _u32, _u32]),
    "paris_hip_make_subvolume_information": (C.c_int, [_P(VolumeGeometry), _P(DetectorGeometry), C.c_int,
                                                       _P(SubvolumeInfo)]),
    "paris_hip_make_subvolume_information_reserving": (C.c_int, [_P(VolumeGeometry), _P(DetectorGeometry), C.c_int, _sz,
                                                                 _P(SubvolumeInfo)]),
    "paris_hip_device_memory": (C.c_int, [C.c_int, _P(_sz), _P(_sz)]),
    "paris_hip_weight": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _f, _f, _f, _f, _f]),
    "paris_hip_weight_rows": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _f, _f, _f, _f, _f]),
    "paris_hip_make_filter": (C.c_int, [_vp, _u32, _f, _P(_vp)]),
    "paris_hip_make_filter_windowed": (C.c_int, [_vp, _u32, _f, C.c_int, _P(_vp)]),
    "paris_hip_set_filter_window": (C.c_int, [_vp, C.c_int]),
    "paris_hip_apply_filter": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _vp, _u32, _u32]),
    "paris_hip_set_filter_variant": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_stage_fusion": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_backproject_skip_invalid": (C.c_int, [_vp, C.c_int]),
    "paris_hip_volume_mark_dirty": (C.c_int, [_vp, _vp, _sz]),
    "paris_hip_volume_mark_clean": (C.c_int, [_vp, _vp, _sz]),
    "paris_hip_set_filter_deferral": (C.c_int, [_vp, C.c_int]),
    "paris_hip_volume_scan_clean": (C.c_int, [_vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    "paris_hip_weight_filter_rows": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _f, _f, _f, _f, _f, _vp, _u32, _vp, _sz]),
    "paris_hip_weight_filter_batch": (C.c_int, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _u32, _u32, _f, _f, _f, _f, _f, _vp, _u32, _vp, _sz, _sz]),
    "paris_hip_stage_weight_filter_batch": (C.c_int, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _u32, _u32, _P(DetectorGeometry), _vp, _sz, _sz]),
    "paris_hip_backproject": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _vp, _u32, _u32, _u32, _u32,
                                        _P(DetectorGeometry), _P(VolumeGeometry), C.c_int,
                                        _P(RegionOfInterest), _f, _f, _f, _f]),
    "paris_hip_backproject_f16": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _vp, _u32, _u32, _u32, _u32,
                                            _P(DetectorGeometry), _P(VolumeGeometry), C.c_int,
                                            _P(RegionOfInterest), _f, _f, _f, _f]),
    "paris_hip_convert_projection_f16": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _u32, _u32]),
    "paris_hip_backproject_batch": (C.c_int, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32,
                                              _P(DetectorGeometry), _P(VolumeGeometry), C.c_int,
                                              _P(RegionOfInterest), _P(_f), _P(_f), _f, _f]),
    "paris_hip_backproject_batch_f16": (C.c_int, [_vp, _vp, _sz, _sz, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32,
                                              _P(DetectorGeometry), _P(VolumeGeometry), C.c_int,
                                              _P(RegionOfInterest), _P(_f), _P(_f), _f, _f]),
    "paris_hip_calculate_volume_geometry": (C.c_int, [_P(DetectorGeometry), _P(VolumeGeometry)]),
    "paris_hip_apply_roi": (C.c_int, [_P(VolumeGeometry), _P(RegionOfInterest), _P(VolumeGeometry)]),
    "paris_hip_stage_weight": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _P(DetectorGeometry)]),
    "paris_hip_filter_size": (_u32, [_u32]),
    "paris_hip_stage_filter": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _P(DetectorGeometry)]),
    "paris_hip_stage_weight_rows": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _P(DetectorGeometry)]),
    "paris_hip_stage_filter_rows": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _P(DetectorGeometry)]),
    "paris_hip_stage_weight_filter_rows": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _P(DetectorGeometry), _vp, _sz]),
    "paris_hip_set_backproject_deferral": (C.c_int, [_vp, _u32]),
    "paris_hip_flush": (C.c_int, [_vp]),
    "paris_hip_lean_division_is_exact": (C.c_int, [_vp, _f, _f, _P(C.c_int)]),
    "paris_hip_lean_weighting_is_exact": (C.c_int, [_vp, _f, _f, _f, _P(C.c_int)]),
    "paris_hip_set_lean_validation": (C.c_int, [_vp, C.c_int]),
    "paris_hip_pending_backprojections": (C.c_int, [_vp, _P(_u32), _P(C.c_void_p)]),
    "paris_hip_set_backproject_overlap": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_backproject_references": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_async_validation": (C.c_int, [_vp, C.c_int]),
    "paris_hip_has_experiments": (C.c_int, []),
    "paris_hip_projection_reserve_bytes": (C.c_int, [_vp, _u32, _u32, _P(_sz)]),
    "paris_hip_slab_row_band": (C.c_int, [_P(DetectorGeometry), _P(VolumeGeometry), _u32, _u32, _u32, _u32, C.c_int,
                                          _P(RegionOfInterest), _P(_u32), _P(_u32)]),
    "paris_hip_stage_angle": (C.c_int, [_P(DetectorGeometry), _u32, C.c_int, _f, _P(_f), _P(_f)]),
    "paris_hip_stage_backproject": (C.c_int, [_vp, _vp, _sz, _u32, _u32, _u32, _f, _vp, _u32, _u32, _u32, _u32,
                                              _P(DetectorGeometry), _P(VolumeGeometry), C.c_int, C.c_int,
                                              _P(RegionOfInterest)]),
    "paris_hip_strerror": (C.c_char_p, [C.c_int]),
    "paris_hip_version": (C.c_char_p, []),
    "paris_hip_last_backproject_ms": (C.c_int, [_vp, _P(_f)]),
    "paris_hip_backproject_timing_arm": (C.c_int, [_vp, _u32]),
    "paris_hip_backproject_timing_collect": (C.c_int, [_vp, _P(_f), _u32, _P(_u32)]),
    "paris_hip_set_backproject_variant": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_backproject_tuning": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "paris_hip_set_backproject_order": (C.c_int, [_vp, C.c_int, C.c_int]),
    "paris_hip_set_backproject_slice_shape": (C.c_int, [_vp, C.c_int, C.c_int]),
    "paris_hip_set_backproject_fast_division": (C.c_int, [_vp, C.c_int]),
    "paris_hip_set_backproject_vector_staging": (C.c_int, [_vp, C.c_int]),
    "paris_hip_fast_division_is_exact": (C.c_int, [_vp, _f, _P(C.c_int)]),
}

_lib = None


def load():
    """Loads libparis_hip.so; raises if it is missing (the product has no CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "paris_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C paris_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        if os.environ.get("PARIS_AMD_NO_TORCH") != "1":
            # PyTorch ships its own libamdhip64.so.7; loading it first makes this library bind to that same
            # runtime instance, so device pointers and streams can be shared with torch (plumbing only).
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def has_experiments():
    """True when the loaded library is the experiments build (make EXPERIMENTS=1)"""
    return bool(load().paris_hip_has_experiments())


class ParisHipError(RuntimeError):
    """Raised for a non-zero status; plays the role of paris::stage_runtime_error (src/exception.h:37-41)."""

    def __init__(self, status, where):
        self.status = status
        msg = load().paris_hip_strerror(status)
        super().__init__("%s failed: %s (status %d)" % (where, msg.decode() if msg else "?", status))


def check(status, where):
    if status != SUCCESS:
        raise ParisHipError(status, where)
