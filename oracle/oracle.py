"""ctypes binding of the parity oracle (oracle/libparis_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; nothing under paris_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libparis_oracle.so")


class DetectorGeometry(C.Structure):
    """src/geometry.h:30-46"""
    _fields_ = [("n_row", C.c_uint32), ("n_col", C.c_uint32),
                ("l_px_row", C.c_float), ("l_px_col", C.c_float),
                ("delta_s", C.c_float), ("delta_t", C.c_float),
                ("d_so", C.c_float), ("d_od", C.c_float),
                ("delta_phi", C.c_float)]


class VolumeGeometry(C.Structure):
    """src/geometry.h:48-57"""
    _fields_ = [("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
                ("l_vx_x", C.c_float), ("l_vx_y", C.c_float), ("l_vx_z", C.c_float)]


class RegionOfInterest(C.Structure):
    """src/region_of_interest.h:30-38"""
    _fields_ = [("x1", C.c_uint32), ("x2", C.c_uint32), ("y1", C.c_uint32),
                ("y2", C.c_uint32), ("z1", C.c_uint32), ("z2", C.c_uint32)]


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "paris_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
_fp = C.POINTER(C.c_float)


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box shows
    256 logical CPUs but grants a 16-core quota; OpenMP's default of one thread per visible CPU then thrashes)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.po_calculate_volume_geometry.argtypes = [C.POINTER(DetectorGeometry), C.POINTER(VolumeGeometry)]
        L.po_apply_roi.argtypes = [C.POINTER(VolumeGeometry), C.POINTER(RegionOfInterest),
                                   C.POINTER(VolumeGeometry)]
        L.po_weight_constants.argtypes = [C.POINTER(DetectorGeometry), _fp, _fp, _fp]
        L.po_weight.argtypes = [_fp, C.c_uint32, C.c_uint32] + [C.c_float] * 5
        L.po_filter_size.argtypes = [C.c_uint32]
        L.po_filter_size.restype = C.c_uint32
        L.po_make_filter_real.argtypes = [_fp, C.c_uint32, C.c_float]
        L.po_make_filter.argtypes = [_fp, C.c_uint32, C.c_float]
        L.po_make_filter_from_spectrum.argtypes = [_fp, _fp, C.c_uint32, C.c_float]
        L.po_apply_filter.argtypes = [_fp, C.c_uint32, C.c_uint32, _fp, C.c_uint32]
        L.po_backproject_constants.argtypes = [C.POINTER(DetectorGeometry), C.c_uint32, C.c_int, C.c_float,
                                               _fp, _fp, _fp, _fp]
        L.po_backproject.argtypes = [_fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                     _fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(DetectorGeometry), C.POINTER(VolumeGeometry),
                                     C.c_int, C.POINTER(RegionOfInterest)] + [C.c_float] * 4
        L.po_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.po_fnv1a64.restype = C.c_uint64
        L.po_lcg_fill.argtypes = [_fp, C.c_size_t, C.c_uint32]
        L.po_num_threads.restype = C.c_int
        L.po_set_num_threads.argtypes = [C.c_int]
        L.po_set_num_threads(usable_cores())
        _lib = L
    return _lib


def _f32(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def calculate_volume_geometry(det):
    out = VolumeGeometry()
    lib().po_calculate_volume_geometry(C.byref(det), C.byref(out))
    return out


def apply_roi(vol_geo, roi):
    out = VolumeGeometry()
    lib().po_apply_roi(C.byref(vol_geo), C.byref(roi), C.byref(out))
    return out


def weight_constants(det):
    h, v, d = C.c_float(), C.c_float(), C.c_float()
    lib().po_weight_constants(C.byref(det), C.byref(h), C.byref(v), C.byref(d))
    return h.value, v.value, d.value


def weight(p, det):
    """paris::weight (src/weighting.cpp:32-45) on a (n_col, n_row) float32 array, in place."""
    h_min, v_min, d_sd = weight_constants(det)
    lib().po_weight(_f32(p), p.shape[1], p.shape[0], h_min, v_min, d_sd, det.l_px_row, det.l_px_col)
    return p


def filter_size(n_row):
    return int(lib().po_filter_size(n_row))


def make_filter_real(size, tau):
    r = np.empty(size, np.float32)
    lib().po_make_filter_real(_f32(r), size, tau)
    return r


def make_filter(size, tau):
    k = np.empty(size // 2 + 1, np.float32)
    lib().po_make_filter(_f32(k), size, tau)
    return k


def make_filter_from_spectrum(spec, size, tau):
    """spec: complex64 array of size/2+1 bins."""
    s = np.ascontiguousarray(spec.astype(np.complex64)).view(np.float32)
    k = np.empty(size // 2 + 1, np.float32)
    lib().po_make_filter_from_spectrum(_f32(s), _f32(k), size, tau)
    return k


def apply_filter(p, k, fsize):
    lib().po_apply_filter(_f32(p), p.shape[1], p.shape[0], _f32(k), fsize)
    return p


def backproject_constants(det, idx, enable_angles=False, phi=0.0):
    s, c, ds, dt = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    lib().po_backproject_constants(C.byref(det), idx, int(enable_angles), phi,
                                   C.byref(s), C.byref(c), C.byref(ds), C.byref(dt))
    return s.value, c.value, ds.value, dt.value


def backproject(vol, p, v_offset, det, vol_geo, sin, cos, delta_s_mm, delta_t_mm, roi=None):
    """openmp::backproject (src/openmp/backprojection.cpp:156-199); vol is (dim_z, dim_y, dim_x)."""
    r = roi if roi is not None else RegionOfInterest()
    lib().po_backproject(_f32(vol), vol.shape[2], vol.shape[1], vol.shape[0],
                         _f32(p), p.shape[1], p.shape[0], v_offset,
                         C.byref(det), C.byref(vol_geo), int(roi is not None), C.byref(r),
                         sin, cos, delta_s_mm, delta_t_mm)
    return vol


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(lib().po_fnv1a64(a.ctypes.data, a.nbytes))


def lcg_projection(n_row, n_col, idx):
    p = np.empty((n_col, n_row), np.float32)
    lib().po_lcg_fill(_f32(p), p.size, idx)
    return p


def reconstruct(det, vol_geo, n_proj, v_dims=None, v_offset=0, roi=None, projections=None,
                filtered_out=None):
    """The reference's hot loop (src/main.cpp:98-105) over n_proj LCG projections, idx = 0..n_proj-1."""
    if v_dims is None:
        v_dims = (vol_geo.dim_z, vol_geo.dim_y, vol_geo.dim_x)
    vol = np.zeros(v_dims, np.float32)
    fs = filter_size(det.n_row)
    k = make_filter(fs, det.l_px_row)
    for i in range(n_proj):
        p = lcg_projection(det.n_row, det.n_col, i) if projections is None else projections[i].copy()
        weight(p, det)
        apply_filter(p, k, fs)
        if filtered_out is not None:
            filtered_out.append(p.copy())
        s, c, ds, dt = backproject_constants(det, i)
        backproject(vol, p, v_offset, det, vol_geo, s, c, ds, dt, roi)
    return vol
