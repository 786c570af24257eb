"""Optional test helper: the FFTW3 single-precision interface exported by MKL's libmkl_rt.so, through ctypes.

The survey's known-answer run of the reference linked this same library for the reference's FFTW calls
(SURVEY.md section 8c). Plugging it into the oracle for the third-party transform lets the oracle reproduce the
reference's volume checksums bit for bit. Only tests use this; it is absent from the product and skipped where
libmkl_rt.so does not exist.
"""
import ctypes as C
import os

import numpy as np

_CANDIDATES = ["/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.2"]
FFTW_MEASURE = 0
FFTW_DESTROY_INPUT = 1 << 0
FFTW_PRESERVE_INPUT = 1 << 4


def available():
    return any(os.path.exists(p) for p in _CANDIDATES)


_mkl = None


def _lib():
    global _mkl
    if _mkl is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        path = next(p for p in _CANDIDATES if os.path.exists(p))
        m = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, ip = C.c_void_p, C.POINTER(C.c_int)
        m.fftwf_plan_dft_r2c_1d.restype = vp
        m.fftwf_plan_dft_r2c_1d.argtypes = [C.c_int, vp, vp, C.c_uint]
        many = [C.c_int, ip, C.c_int, vp, ip, C.c_int, C.c_int, vp, ip, C.c_int, C.c_int, C.c_uint]
        m.fftwf_plan_many_dft_r2c.restype = vp
        m.fftwf_plan_many_dft_r2c.argtypes = many
        m.fftwf_plan_many_dft_c2r.restype = vp
        m.fftwf_plan_many_dft_c2r.argtypes = many
        m.fftwf_execute.argtypes = [vp]
        _mkl = m
    return _mkl


def make_filter(O, size, tau):
    """openmp::make_filter (src/openmp/filtering.cpp:139-165) with MKL doing the r2c."""
    m = _lib()
    r = np.zeros(size, np.float32)
    k = np.zeros(size // 2 + 1, np.complex64)
    plan = m.fftwf_plan_dft_r2c_1d(size, r.ctypes.data, k.ctypes.data, FFTW_MEASURE | FFTW_PRESERVE_INPUT)
    r[:] = O.make_filter_real(size, tau)  # input initialised after planning, as the reference does (:143-151)
    m.fftwf_execute(plan)
    return O.make_filter_from_spectrum(k, size, tau)


class RowFilter:
    """openmp::apply_filter (src/openmp/filtering.cpp:167-219) with MKL doing the batched r2c / c2r."""

    def __init__(self, fsize, n_col):
        m = _lib()
        self.fs, self.n_col = fsize, n_col
        st = fsize // 2 + 1
        self.exp = np.zeros((n_col, fsize), np.float32)
        self.tr = np.zeros((n_col, st), np.complex64)
        n, e1, e2 = C.c_int(fsize), C.c_int(fsize), C.c_int(st)
        self.fwd = m.fftwf_plan_many_dft_r2c(1, C.byref(n), n_col, self.exp.ctypes.data, C.byref(e1), 1, fsize,
                                             self.tr.ctypes.data, C.byref(e2), 1, st,
                                             FFTW_MEASURE | FFTW_PRESERVE_INPUT)
        self.inv = m.fftwf_plan_many_dft_c2r(1, C.byref(n), n_col, self.tr.ctypes.data, C.byref(e2), 1, st,
                                             self.exp.ctypes.data, C.byref(e1), 1, fsize,
                                             FFTW_MEASURE | FFTW_DESTROY_INPUT)

    def apply(self, p, k):
        m = _lib()
        n_row = p.shape[1]
        self.exp[:] = 0
        self.exp[:, :n_row] = p                                   # expand :75-90
        m.fftwf_execute(self.fwd)
        v = self.tr.view(np.float32).reshape(self.n_col, -1, 2)
        v[:, :, 0] *= k[None, :]                                  # do_filtering :92-105
        v[:, :, 1] *= k[None, :]
        m.fftwf_execute(self.inv)
        p[:] = self.exp[:, :n_row]                                # shrink :107-118
        p /= np.float32(self.fs)                                  # normalize :120-131
        return p
