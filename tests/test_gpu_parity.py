"""Parity of the HIP path (through the C ABI, include/paris_hip.h) against the CPU oracle and the golden fixtures.

Bars (DESIGN.md "Numerics"):
  weighting, backprojection : bit-exact (same fp32 operations in the same order, IEEE divide/sqrt, no FMA)
  make_filter / apply_filter: the FFT is a third-party library in the reference (FFTW3f), so parity is to FFT
                              rounding: max-abs error <= 1e-5 * max|x| and relative L2 error <= 1e-5
"""
import ctypes as C
import os

import numpy as np
import pytest

from paris_amd import _lib
from paris_amd import backend as B

pytestmark = pytest.mark.gpu

FILTER_TOL = 1e-5

KAT = (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)


@pytest.fixture(scope="module")
def be():
    b = B.set_device(B.get_devices()[0])
    yield b
    b.close()


@pytest.fixture(scope="module")
def kat_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "kat.npz"))


def to_device(be, host_array, idx=0, phi=0.0):
    h = B.Projection(np.ascontiguousarray(host_array, np.float32), host_array.shape[1], host_array.shape[0], idx, phi)
    return B.load(be, h)


def to_host(be, d_p):
    h = be.make_projection_host(d_p.dim_x, d_p.dim_y)
    be.copy_d2h(d_p, h)
    return h.buf


def volume_to_host(be, d_v):
    h = be.make_volume_host(d_v.dim_x, d_v.dim_y, d_v.dim_z)
    be.copy_d2h(d_v, h)
    return h.buf


def product_refuses(call, *args):
    """Kernels, tile orders and switches that lost their A/B runs are compiled into the experiments build only (make EXPERIMENTS=1,
    PARIS_HIP_LIBRARY=.../libparis_hip_experiments.so: tests/test_gpu_experiments_build.py runs these tests against it). Against the
    product library the setter must answer PARIS_HIP_ERROR_UNSUPPORTED -- then there is nothing more to test: returns True."""
    if _lib.has_experiments():
        return False
    with pytest.raises(_lib.ParisHipError) as e:
        call(*args)
    assert e.value.status == _lib.ERROR_UNSUPPORTED
    return True


def rel_l2(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300))


# ---- weighting ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("g", [KAT, (512, 384, 0.2, 0.2, 0, 0, 500, 500, 1.0),
                               (333, 217, 0.127, 0.254, -3.25, 2.5, 321.5, 123.25, 0.7)])
def test_weight_bit_exact(be, oracle, g):
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    p = oracle.lcg_projection(det.n_row, det.n_col, 3)
    d_p = to_device(be, p)
    B.weight(be, d_p, det)  # paris::weight wrapper
    got = to_host(be, d_p)
    want = oracle.weight(p.copy(), odet)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    be.free(d_p)


def test_weight_golden(be, oracle, kat_golden):
    det = B.DetectorGeometry(*KAT)
    d_p = to_device(be, oracle.lcg_projection(64, 48, 0))
    B.weight(be, d_p, det)
    assert np.array_equal(to_host(be, d_p), kat_golden["weighted_p0"])
    be.free(d_p)


def test_weight_backend_level_call(be, oracle):
    """backend::weight with explicit constants (src/openmp/weighting.cpp:32) == wrapper"""
    odet = oracle.DetectorGeometry(*KAT)
    h_min, v_min, d_sd = oracle.weight_constants(odet)
    p = oracle.lcg_projection(64, 48, 1)
    d_p = to_device(be, p)
    be.weight(d_p, h_min, v_min, d_sd, odet.l_px_row, odet.l_px_col)
    assert np.array_equal(to_host(be, d_p), oracle.weight(p.copy(), odet))
    be.free(d_p)


# ---- filtering ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,tau", [(8, 0.5), (128, 0.2), (1024, 0.2), (2048, 0.127), (4096, 0.2), (16384, 0.1)])
def test_make_filter(be, oracle, n, tau):
    k = be.make_filter(n, tau)
    got = be.filter_to_host(k)
    want = oracle.make_filter(n, tau)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
    r = oracle.make_filter_real(n, tau).astype(np.float64)
    exact = tau * np.abs(np.fft.rfft(r))
    assert np.max(np.abs(got - exact)) <= FILTER_TOL * exact.max()
    be.free(k)


@pytest.mark.parametrize("n,tau", [(128, 0.2), (2048, 0.2), (4096, 0.127)])
def test_shepp_logan_window_extension(be, oracle, n, tau):
    """Extension (the reference has the ramp only): K_ramp[f] * sinc(pi f / N), against float64."""
    k_ramp = oracle.make_filter(n, tau).astype(np.float64)
    f = np.arange(n // 2 + 1, dtype=np.float64)
    x = np.pi * f / n
    want = k_ramp * np.where(f == 0, 1.0, np.sin(x) / np.where(f == 0, 1.0, x))
    k = be.make_filter(n, tau, window=1)
    got = be.filter_to_host(k)
    be.free(k)
    assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
    assert abs(got[n // 2] / k_ramp[n // 2] - 2 / np.pi) < 1e-5
    # the stage wrapper follows the ctx's window and rebuilds its cached K when it changes
    det = B.DetectorGeometry(64, 48, 0.2, 0.25, 0, 0, 100, 200, 45)
    p = oracle.lcg_projection(64, 48, 3)
    outs = []
    for window in (0, 1, 0):
        be.set_filter_window(window)
        d_p = to_device(be, p)
        B.filter(be, d_p, det)
        outs.append(to_host(be, d_p))
        be.free(d_p)
    assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
    ref = p.astype(np.float64)
    kk = oracle.make_filter(128, 0.2).astype(np.float64)
    ff = np.arange(65)
    kk_sl = kk * np.where(ff == 0, 1.0, np.sin(np.pi * ff / 128) / np.where(ff == 0, 1.0, np.pi * ff / 128))
    want_rows = np.fft.irfft(np.fft.rfft(ref, 128, axis=1) * kk_sl, 128, axis=1)[:, :64]
    assert np.max(np.abs(outs[1] - want_rows)) <= FILTER_TOL * np.abs(want_rows).max()


def test_make_filter_golden(be, kat_golden):
    for n in (128, 1024, 2048, 4096):
        k = be.make_filter(n, 0.2)
        want = kat_golden["k_%d" % n]
        assert np.max(np.abs(be.filter_to_host(k) - want)) <= FILTER_TOL * want.max()
        be.free(k)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("n_row,n_col", [(64, 48), (512, 37), (1000, 8), (5, 3), (2048, 6), (4097, 2), (3000, 5), (8192, 3)])
def test_apply_filter(be, oracle, n_row, n_col, variant):
    fs = oracle.filter_size(n_row)
    assert B.filter_size(n_row) == fs
    tau = 0.2
    p = oracle.lcg_projection(n_row, n_col, 11) - np.float32(0.25)
    want = oracle.apply_filter(p.copy(), oracle.make_filter(fs, tau), fs)
    if variant == 2 and product_refuses(be.set_filter_variant, 2):
        return
    d_p = to_device(be, p)
    k = be.make_filter(fs, tau)
    be.set_filter_variant(variant)  # 0: radix-16 passes with table twiddles (N >= 1024), 1: radix-2 in LDS, 2: first radix-16 kernel (experiments build)
    try:
        be.apply_filter(d_p, k, fs, n_col)
    finally:
        be.set_filter_variant(0)
    got = to_host(be, d_p)
    assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
    assert rel_l2(got, want) <= FILTER_TOL
    # float64 reference of the same operation
    kk = tau * np.abs(np.fft.rfft(oracle.make_filter_real(fs, tau).astype(np.float64)))
    exact = np.fft.irfft(np.fft.rfft(p.astype(np.float64), fs, axis=1) * kk[None, :], fs, axis=1)[:, :n_row]
    assert rel_l2(got, exact) <= FILTER_TOL
    be.free(d_p)
    be.free(k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARIS_FILTER_FUZZ_SEEDS", "10"))))
def test_weight_and_filter_random_sizes(be, oracle, seed):
    """Seeded random detector sizes (odd and even row counts, widths on both sides of the power-of-two boundaries where the
    filter switches kernels and lengths), pixel pitches and offsets: weighting bit-exact, filter within FFT rounding of the
    oracle and of a float64 FFT, through the stage wrappers and through a row band of the same frame."""
    rng = np.random.default_rng(4000 + seed)
    n_row = int(rng.choice([rng.integers(3, 64), rng.integers(64, 513), rng.integers(500, 1030), rng.integers(1020, 2060),
                            rng.integers(2040, 4100), 512, 513, 1024, 1025, 2048]))
    n_col = int(rng.integers(1, 70))
    g = (n_row, n_col, float(rng.choice([0.1, 0.2, 0.127, 0.4])), float(rng.choice([0.1, 0.2, 0.25])), float(rng.uniform(-5, 5)),
         float(rng.uniform(-5, 5)), float(rng.uniform(50, 600)), float(rng.uniform(50, 600)), 1.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    p = (oracle.lcg_projection(n_row, n_col, seed) - np.float32(0.4)) * np.float32(rng.uniform(0.5, 200))
    d_p = to_device(be, p)
    B.weight(be, d_p, det)
    want_w = oracle.weight(p.copy(), odet)
    got_w = to_host(be, d_p)
    assert np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32))
    B.filter(be, d_p, det)
    got = to_host(be, d_p)
    fs = oracle.filter_size(n_row)
    want = oracle.apply_filter(want_w.copy(), oracle.make_filter(fs, det.l_px_row), fs)
    scale = max(np.abs(want).max(), 1e-30)
    assert np.max(np.abs(got - want)) <= FILTER_TOL * scale
    kk = det.l_px_row * np.abs(np.fft.rfft(oracle.make_filter_real(fs, det.l_px_row).astype(np.float64)))
    exact = np.fft.irfft(np.fft.rfft(want_w.astype(np.float64), fs, axis=1) * kk[None, :], fs, axis=1)[:, :n_row]
    assert np.max(np.abs(got - exact)) <= FILTER_TOL * scale
    be.free(d_p)
    # the same rows through the band entry points (whole filter row pairs): bit-identical to the full call
    first = 2 * int(rng.integers(0, max(1, n_col // 2)))
    count = min(n_col - first, 2 * int(rng.integers(1, 8)))
    d_q = to_device(be, p)
    B.weight_rows(be, d_q, det, first, count)
    B.filter_rows(be, d_q, det, first, count)
    band = to_host(be, d_q)
    be.free(d_q)
    assert np.array_equal(band[first:first + count].view(np.uint32), got[first:first + count].view(np.uint32))
    assert np.array_equal(band[:first], p[:first]) and np.array_equal(band[first + count:], p[first + count:])


@pytest.fixture(scope="module")
def be_async():
    """a ctx whose calls only enqueue (stage fusion is ignored under PARIS_HIP_CTX_SYNCHRONOUS)"""
    b = B.set_device(B.get_devices()[0], synchronous=False)
    yield b
    b.close()


def stage_constants(det):
    """h_min, v_min, d_sd as paris::weight derives them (src/weighting.cpp:37-42), in fp32"""
    f = np.float32
    h_min = f(det.delta_s) * f(det.l_px_row) - (f(det.n_row) * f(det.l_px_row)) / f(2)
    v_min = f(det.delta_t) * f(det.l_px_col) - (f(det.n_col) * f(det.l_px_col)) / f(2)
    return float(h_min), float(v_min), float(abs(f(det.d_so)) + abs(f(det.d_od)))


@pytest.mark.parametrize("n_row,n_col,fs_override", [(512, 37, 0), (1000, 9, 0), (1024, 16, 0), (2048, 12, 0), (3000, 5, 0),
                                                     (4097, 3, 0), (1500, 6, 2048), (700, 4, 1024)])
def test_fused_weight_filter_equals_the_two_stages(be_async, oracle, n_row, n_col, fs_override):
    """weighting rides along in the row filter's load. One launch must give the bits of the two launches
    (weight kernel, then the same filter kernel without weighting) -- through the held-back weight() + apply_filter() pair
    (stage fusion) and through the explicit paris_hip_weight_filter_rows -- and stay within the filter tolerance of the CPU
    oracle. fs_override: a filter length below 2 * n_row (dim_x > N/2: the kernel variant without the zero-padding shortcuts).
    The half-precision store variant must equal the fp32 result rounded to half, and leave the fp32 rows alone."""
    be = be_async
    g = (n_row, n_col, 0.127, 0.2, 2.5, -1.25, 300.0, 250.0, 1.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    fs = fs_override or oracle.filter_size(n_row)
    p = (oracle.lcg_projection(n_row, n_col, 3) - np.float32(0.3)) * np.float32(17.0)
    h_min, v_min, d_sd = stage_constants(det)
    k = be.make_filter(fs, det.l_px_row)

    be.set_stage_fusion(False)
    d_a = to_device(be, p)
    be.weight(d_a, h_min, v_min, d_sd, det.l_px_row, det.l_px_col)
    got_w = to_host(be, d_a)
    want_w = oracle.weight(p.copy(), odet)
    assert np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32))
    be.apply_filter(d_a, k, fs, n_col)
    two = to_host(be, d_a)
    want = oracle.apply_filter(want_w.copy(), oracle.make_filter(fs, det.l_px_row), fs)
    assert np.max(np.abs(two - want)) <= FILTER_TOL * np.abs(want).max() and rel_l2(two, want) <= FILTER_TOL

    be.set_stage_fusion(True)
    try:
        d_b = to_device(be, p)
        be.weight(d_b, h_min, v_min, d_sd, det.l_px_row, det.l_px_col)   # held back
        be.apply_filter(d_b, k, fs, n_col)                                 # one launch
        one = to_host(be, d_b)
        assert np.array_equal(one.view(np.uint32), two.view(np.uint32))
        # a held-back weighting that no filter picks up is run by whatever looks at the projection next
        d_c = to_device(be, p)
        be.weight(d_c, h_min, v_min, d_sd, det.l_px_row, det.l_px_col)
        assert np.array_equal(to_host(be, d_c).view(np.uint32), want_w.view(np.uint32))
        # ... and a filter call on OTHER rows does not swallow it
        if n_col >= 4:
            d_d = to_device(be, p)
            B._lib.check(be._L.paris_hip_weight_rows(be._ctx, d_d.ptr, d_d.pitch, n_row, n_col, 0, 2, h_min, v_min, d_sd, det.l_px_row,
                                                    det.l_px_col), "weight_rows")
            B._lib.check(be._L.paris_hip_apply_filter(be._ctx, d_d.ptr + 2 * d_d.pitch, d_d.pitch, n_row, 2, k.ptr, fs, 2), "apply_filter")
            mixed = to_host(be, d_d)
            assert np.array_equal(mixed[:2].view(np.uint32), want_w[:2].view(np.uint32))       # weighted only
            raw_f = oracle.apply_filter(p[2:4].copy(), oracle.make_filter(fs, det.l_px_row), fs)
            assert np.max(np.abs(mixed[2:4] - raw_f)) <= FILTER_TOL * np.abs(raw_f).max()        # filtered only
            assert np.array_equal(mixed[4:], p[4:])
            be.free(d_d)
        be.free(d_b)
        be.free(d_c)
    finally:
        be.set_stage_fusion(False)

    # explicit one-launch entry point, a band of whole row pairs; rows outside stay raw
    first = 2 * (n_col // 4)
    count = min(n_col - first, 6)
    d_e = to_device(be, p)
    be.weight_filter_rows(d_e, first, count, h_min, v_min, d_sd, det.l_px_row, det.l_px_col, k, fs)
    band = to_host(be, d_e)
    assert np.array_equal(band[first:first + count].view(np.uint32), two[first:first + count].view(np.uint32))
    assert np.array_equal(band[:first], p[:first]) and np.array_equal(band[first + count:], p[first + count:])
    # half-precision store
    import torch
    d_f = to_device(be, p)
    half = torch.full((n_col, n_row + 3), -3.0, dtype=torch.float16, device="cuda:%d" % be.device)
    be.weight_filter_rows(d_f, 0, n_col, h_min, v_min, d_sd, det.l_px_row, det.l_px_col, k, fs, half.data_ptr(), half.stride(0) * 2)
    be.synchronize()
    h = half.cpu().numpy()
    assert np.array_equal(h[:, :n_row].view(np.uint16), two.astype(np.float16).view(np.uint16))
    assert np.all(h[:, n_row:] == np.float16(-3.0))
    assert np.array_equal(to_host(be, d_f), p)  # the fp32 rows were only read
    for d in (d_a, d_e, d_f):
        be.free(d)
    be.free(k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARIS_FILTER_FUZZ_SEEDS", "10"))))
def test_fused_weight_filter_random_sizes(be_async, oracle, seed):
    """Seeded random detector sizes for the one-launch weight + filter (filter lengths 1024 ... 8192, widths on both sides of the
    power-of-two boundaries, odd row counts, random pitches / offsets / distances, a random band of whole row pairs): the held-back
    weight() + filter() pair through the stage wrappers equals the two launches bit for bit and the oracle within the filter
    tolerance; rows outside the band stay raw."""
    be = be_async
    rng = np.random.default_rng(7000 + seed)
    n_row = int(rng.choice([rng.integers(257, 513), rng.integers(500, 1030), rng.integers(1020, 2060), rng.integers(2040, 4097),
                            512, 513, 1024, 1025, 2048, 2049]))
    n_col = int(rng.integers(1, 40))
    g = (n_row, n_col, float(rng.choice([0.1, 0.2, 0.127, 0.4])), float(rng.choice([0.1, 0.2, 0.25])), float(rng.uniform(-5, 5)),
         float(rng.uniform(-5, 5)), float(rng.uniform(50, 600)), float(rng.uniform(50, 600)), 1.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    fs = oracle.filter_size(n_row)
    assert fs >= 1024
    p = (oracle.lcg_projection(n_row, n_col, seed) - np.float32(0.4)) * np.float32(rng.uniform(0.5, 200))
    first = 2 * int(rng.integers(0, max(1, n_col // 2)))
    count = min(n_col - first, 2 * int(rng.integers(1, 12)))

    be.set_stage_fusion(False)
    d_a = to_device(be, p)
    B.weight_rows(be, d_a, det, first, count)
    B.filter_rows(be, d_a, det, first, count)
    two = to_host(be, d_a)
    be.set_stage_fusion(True)
    try:
        d_b = to_device(be, p)
        B.weight_rows(be, d_b, det, first, count)   # held back
        B.filter_rows(be, d_b, det, first, count)   # weights in its load
        one = to_host(be, d_b)
    finally:
        be.set_stage_fusion(False)
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32))
    assert np.array_equal(one[:first], p[:first]) and np.array_equal(one[first + count:], p[first + count:])
    want_w = oracle.weight(p.copy(), odet)
    want = oracle.apply_filter(want_w.copy(), oracle.make_filter(fs, det.l_px_row), fs)[first:first + count]
    scale = max(np.abs(want).max(), 1e-30)
    assert np.max(np.abs(one[first:first + count] - want)) <= FILTER_TOL * scale
    be.free(d_a)
    be.free(d_b)


def test_fused_filter_needs_a_library_filter(be_async, oracle):
    """the one-launch entry point refuses a K it has no permuted copy of, and lengths below 1024 (the caller runs two stages)"""
    be = be_async
    det = B.DetectorGeometry(64, 8, 0.2, 0.2, 0, 0, 100, 200, 1.0)
    d_p = to_device(be, oracle.lcg_projection(64, 8, 1))
    k = be.make_filter(128, 0.2)
    with pytest.raises(B.ParisHipError):
        be.weight_filter_rows(d_p, 0, 8, -6.4, -0.8, 300.0, 0.2, 0.2, k, 128)
    k2 = be.make_filter(1024, 0.2)
    fake = B.FilterBuffer(k.ptr, 1024, be)  # a buffer that is not a 1024-point K of this ctx
    with pytest.raises(B.ParisHipError):
        be.weight_filter_rows(d_p, 0, 8, -6.4, -0.8, 300.0, 0.2, 0.2, fake, 1024)
    be.free(d_p)
    be.free(k)
    be.free(k2)


def test_filter_wrapper_and_golden(be, oracle, kat_golden):
    det = B.DetectorGeometry(*KAT)
    for i in (0, 7):
        d_p = to_device(be, oracle.lcg_projection(64, 48, i))
        B.weight(be, d_p, det)
        B.filter(be, d_p, det)  # paris::filter wrapper (K cached in the ctx)
        got = to_host(be, d_p)
        want = kat_golden["filtered"][i]
        assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
        be.free(d_p)


def test_apply_filter_rejects_bad_arguments(be, oracle):
    d_p = to_device(be, oracle.lcg_projection(64, 4, 0))
    k = be.make_filter(64, 0.2)
    L = _lib.load()
    assert L.paris_hip_apply_filter(be._ctx, d_p.ptr, d_p.pitch, 64, 4, k.ptr, 32, 4) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_apply_filter(be._ctx, d_p.ptr, d_p.pitch, 64, 4, k.ptr, 100, 4) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_apply_filter(be._ctx, d_p.ptr, d_p.pitch, 64, 4, k.ptr, 64, 5) == _lib.ERROR_INVALID_ARGUMENT
    import ctypes as C
    out = C.c_void_p()
    assert L.paris_hip_make_filter(be._ctx, 100, 0.2, C.byref(out)) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_make_filter(be._ctx, 32768, 0.2, C.byref(out)) == _lib.ERROR_INVALID_ARGUMENT
    be.free(d_p)
    be.free(k)


# ---- backprojection --------------------------------------------------------------------------------------------

def hip_backproject_all(be, projections, det, vol_geo, v_dims, v_offset=0, roi=None, enable_angles=False):
    """The reference's hot loop (src/main.cpp:98-105) from already filtered projections."""
    dz, dy, dx = v_dims
    d_v = be.make_volume_device(dx, dy, dz)
    for i, p in enumerate(projections):
        d_p = to_device(be, p, idx=i)
        B.backproject(be, d_p, d_v, v_offset, det, vol_geo, enable_angles, roi is not None, roi)
        be.free(d_p)
    out = volume_to_host(be, d_v)
    be.free(d_v)
    return out


def oracle_backproject_all(oracle, projections, odet, ovg, v_dims, v_offset=0, roi=None):
    vol = np.zeros(v_dims, np.float32)
    for i, p in enumerate(projections):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(vol, p, v_offset, odet, ovg, s, c, ds, dt, roi)
    return vol


def assert_bit_equal(got, want):
    same = got.view(np.uint32) == want.view(np.uint32)
    if not same.all():
        bad = np.argwhere(~same)
        z, y, x = bad[0]
        raise AssertionError("%d of %d voxels differ; first at (x=%d,y=%d,z=%d): got %r want %r" % (
            len(bad), got.size, x, y, z, got[z, y, x], want[z, y, x]))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_backproject_kat_full_bit_exact(be, oracle, kat_golden, variant):
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    if variant == 3 and product_refuses(be.set_backproject_variant, 3):   # the slice kernel
        return
    be.set_backproject_variant(variant)
    try:
        got = hip_backproject_all(be, kat_golden["filtered"], det, vg, (61, 67, 67))
    finally:
        be.set_backproject_variant(0)
    assert_bit_equal(got, kat_golden["volume"])
    ovg = oracle.calculate_volume_geometry(odet)
    assert_bit_equal(got, oracle_backproject_all(oracle, kat_golden["filtered"], odet, ovg, (61, 67, 67)))


def test_backproject_kat_slab_and_roi_bit_exact(be, oracle, kat_golden):
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    full = kat_golden["volume"]
    slab = hip_backproject_all(be, kat_golden["filtered"], det, vg, (31, 67, 67), v_offset=30)
    assert_bit_equal(slab, full[30:61])
    roi = B.RegionOfInterest(8, 40, 4, 36, 10, 30)
    rg = B.apply_roi(vg, 8, 40, 4, 36, 10, 30)
    assert (rg.dim_x, rg.dim_y, rg.dim_z) == (32, 32, 20)
    rv = hip_backproject_all(be, kat_golden["filtered"], det, vg, (20, 32, 32), roi=roi)
    assert_bit_equal(rv, full[10:30, 4:36, 8:40])
    # ROI and slab offset together (src/openmp/backprojection.cpp:105-113): second half of the ROI's slices
    rs = hip_backproject_all(be, kat_golden["filtered"], det, vg, (10, 32, 32), v_offset=10, roi=roi)
    assert_bit_equal(rs, full[20:30, 4:36, 8:40])


@pytest.mark.parametrize("tuning", [dict(vx=1, unroll=1), dict(vx=1, unroll=4), dict(vx=2, unroll=2),
                                    dict(vx=2, unroll=4), dict(vx=4, unroll=1), dict(vx=4, unroll=2),
                                    dict(vx=4, unroll=4), dict(vx=4, unroll=2, tz=8), dict(vx=4, unroll=2, tz=50),
                                    dict(vx=4, unroll=2, lds_bytes=1024), dict(vx=2, unroll=4, lds_bytes=4096),
                                    dict(vx=1, unroll=2, lds_bytes=65536, tz=64)])
def test_backproject_every_kernel_shape_bit_exact(be, oracle, tuning):
    """Noise projections into a 72 x 40 x 45 volume whose dims exercise partial tiles; every lane width, unroll,
    tile depth and LDS budget (down to one that forces the global-memory tap path) gives the oracle's bits."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 40.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(72, 40, 45, nat.l_vx_x * 1.3, nat.l_vx_x * 2.0, nat.l_vx_x * 1.7)
    ovg = oracle.VolumeGeometry(72, 40, 45, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    projs = [oracle.lcg_projection(96, 80, i) - np.float32(0.5) for i in range(9)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, (45, 40, 72))
    if tuning.get("unroll", 0) > 2 and product_refuses(lambda: be.set_backproject_tuning(**tuning)):
        return
    be.set_backproject_variant(2)
    be.set_backproject_tuning(**tuning)
    try:
        got = hip_backproject_all(be, projs, det, vg, (45, 40, 72))
    finally:
        be.set_backproject_tuning()
        be.set_backproject_variant(0)
    assert_bit_equal(got, want)


@pytest.mark.parametrize("order", [0, 1, 5, 8, 9, 12, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("tz,dims", [(8, (300, 150, 200)), (16, (70, 260, 136)), (3, (45, 40, 72)), (8, (128, 150, 200))])
def test_backproject_every_tile_order_bit_exact(be, oracle, order, tz, dims):
    """Every workgroup -> tile mapping of the tile kernel and of the fused kernel (x / z fastest, a contiguous run per XCD,
    a y band per XCD swept x -> z -> y, x -> y -> z, and in chunks of 256 slices; round 3: y tiles dealt to the XCDs singly, in
    pairs, fours and eights in shallow z chunks, z tiles dealt to the XCDs) covers every voxel exactly once: volumes
    whose tile counts are not multiples of 8 (the XCD count) or of the deal's group in y, deeper than one chunk, 16 z tiles (order 18's whole rounds of
    eight planes), 38, 19 and 15 z tiles (round 4: the planes left over after the whole rounds are shared by all XCDs, y tiles dealt),
    5 z tiles (fewer than 8: order 18 falls back to order 5), with a slab offset; bit-identical to the oracle through single launches and through one fused launch."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 40.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    full_z = dims[0] + 20
    vg = B.VolumeGeometry(dims[2], dims[1], full_z, nat.l_vx_x * 0.5, nat.l_vx_x * 0.35, nat.l_vx_x * 0.25)
    ovg = oracle.VolumeGeometry(dims[2], dims[1], full_z, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    if order in (0, 1, 8, 9, 12) and product_refuses(be.set_backproject_order, order, -1):
        return
    projs = [oracle.lcg_projection(96, 80, i) - np.float32(0.5) for i in range(3)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, dims, v_offset=13)
    be.set_backproject_tuning(tz=tz)
    be.set_backproject_order(order, -1)
    try:
        got = hip_backproject_all(be, projs, det, vg, dims, v_offset=13)
        assert_bit_equal(got, want)
        if tz in (8, 16):  # the fused kernel is built for these tile depths
            stack = np.stack(projs)
            d_stack = to_device(be, stack.reshape(3 * 80, 96))
            d_v = be.make_volume_device(dims[2], dims[1], dims[0])
            sc = [B.stage_angle(det, i) for i in range(3)]
            be.backproject_batch(d_stack.ptr, d_stack.pitch, d_stack.pitch * 80, 3, 96, 80, d_v, 13, det, vg, False, None,
                                 [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
            assert_bit_equal(volume_to_host(be, d_v), want)
            be.free(d_v)
            be.free(d_stack)
    finally:
        be.set_backproject_tuning()
        be.set_backproject_order()


def test_deep_volume_nesting_on_small_volumes():
    """Volumes deeper than 512 slices run the dealt orders with another nesting (BpParams::yfast = 2: a dealt group's y tiles before
    the next z tile); the library reads PARIS_TILE_NEST once per process, so the every-order cases above -- uneven tile counts,
    padding workgroups, every dealt group size, the fused batch beside them -- are run once more in a child process with that
    nesting forced on their small volumes."""
    import subprocess
    import sys
    # (PARIS_TILE_NEST is an A/B switch of the experiments build; the product picks the nesting by volume depth alone -- its deep
    # nesting meets the oracle on deep volumes in tests/test_gpu_full_volume.py)
    env = dict(os.environ, PARIS_TILE_NEST="2", PARIS_HIP_LIBRARY=_lib.EXPERIMENTS_LIB_PATH)
    assert os.path.exists(_lib.EXPERIMENTS_LIB_PATH), "build it: make -C paris_amd/csrc EXPERIMENTS=1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "every_tile_order", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tz,order,unroll", [(4, 12, 1), (8, 12, 2), (16, 5, 2), (2, 8, 1)])
def test_backproject_two_pass_variant_bit_exact(be, oracle, tz, order, unroll):
    """Variant 5: the column constants of the whole (x, y) plane are computed once per projection by a first kernel and read by
    the tile kernel (shallow tiles then pay no column setup). Same arithmetic, so the oracle's bits; ROI + slab offset."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 40.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(260, 170, 120, nat.l_vx_x * 0.4, nat.l_vx_x * 0.45, nat.l_vx_x * 0.3)
    ovg = oracle.VolumeGeometry(260, 170, 120, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    roi, oroi = B.RegionOfInterest(20, 220, 10, 160, 5, 100), oracle.RegionOfInterest(20, 220, 10, 160, 5, 100)
    dims = (60, 150, 200)
    if product_refuses(be.set_backproject_variant, 5):
        return
    projs = [oracle.lcg_projection(96, 80, i) - np.float32(0.5) for i in range(3)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, dims, v_offset=17, roi=oroi)
    be.set_backproject_variant(5)
    be.set_backproject_tuning(unroll=unroll, tz=tz)
    be.set_backproject_order(order, -1)
    try:
        got = hip_backproject_all(be, projs, det, vg, dims, v_offset=17, roi=roi)
    finally:
        be.set_backproject_variant(0)
        be.set_backproject_tuning()
        be.set_backproject_order()
    assert_bit_equal(got, want)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("vec", [False, True])
def test_backproject_fast_division_and_staging_switches(be, oracle, kat_golden, fast, vec):
    """The multiply + 2 FMA division by the pixel pitch (used only after the exhaustive per-divisor check) vs the IEEE
    sequence, and 4-pixel vs 1-pixel staging of the detector box: same bits."""
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    be.set_backproject_fast_division(fast)
    be.set_backproject_vector_staging(vec)
    try:
        got = hip_backproject_all(be, kat_golden["filtered"], det, vg, (61, 67, 67))
        roi = B.RegionOfInterest(8, 40, 4, 36, 10, 30)
        rv = hip_backproject_all(be, kat_golden["filtered"], det, vg, (20, 32, 32), roi=roi)  # 16-byte lanes
    finally:
        be.set_backproject_fast_division(True)
        be.set_backproject_vector_staging(True)
    assert_bit_equal(got, kat_golden["volume"])
    assert_bit_equal(rv, kat_golden["volume"][10:30, 4:36, 8:40])


def test_backproject_unaligned_projection_pitch(be, oracle, kat_golden):
    """A projection whose rows are not 16-byte aligned (pitch = 65 floats) takes the 1-pixel staging path."""
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    d_v = be.make_volume_device(67, 67, 61)
    raw = be.make_projection_device(65, 48)  # 65-float rows requested; use a 65-float pitch inside its buffer
    for i in range(8):
        padded = np.zeros((48, 65), np.float32)
        padded[:, :64] = kat_golden["filtered"][i]
        _lib.check(_lib.load().paris_hip_memcpy_volume_h2d(be._ctx, raw.ptr, padded.ctypes.data, 65, 48, 1), "h2d")
        be.synchronize()
        p = be.wrap_projection(raw.ptr, 65 * 4, 64, 48, idx=i)
        B.backproject(be, p, d_v, 0, det, vg, False, False, None)
    assert_bit_equal(volume_to_host(be, d_v), kat_golden["volume"])
    be.free(d_v)
    be.free(raw)


def test_fast_division_exhaustive_check(be):
    """The per-divisor check sweeps all 2^32 dividends on the GPU. Typical pixel pitches pass; whatever it reports,
    the kernel only uses the fast form when it passed. A divisor the check must reject: 0 / negative / inf."""
    for c in (0.2, 0.25, 0.127, 0.4, 0.1, 0.05, 1.0, 0.074):
        assert be.fast_division_is_exact(c) is True, c
    for c in (0.0, -0.2, float("inf")):
        assert be.fast_division_is_exact(c) is False
    # odd divisors: the answer may be either, it just has to come back
    for c in (float(np.float32(1.0) - np.float32(2.0 ** -24)), 1e-38, 3e38, 1e-42):
        assert be.fast_division_is_exact(c) in (True, False)


@pytest.mark.parametrize("shape", [(16, 4), (16, 2), (8, 4), (8, 2), (8, 1)])
@pytest.mark.parametrize("lds_bytes", [0, 1024])
def test_backproject_slice_kernel_shapes_bit_exact(be, oracle, shape, lds_bytes):
    """The slice kernel (one slice per wave, column state shared through LDS) in every shape, on a volume whose
    dims leave partial tiles in x, y and z; lds_bytes = 1024 forces the global-memory tap path."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 40.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    dims = (45, 42, 72)  # z, y, x: 72 = 64 + 8, 42 rows, 45 slices
    vg = B.VolumeGeometry(dims[2], dims[1], dims[0], nat.l_vx_x * 1.3, nat.l_vx_x * 2.0, nat.l_vx_x * 1.7)
    ovg = oracle.VolumeGeometry(dims[2], dims[1], dims[0], vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    if product_refuses(be.set_backproject_slice_shape, *shape):
        assert product_refuses(be.set_backproject_variant, 3)
        return
    projs = [oracle.lcg_projection(96, 80, i) - np.float32(0.5) for i in range(5)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    be.set_backproject_variant(3)
    be.set_backproject_slice_shape(*shape)
    be.set_backproject_tuning(lds_bytes=lds_bytes)
    try:
        for order in (0, 1, 5, 8):
            be.set_backproject_order(order, 1)
            got = hip_backproject_all(be, projs, det, vg, dims)
            assert_bit_equal(got, want)
    finally:
        be.set_backproject_variant(0)
        be.set_backproject_slice_shape()
        be.set_backproject_tuning()
        be.set_backproject_order()


def test_backproject_cube64_golden(be, oracle, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cube64.npz"))
    g = (64, 64, 0.2, 0.2, 0, 0, 100, 200, 45.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    filtered = []
    oracle.reconstruct(odet, oracle.calculate_volume_geometry(odet), 8, filtered_out=filtered)
    got = hip_backproject_all(be, filtered, det, vg, (64, 64, 64))
    assert_bit_equal(got[31:34], gold["slices"])
    assert got.sum(dtype=np.float64) == float(gold["sum"])


def test_backproject_detector_edges_and_empty_views(be, oracle):
    """Volume much larger than the cone: most rays miss the detector (strict-inside rule); a detector
    offset pushes the valid region to one side. Negative d_so gives whatever the oracle gives."""
    for g, dims, scale in (((40, 30, 0.4, 0.4, 6.0, -4.0, 120, 80, 33.0), (50, 52, 56), 3.0),
                           ((40, 30, 0.4, 0.4, 0.0, 0.0, -120, 80, 33.0), (20, 24, 28), 1.0)):
        det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
        nat = B.calculate_volume_geometry(det)
        dz, dy, dx = dims
        vg = B.VolumeGeometry(dx, dy, dz, nat.l_vx_x * scale, nat.l_vx_x * scale, nat.l_vx_x * scale)
        ovg = oracle.VolumeGeometry(dx, dy, dz, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
        projs = [oracle.lcg_projection(40, 30, i) for i in range(11)]
        want = oracle_backproject_all(oracle, projs, odet, ovg, dims)
        got = hip_backproject_all(be, projs, det, vg, dims)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) or \
            (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]))


@pytest.mark.parametrize("reach_over_d_so", [0.5, 0.89, 0.91, 1.3])
@pytest.mark.parametrize("fused", [False, True])
def test_backproject_shared_reciprocal_division_range(be, oracle, reach_over_d_so, fused):
    """The two per-column divisions by s + d_so share one reciprocal only while every denominator of the launch stays in
    [0.1, 1.9] d_so (bp_device.h column_constants, backproject.hip fill_params); beyond that -- here the grid is scaled until
    its corner passes 0.9 d_so and then the source itself, denominators near and below zero -- the IEEE sequence runs. Same
    bits as the oracle on both sides of the switch, per-projection and fused launches."""
    g = (48, 40, 0.4, 0.4, 1.0, -2.0, 90, 60, 29.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    dims = (24, 40, 44)
    dz, dy, dx = dims
    corner = np.hypot(dx / 2.0, dy / 2.0)
    l_vx = float(np.float32(reach_over_d_so * 90.0 / corner))
    vg = B.VolumeGeometry(dx, dy, dz, l_vx, l_vx, l_vx)
    ovg = oracle.VolumeGeometry(dx, dy, dz, l_vx, l_vx, l_vx)
    projs = [oracle.lcg_projection(48, 40, i) for i in range(12)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    if fused:
        n, rows, cols = 12, 40, 48  # n_row = pixels per detector row
        stack = be.make_projection_device(cols, rows * n)
        be.copy_h2d(B.Projection(np.ascontiguousarray(np.stack(projs).reshape(n * rows, cols)), cols, rows * n), stack)
        sc = [B.stage_angle(det, i) for i in range(n)]
        d_v = be.make_volume_device(dx, dy, dz)
        be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * rows, n, cols, rows, d_v, 0, det, vg, False, None,
                             [s_ for s_, _ in sc], [c_ for _, c_ in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
        got = volume_to_host(be, d_v)
        be.free(d_v)
        be.free(stack)
    else:
        got = hip_backproject_all(be, projs, det, vg, dims)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))


@pytest.mark.parametrize("fused", [False, True])
def test_backproject_tiles_no_ray_reaches(be, oracle, fused):
    """A grid much wider and taller than the cone: for every projection most waves' tiles lie entirely outside the rays (x taps
    off the detector, or rows above / below it over the tile's whole depth). On a volume the library allocated those waves skip
    their tile (paris_hip_set_backproject_skip_invalid, default); with the switch off, and on a volume the host uploaded -- it may
    hold -0, which the reference's + 0 turns into +0 -- every addition is made. All equal the oracle bit for bit, including the
    sign of the zeros."""
    g = (64, 48, 0.4, 0.4, 1.5, -2.0, 200, 150, 23.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    dims = (96, 72, 192)  # z, y, x: about three times the field of view across, twice its height
    dz, dy, dx = dims
    l_vx = float(nat.l_vx_x)
    vg = B.VolumeGeometry(dx, dy, dz, l_vx * 1.1, l_vx * 2.6, l_vx * 1.7)
    ovg = oracle.VolumeGeometry(dx, dy, dz, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    n = 9
    projs = [oracle.lcg_projection(64, 48, i) - np.float32(0.5) for i in range(n)]
    start = np.zeros(dims, np.float32)
    start[::3, ::5, ::7] = np.float32(-0.0)  # the uploaded volume's -0 entries
    want_zero = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    want_up = start.copy()
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want_up, p, 0, odet, ovg, s, c, ds, dt, None)
    assert np.signbit(want_up[want_up == 0]).sum() == 0  # the reference's + 0 leaves no -0 behind

    def run(d_v):
        if fused:
            rows, cols = 48, 64
            stack = be.make_projection_device(cols, rows * n)
            be.copy_h2d(B.Projection(np.ascontiguousarray(np.stack(projs).reshape(n * rows, cols)), cols, rows * n), stack)
            sc = [B.stage_angle(det, i) for i in range(n)]
            be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * rows, n, cols, rows, d_v, 0, det, vg, False, None,
                                 [s_ for s_, _ in sc], [c_ for _, c_ in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
            be.free(stack)
        else:
            for i, p in enumerate(projs):
                d_p = to_device(be, p, idx=i)
                B.backproject(be, d_p, d_v, 0, det, vg, False, False, None)
                be.free(d_p)
        out = volume_to_host(be, d_v)
        be.free(d_v)
        return out

    for skip in (True, False):
        be.set_backproject_skip_invalid(skip)
        try:
            assert_bit_equal(run(be.make_volume_device(dx, dy, dz)), want_zero)
            d_v = be.make_volume_device(dx, dy, dz)
            be.copy_h2d(B.Volume(start.copy(), dx, dy, dz), d_v)
            assert_bit_equal(run(d_v), want_up)
        finally:
            be.set_backproject_skip_invalid(True)


def test_backproject_with_angle_file_values(be, oracle):
    """enable_angles: phi comes from projection::phi instead of idx * delta_phi (src/backprojection.cpp:52-57)."""
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    angles = [3.5, 91.25, 200.0, 359.9]
    projs = [oracle.lcg_projection(64, 48, i) for i in range(4)]
    want = np.zeros((61, 67, 67), np.float32)
    d_v = be.make_volume_device(67, 67, 61)
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i, True, angles[i])
        oracle.backproject(want, p, 0, odet, ovg, s, c, ds, dt)
        d_p = to_device(be, p, idx=i, phi=angles[i])
        B.backproject(be, d_p, d_v, 0, det, vg, True, False, None)
        be.free(d_p)
    assert_bit_equal(volume_to_host(be, d_v), want)
    be.free(d_v)


def test_backproject_batch_equals_sequence(be, oracle, kat_golden):
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    f = kat_golden["filtered"]
    n, rows, cols = f.shape
    stack = be.make_projection_device(cols, rows * n)  # n projections back to back, one pitch
    h = B.Projection(np.ascontiguousarray(f.reshape(n * rows, cols)), cols, rows * n)
    be.copy_h2d(h, stack)
    sc = [B.stage_angle(det, i) for i in range(n)]
    d_v = be.make_volume_device(67, 67, 61)
    be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * rows, n, cols, rows, d_v, 0, det, vg, False, None,
                         [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
    assert_bit_equal(volume_to_host(be, d_v), kat_golden["volume"])
    be.free(d_v)
    be.free(stack)


def test_backproject_f16_projections(be, oracle, kat_golden):
    """BASELINE config 5 storage: projections held as IEEE half, widened when staged, fp32 everywhere else. Equals
    the oracle fed the half-rounded projection, bit for bit; the conversion kernel rounds like numpy (nearest even)."""
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    rounded = [(kat_golden["filtered"][i] * np.float32(3.0)).astype(np.float16).astype(np.float32) for i in range(8)]
    want = oracle_backproject_all(oracle, rounded, odet, ovg, (61, 67, 67))
    for variant in (2, 1):
        be.set_backproject_variant(variant)
        d_v = be.make_volume_device(67, 67, 61)
        for i in range(8):
            p = kat_golden["filtered"][i] * np.float32(3.0)
            d_p = to_device(be, p, idx=i)
            h_ptr, h_pitch = be.convert_projection_f16(d_p)
            back = np.empty((48, h_pitch // 2), np.float16)
            _lib.check(_lib.load().paris_hip_memcpy_volume_d2h(be._ctx, back.ctypes.data, h_ptr, h_pitch // 4, 48, 1), "d2h")
            be.synchronize()
            assert np.array_equal(back[:, :64].view(np.uint16), p.astype(np.float16).view(np.uint16))
            s, c = B.stage_angle(det, i)
            be.backproject_f16(h_ptr, h_pitch, 64, 48, d_v, 0, det, vg, False, None, s, c,
                               det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
            be.free(h_ptr)
            be.free(d_p)
        got = volume_to_host(be, d_v)
        be.free(d_v)
        be.set_backproject_variant(0)
        assert_bit_equal(got, want)


def test_backproject_f16_calls_are_deferred_too(oracle, kat_golden):
    """paris_hip_backproject_f16 behind the deferral: 8 half-precision calls with depth 3 (groups of 3 + 3 + 2 at read-back),
    one fp32 call in between (another precision: the pending half group is flushed first); bit-identical to immediate calls."""
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    frames = [kat_golden["filtered"][i] * np.float32(3.0) for i in range(8)]

    def run(depth):
        with B.Backend(0, synchronous=False) as abe:
            abe.set_backproject_deferral(depth)
            d_v = abe.make_volume_device(67, 67, 61)
            for i, p in enumerate(frames):
                d_p = to_device(abe, p, idx=i)
                if i == 4:  # one fp32 call among the half ones
                    B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
                else:
                    h_ptr, h_pitch = abe.convert_projection_f16(d_p)
                    s, c = B.stage_angle(det, i)
                    abe.backproject_f16(h_ptr, h_pitch, 64, 48, d_v, 0, det, vg, False, None, s, c, det.delta_s * det.l_px_row,
                                        det.delta_t * det.l_px_col)
                    abe.free(h_ptr)  # released while the call is still pending: the ring holds the snapshot
                abe.free(d_p)
            return volume_to_host(abe, d_v)

    rounded = [p if i == 4 else p.astype(np.float16).astype(np.float32) for i, p in enumerate(frames)]
    want = oracle_backproject_all(oracle, rounded, odet, ovg, (61, 67, 67))
    assert_bit_equal(run(1), want)
    assert_bit_equal(run(3), want)


@pytest.mark.parametrize("dims,roi_x2", [((21, 42, 72), 82), ((21, 42, 71), 81)])
def test_backproject_f16_batch_bit_exact(be, oracle, dims, roi_x2):
    """paris_hip_backproject_batch_f16: the fused kernel on IEEE-half projections (one under the other in one buffer) equals
    the oracle fed the half-rounded frames, in order, bit for bit; even and odd volume widths (lane widths 2 and 1)."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 9.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(90, 50, 60, nat.l_vx_x * 1.1, nat.l_vx_x * 1.6, nat.l_vx_x * 1.4)
    ovg = oracle.VolumeGeometry(90, 50, 60, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    roi, oroi = B.RegionOfInterest(10, roi_x2, 5, 47, 6, 51), oracle.RegionOfInterest(10, roi_x2, 5, 47, 6, 51)
    n_proj, v_offset = 11, 12
    frames = [(oracle.lcg_projection(96, 80, i) - np.float32(0.5)) * np.float32(2.5) for i in range(n_proj)]
    want = np.zeros(dims, np.float32)
    for i, p in enumerate(frames):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, p.astype(np.float16).astype(np.float32), v_offset, odet, ovg, s, c, ds, dt, oroi)
    # the half frames, 80 rows each, in one pitched device buffer (the fp32 allocator with half the width)
    stack = be.make_projection_device(48, 80 * n_proj)
    halves = np.zeros((80 * n_proj, stack.pitch // 2), np.float16)
    for i, p in enumerate(frames):
        halves[80 * i:80 * (i + 1), :96] = p.astype(np.float16)
    _lib.check(_lib.load().paris_hip_memcpy_volume_h2d(be._ctx, stack.ptr, halves.ctypes.data, stack.pitch // 4, 80 * n_proj, 1), "h2d")
    sc = [B.stage_angle(det, i) for i in range(n_proj)]
    d_v = be.make_volume_device(dims[2], dims[1], dims[0])
    be.backproject_batch_f16(stack.ptr, stack.pitch, stack.pitch * 80, n_proj, 96, 80, d_v, v_offset, det, vg, True, roi,
                             [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
    assert_bit_equal(volume_to_host(be, d_v), want)
    be.free(d_v)
    be.free(stack)


def test_config5_grid_roi_slab_f16(be, oracle):
    """BASELINE config 5 in miniature on its real grid: 2048^2 detector, 4096^3 voxel grid (half the natural voxel
    size), ROI {1024..3072}^3, one 2048 x 2048 x 8 slab of the ROI (v_offset inside the ROI), fp16 projections.
    Crops computed by the oracle through its own ROI + offset path on the half-rounded projection must match."""
    n = 2048
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 3600)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(4096))
    vg = B.VolumeGeometry(4096, 4096, 4096, l, l, l)
    ovg = oracle.VolumeGeometry(4096, 4096, 4096, l, l, l)
    roi = B.RegionOfInterest(1024, 3072, 1024, 3072, 1024, 3072)
    rg = B.apply_roi(vg, 1024, 3072, 1024, 3072, 1024, 3072)
    assert (rg.dim_x, rg.dim_y, rg.dim_z) == (2048, 2048, 2048)
    v_offset, nz = 1536, 8  # slab 6 of 8 would start at 1536
    idxs = (5, 1234)
    projs = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    d_v = be.make_volume_device(2048, 2048, nz)
    for i, p in zip(idxs, projs):
        d_p = to_device(be, p, idx=i)
        h_ptr, h_pitch = be.convert_projection_f16(d_p)
        s, c = B.stage_angle(det, i)
        be.backproject_f16(h_ptr, h_pitch, n, n, d_v, v_offset, det, vg, True, roi, s, c, 0.0, 0.0)
        be.free(h_ptr)
        be.free(d_p)
    got = volume_to_host(be, d_v)
    be.free(d_v)
    for (x1, y1, z1) in ((0, 0, 0), (1000, 1100, 2), (1984, 1984, 4)):
        oroi = oracle.RegionOfInterest(1024 + x1, 1024 + x1 + 64, 1024 + y1, 1024 + y1 + 64, 1024, 3072)
        want = np.zeros((4, 64, 64), np.float32)
        for i, p in zip(idxs, projs):
            s, c, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want, p.astype(np.float16).astype(np.float32), v_offset + z1, odet, ovg, s, c, ds, dt, oroi)
        assert_bit_equal(got[z1:z1 + 4, y1:y1 + 64, x1:x1 + 64], want)


def test_make_subvolume_information(be):
    """src/cuda/subvolume_information.cpp:63-118: slab count doubles until volume/devices + 10 projections fit."""
    det = B.DetectorGeometry(2048, 2048, 0.2, 0.2, 0, 0, 500, 500, 0.25)
    fits = be.make_subvolume_information(B.VolumeGeometry(2048, 2048, 2048, 0.1, 0.1, 0.1), det, 1)
    assert (fits.num, fits.geo.dim_z, fits.geo.remainder) == (1, 2048, 0)  # 32 GiB fits one 288 GB device
    huge = B.VolumeGeometry(8192, 8192, 8191, 0.1, 0.1, 0.1)               # 2 TiB
    info = be.make_subvolume_information(huge, det, 1)
    assert info.num >= 8 and info.num & (info.num - 1) == 0
    assert info.geo.dim_z * info.num + info.geo.remainder == 8191
    assert (info.geo.dim_x, info.geo.dim_y) == (8192, 8192)
    assert info.geo.dim_z * 8192 * 8192 * 4 < 288e9


@pytest.mark.parametrize("n_proj,tz,lds_bytes,vx,x2", [(9, 0, 0, 0, 82), (40, 0, 0, 0, 82), (9, 8, 0, 0, 82), (5, 0, 1024, 0, 82),
                                                         (33, 8, 4096, 0, 82), (9, 0, 0, 4, 82), (9, 8, 0, 4, 82), (9, 0, 0, 1, 82),
                                                         (9, 0, 0, 0, 81), (7, 0, 1024, 0, 79), (9, 0, 0, 0, 80),
                                                         (9, 0, 0, 2, 82), (9, 8, 0, 2, 82), (33, 16, 2048, 2, 80), (9, 16, 0, 1, 82),
                                                         (9, 32, 0, 1, 81), (5, 16, 1024, 0, 82),
                                                         (64, 0, 0, 0, 82), (65, 8, 0, 2, 82), (70, 0, 0, 1, 81), (49, 16, 2048, 4, 82)])
def test_backproject_fused_batch_bit_exact(be, oracle, n_proj, tz, lds_bytes, vx, x2):
    """paris_hip_backproject_batch's fused kernel (n_proj projections per launch, split at 64) adds the projections
    to every voxel in projection order: bit-identical to the oracle's sequential loop. Partial tiles in x, y, z;
    ROI and slab offset; every tile depth (8, 16, 32); an LDS budget that forces the global tap path; every lane width (default:
    one voxel per lane, 32 slices; 2 and 4 voxels on request where the rows are aligned for them)."""
    g = (96, 80, 0.2, 0.25, -2.5, 1.25, 150, 250, 9.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    full = (60, 50, 90)
    vg = B.VolumeGeometry(full[2], full[1], full[0], nat.l_vx_x * 1.1, nat.l_vx_x * 1.6, nat.l_vx_x * 1.4)
    ovg = oracle.VolumeGeometry(full[2], full[1], full[0], vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    roi = B.RegionOfInterest(10, x2, 5, 47, 6, 51)        # 72 x 42 x 45 for x2 = 82
    oroi = oracle.RegionOfInterest(10, x2, 5, 47, 6, 51)
    dims, v_offset = (21, 42, x2 - 10), 12                 # second part of the ROI's slices
    projs = [oracle.lcg_projection(96, 80, i) - np.float32(0.5) for i in range(n_proj)]
    want = np.zeros(dims, np.float32)
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, p, v_offset, odet, ovg, s, c, ds, dt, oroi)

    stack = be.make_projection_device(96, 80 * n_proj)
    be.copy_h2d(B.Projection(np.ascontiguousarray(np.concatenate(projs)), 96, 80 * n_proj), stack)
    sc = [B.stage_angle(det, i) for i in range(n_proj)]
    d_v = be.make_volume_device(dims[2], dims[1], dims[0])
    be.set_backproject_tuning(vx=vx, tz=tz, lds_bytes=lds_bytes)
    try:
        be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * 80, n_proj, 96, 80, d_v, v_offset, det, vg, True, roi,
                             [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
    finally:
        be.set_backproject_tuning()
    assert_bit_equal(volume_to_host(be, d_v), want)
    be.free(d_v)
    be.free(stack)


def test_upload_stream_orders_compute_behind_transfers(oracle, kat_golden):
    """paris_hip_upload_projection on an asynchronous ctx: 8 frames go up on the upload stream while the
    backprojections of the earlier ones are still queued; each backprojection must see its own frame. Two device
    slots are reused, guarded by host-side fences the way the pipelined driver does it (reconstruct.h)."""
    import torch
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    filtered = kat_golden["filtered"]
    with B.Backend(0, synchronous=False) as abe:
        L, ctx = abe._L, abe._ctx
        pinned = [torch.from_numpy(np.ascontiguousarray(f)).pin_memory() for f in filtered]
        slots = [abe.make_projection_device(64, 48) for _ in range(2)]
        fences = []
        for _ in slots:
            f = C.c_void_p()
            assert L.paris_hip_fence_create(ctx, C.byref(f)) == 0
            fences.append(f)
        d_v = abe.make_volume_device(67, 67, 61)
        for i, t in enumerate(pinned):
            s = i % len(slots)
            assert L.paris_hip_fence_wait(ctx, fences[s]) == 0   # the backprojection that last read this slot is done
            h = B.Projection(t.numpy(), 64, 48, idx=i)
            abe.upload(h, slots[s])
            B.backproject(abe, slots[s], d_v, 0, det, vg, False, False, None)
            assert L.paris_hip_fence_record(ctx, fences[s]) == 0
        abe.synchronize()
        got = volume_to_host(abe, d_v)
        for f in fences:
            assert L.paris_hip_fence_destroy(ctx, f) == 0
    assert_bit_equal(got, kat_golden["volume"])


@pytest.mark.parametrize("band", [False, True])
def test_upload_into_a_busy_slot_waits_for_its_readers(oracle, band):
    """paris_hip_upload_projection runs on the upload stream; re-uploading into a device slot whose
    backprojections are still queued on the compute stream would overwrite pixels they have yet to read. The library orders
    the upload behind the LAST call that touched that buffer (and nothing else). 1024 x 1024 frames into a 1024 x 1024 x 192
    slab, four backprojections (~0.6 ms of queued kernels) read the slot, then a different frame is uploaded into it with NO
    fence or sync in between and backprojected into a second volume: both volumes must equal the runs with a synchronize before
    the re-upload. band=True: the upload and the stage calls address only a band of rows (an interior pointer of the slot)."""
    import torch
    n = 1024
    det = B.DetectorGeometry(n, n, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 1.0)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    z0, dz = 416, 192
    first, count = (B.slab_row_band(det, vg, n, n, dz, z0) if band else (0, n))
    frames = [torch.from_numpy(oracle.lcg_projection(n, n, 40 + i) - np.float32(0.5)).pin_memory() for i in range(2)]

    def run(sync_before_reupload):
        with B.Backend(0, synchronous=False) as abe:
            L = abe._L
            slot = abe.make_projection_device(n, n)
            va, vb = abe.make_volume_device(n, n, dz), abe.make_volume_device(n, n, dz)

            def up(t):
                rows = t.numpy()[first:first + count]
                B._lib.check(L.paris_hip_upload_projection(abe._ctx, slot.ptr + first * slot.pitch, slot.pitch, rows.ctypes.data, n * 4, n,
                                                           count), "upload")
            up(frames[0])
            for i in range(4):
                slot.idx = 7 * i
                B.backproject(abe, slot, va, z0, det, vg, False, False, None)   # four queued readers of the slot
            if sync_before_reupload:
                abe.synchronize()
            up(frames[1])                                                        # same slot, no fence
            slot.idx = 3
            B.backproject(abe, slot, vb, z0, det, vg, False, False, None)
            abe.synchronize()
            return volume_to_host(abe, va), volume_to_host(abe, vb)

    want_a, want_b = run(True)
    got_a, got_b = run(False)
    assert_bit_equal(got_a, want_a)
    assert_bit_equal(got_b, want_b)
    assert np.abs(want_a).max() > 0 and not np.array_equal(want_a, want_b)


@pytest.mark.parametrize("depth,launches", [(2, 4), (5, 2), (16, 1), (64, 1)])
def test_deferred_backprojection_is_bit_identical(oracle, kat_golden, depth, launches):
    """paris_hip_set_backproject_deferral: the per-projection calls of the plugin boundary are snapshotted and added by
    one fused launch per `depth` calls. One device buffer is overwritten for every projection (the snapshot must be taken
    in stream order), nothing synchronises in between; the launch count comes from the timing ring."""
    import torch
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    filtered = kat_golden["filtered"]
    with B.Backend(0, synchronous=False) as abe:
        L, ctx = abe._L, abe._ctx
        pinned = [torch.from_numpy(np.ascontiguousarray(f)).pin_memory() for f in filtered]
        d_p = abe.make_projection_device(64, 48)
        d_v = abe.make_volume_device(67, 67, 61)
        abe.set_backproject_deferral(depth)
        abe.backproject_timing_arm(64)
        for i, t in enumerate(pinned):
            assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, t.data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = i
            B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
        assert len(abe.backproject_timing_collect()) == launches       # collecting flushes what is pending
        assert_bit_equal(volume_to_host(abe, d_v), kat_golden["volume"])

        # a call with other slab arguments flushes the pending ones first; so does reading the volume back
        roi = B.RegionOfInterest(8, 40, 4, 36, 10, 30)
        d_a = abe.make_volume_device(32, 32, 10)
        d_b = abe.make_volume_device(32, 32, 10)
        for i, t in enumerate(pinned):
            assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, t.data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = i
            B.backproject(abe, d_p, d_a, 0, det, vg, False, True, roi)
            if i % 3 == 2:
                B.backproject(abe, d_p, d_b, 10, det, vg, False, True, roi)
        full = kat_golden["volume"]
        assert_bit_equal(volume_to_host(abe, d_a), full[10:20, 4:36, 8:40])
        want_b = np.zeros((10, 32, 32), np.float32)
        odet, ovg = oracle.DetectorGeometry(*KAT), oracle.calculate_volume_geometry(oracle.DetectorGeometry(*KAT))
        for i in (2, 5):
            s, c, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want_b, filtered[i], 10, odet, ovg, s, c, ds, dt, oracle.RegionOfInterest(8, 40, 4, 36, 10, 30))
        assert_bit_equal(volume_to_host(abe, d_b), want_b)

        # pending work and an explicit flush; a bad argument is reported by the call that makes it
        B.backproject(abe, d_p, d_a, 0, det, vg, False, True, roi)
        abe.flush()
        abe.synchronize()
        assert L.paris_hip_backproject(ctx, None, d_p.pitch, 64, 48, d_a.ptr, 32, 32, 10, 0, C.byref(det), C.byref(vg), 0, None,
                                       0.0, 1.0, 0.0, 0.0) == _lib.ERROR_INVALID_ARGUMENT
        B.backproject(abe, d_p, d_a, 0, det, vg, False, True, roi)      # left pending: ctx destruction drops it


@pytest.mark.parametrize("depth,band", [(5, None), (3, (6, 21)), (64, None)])
def test_filter_deferral_is_bit_identical(oracle, depth, band):
    """paris_hip_set_filter_deferral: with stage fusion and backprojection deferral on, the filter() that follows a weight() is held
    back as well; a backproject() of that projection takes both into its ring slot (the UNFILTERED frame is snapshotted) and the
    group's weighting + filter run as one launch before the fused backprojection. One device buffer is refilled for every
    projection. Same volume bit for bit as with the switch off; every other follow-up call runs the held-back launch first, in
    place; after a deferred backprojection the caller's buffer still holds the raw pixels (the documented difference)."""
    g = (512, 40, 0.2, 0.2, 1.25, -0.5, 300, 200, 7.0)   # 512 pixels per row: a 1024-point filter, the fused weight + filter kernel
    det = B.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(96, 80, 24, nat.l_vx_x * 4.0, nat.l_vx_x * 4.5, nat.l_vx_z * 1.2)
    raws = [oracle.lcg_projection(512, 40, 40 + i) for i in range(11)]
    first, count = band if band is not None else (0, 40)

    def run(hold):
        with B.Backend(0, synchronous=False) as abe:
            abe.set_stage_fusion(True)
            abe.set_backproject_deferral(depth)
            abe.set_filter_deferral(hold)
            L, ctx = abe._L, abe._ctx
            d_p = abe.make_projection_device(512, 40)
            d_v = abe.make_volume_device(96, 80, 24)
            seen = {}
            for i, raw in enumerate(raws):
                h = np.ascontiguousarray(raw)
                assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, h.ctypes.data, 512 * 4, 512, 40) == 0
                d_p.idx = i
                if i == 4:                      # the one-call form: nothing is held, the slot is filtered already (mixed group)
                    B.weight_filter_rows(abe, d_p, det, first, count)
                else:
                    B.weight_rows(abe, d_p, det, first, count)
                    B.filter_rows(abe, d_p, det, first, count)
                if i == 6:                      # the projection is read back instead: the held-back launch runs first, in place
                    seen["filtered"] = to_host(abe, d_p).copy()
                if i == 8:                      # a second filter on the same rows: the first one must not be lost
                    B.filter_rows(abe, d_p, det, first, count)
                B.backproject(abe, d_p, d_v, 3, det, vg, False, False, None)
                if i == 2:
                    seen["after"] = to_host(abe, d_p).copy()   # (flushes the group; the buffer: raw with the switch, filtered without)
            B.weight_rows(abe, d_p, det, first, count)
            B.filter_rows(abe, d_p, det, first, count)         # held and never backprojected: dropped with the buffer
            abe.free(d_p)
            vol = volume_to_host(abe, d_v)
            return vol, seen

    want, seen_off = run(False)
    got, seen_on = run(True)
    assert_bit_equal(got, want)
    assert np.abs(want).max() > 0
    assert np.array_equal(seen_on["filtered"].view(np.uint32), seen_off["filtered"].view(np.uint32))
    # the documented difference: the buffer of a projection whose filter went into the ring keeps its raw pixels
    assert np.array_equal(seen_on["after"], raws[2]) and not np.array_equal(seen_off["after"], raws[2])
    # and against the oracle's pipeline (FFT rounding only)
    odet = oracle.DetectorGeometry(*g)
    ovg = oracle.VolumeGeometry(96, 80, 24, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    fs = oracle.filter_size(512)
    k = oracle.make_filter(fs, odet.l_px_row)
    ref = np.zeros((24, 80, 96), np.float32)
    for i, raw in enumerate(raws):
        p = raw.copy()
        oracle.weight(p, odet)
        oracle.apply_filter(p, k, fs)
        if i == 8:
            oracle.apply_filter(p, k, fs)
        if band is not None:   # rows outside the band stay raw on the GPU; the slab below only reads rows inside it
            q = raw.copy()
            q[first:first + count] = p[first:first + count]
            p = q
        sn, cs, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(ref, p, 3, odet, ovg, sn, cs, ds, dt, None)
    if band is None:
        assert np.max(np.abs(got - ref)) <= FILTER_TOL * np.abs(ref).max()


# ---- the whole hot path ------------------------------------------------------------------------------------------

def test_pipeline_against_oracle(be, oracle, kat_golden):
    """weight -> filter -> backproject on the GPU vs the oracle's pipeline. The only difference is FFT rounding in
    the filter, which the (linear) backprojection carries through: same tolerance as the filter."""
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    d_v = be.make_volume_device(67, 67, 61)
    for i in range(8):
        d_p = to_device(be, oracle.lcg_projection(64, 48, i), idx=i)
        B.weight(be, d_p, det)
        B.filter(be, d_p, det)
        B.backproject(be, d_p, d_v, 0, det, vg, False, False, None)
        be.free(d_p)
    got = volume_to_host(be, d_v)
    want = kat_golden["volume"]
    assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
    assert rel_l2(got, want) <= FILTER_TOL
    be.free(d_v)


def test_row_band_pipeline_is_bit_identical_to_the_full_projection(be, oracle):
    """f4: upload + weight + filter only the slab's detector band (paris_hip_slab_row_band, *_rows entry points); the
    band's pixels and the slab volume must equal the full-projection run bit for bit. Rows outside the band hold NaN."""
    g = (200, 180, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 11.25)
    det = B.DetectorGeometry(*g)
    vg = B.calculate_volume_geometry(det)
    n_slabs = 6
    dz = vg.dim_z // n_slabs
    seen_partial = 0
    for variant in (0, 1):  # radix-16 register filter is used from N = 1024 up; N = 512 here takes the radix-2 kernel either way
        be.set_filter_variant(variant)
        for slab in range(n_slabs):
            v_offset = slab * dz
            first, count = B.slab_row_band(det, vg, vg.dim_x, vg.dim_y, dz, v_offset)
            assert count > 0
            seen_partial += count < det.n_col
            d_full = be.make_volume_device(vg.dim_x, vg.dim_y, dz)
            d_band = be.make_volume_device(vg.dim_x, vg.dim_y, dz)
            for i in range(4):
                raw = oracle.lcg_projection(200, 180, i)
                d_p = to_device(be, raw, idx=i)
                B.weight(be, d_p, det)
                B.filter(be, d_p, det)
                B.backproject(be, d_p, d_full, v_offset, det, vg, False, False, None)
                full_pixels = to_host(be, d_p)
                be.free(d_p)
                poisoned = np.full_like(raw, np.nan)
                poisoned[first:first + count] = raw[first:first + count]
                d_q = to_device(be, poisoned, idx=i)
                B.weight_rows(be, d_q, det, first, count)
                B.filter_rows(be, d_q, det, first, count)
                band_pixels = to_host(be, d_q)
                assert np.array_equal(band_pixels[first:first + count].view(np.uint32),
                                      full_pixels[first:first + count].view(np.uint32))
                assert np.isnan(band_pixels[:first]).all() and np.isnan(band_pixels[first + count:]).all()
                B.backproject(be, d_q, d_band, v_offset, det, vg, False, False, None)
                be.free(d_q)
            assert_bit_equal(volume_to_host(be, d_band), volume_to_host(be, d_full))
            be.free(d_full)
            be.free(d_band)
    be.set_filter_variant(0)
    assert seen_partial >= 8


def test_row_band_with_the_register_filter(be, oracle):
    """Same on a 1024-wide detector (N = 2048: the radix-16 register-pass filter), one thin slab off centre."""
    det = B.DetectorGeometry(1024, 256, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 30.0)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(256, 256, nat.dim_z, nat.l_vx_x * 4, nat.l_vx_y * 4, nat.l_vx_z)
    dz, v_offset = 24, nat.dim_z // 2 + 40
    first, count = B.slab_row_band(det, vg, 256, 256, dz, v_offset)
    assert 0 < count < 128
    d_full = be.make_volume_device(256, 256, dz)
    d_band = be.make_volume_device(256, 256, dz)
    for i in range(3):
        raw = oracle.lcg_projection(1024, 256, i)
        d_p = to_device(be, raw, idx=i)
        B.weight(be, d_p, det)
        B.filter(be, d_p, det)
        B.backproject(be, d_p, d_full, v_offset, det, vg, False, False, None)
        full_pixels = to_host(be, d_p)
        be.free(d_p)
        poisoned = np.full_like(raw, np.nan)
        poisoned[first:first + count] = raw[first:first + count]
        d_q = to_device(be, poisoned, idx=i)
        B.weight_rows(be, d_q, det, first, count)
        B.filter_rows(be, d_q, det, first, count)
        assert np.array_equal(to_host(be, d_q)[first:first + count].view(np.uint32),
                              full_pixels[first:first + count].view(np.uint32))
        B.backproject(be, d_q, d_band, v_offset, det, vg, False, False, None)
        be.free(d_q)
    got = volume_to_host(be, d_band)
    assert np.abs(got).max() > 0
    assert_bit_equal(got, volume_to_host(be, d_full))
    be.free(d_full)
    be.free(d_band)


def test_config1_shepp_logan_256_cube(be, oracle):
    """BASELINE config 1: 256^3 volume (voxels twice the natural size), 360 projections @ 512x512 of the analytic 3-D
    Shepp-Logan phantom, the case the reference's OpenMP backend is quoted on. GPU pipeline vs oracle pipeline to the
    filter tolerance; GPU backprojection of the oracle's filtered frames bit for bit."""
    import phantom
    n, n_proj = 512, 360
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / n_proj)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l = float(np.float32(nat.l_vx_x) * np.float32(n) / np.float32(256))
    vg = B.VolumeGeometry(256, 256, 256, l, l, l)
    ovg = oracle.VolumeGeometry(256, 256, 256, l, l, l)
    radius = 0.45 * 256 * l
    fs = oracle.filter_size(n)
    ok = oracle.make_filter(fs, det.l_px_row)
    want = np.zeros((256, 256, 256), np.float32)
    d_pipe = be.make_volume_device(256, 256, 256)
    d_exact = be.make_volume_device(256, 256, 256)
    for i in range(n_proj):
        raw = phantom.projection(n, n, 0.2, 0.2, 500, 500, i * det.delta_phi, radius)
        d_p = to_device(be, raw, idx=i)
        B.weight(be, d_p, det)
        B.filter(be, d_p, det)
        B.backproject(be, d_p, d_pipe, 0, det, vg, False, False, None)
        be.free(d_p)
        of = oracle.apply_filter(oracle.weight(raw.copy(), odet), ok, fs)
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, of, 0, odet, ovg, s, c, ds, dt)
        d_f = to_device(be, of, idx=i)
        B.backproject(be, d_f, d_exact, 0, det, vg, False, False, None)
        be.free(d_f)
    assert_bit_equal(volume_to_host(be, d_exact), want)
    got = volume_to_host(be, d_pipe)
    assert np.max(np.abs(got - want)) <= FILTER_TOL * np.abs(want).max()
    assert rel_l2(got, want) <= FILTER_TOL
    # it is a reconstruction: the skull shell is brighter than the brain, which is brighter than the outside
    centre = want[128, 128, 128]
    assert want[128, 128, 128 + int(0.67 * 0.45 * 256)] > centre > want[128, 128, 4]
    be.free(d_pipe)
    be.free(d_exact)


# ---- properties at bench-like sizes ---------------------------------------------------------------------------------

def test_large_slab_properties(be, oracle):
    """1024 x 1024 detector, 1024 x 1024 x 96 slab of the 1024^3 grid (BASELINE config C2 geometry):
    (a) a crop computed by the oracle (through its ROI path) equals the GPU's voxels bit for bit;
    (b) two z-slabs with offsets equal the single run (slab identity);
    (c) backprojecting 2p gives exactly 2x the voxels of backprojecting p (power-of-two scaling is exact)."""
    n = 1024
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 720)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    assert (vg.dim_x, vg.dim_y) == (n, n)
    z0, nz = 400, 96
    idxs = (0, 77, 400)
    projs = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]

    def run(scale, z_first, z_count):
        d_v = be.make_volume_device(n, n, z_count)
        for i, p in zip(idxs, projs):
            d_p = to_device(be, p * np.float32(scale), idx=i)
            B.backproject(be, d_p, d_v, z_first, det, vg, False, False, None)
            be.free(d_p)
        out = volume_to_host(be, d_v)
        be.free(d_v)
        return out

    full = run(1.0, z0, nz)
    # (a) oracle crops: centre, an x/y edge, a corner
    for (x1, y1, z1) in ((480, 500, 40), (0, 300, 0), (960, 960, 64)):
        roi = oracle.RegionOfInterest(x1, x1 + 64, y1, y1 + 64, z0 + z1, z0 + z1 + 16)
        want = np.zeros((16, 64, 64), np.float32)
        for i, p in zip(idxs, projs):
            s, c, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want, p, 0, odet, ovg, s, c, ds, dt, roi)
        assert_bit_equal(full[z1:z1 + 16, y1:y1 + 64, x1:x1 + 64], want)
    # (b)
    lo = run(1.0, z0, 40)
    hi = run(1.0, z0 + 40, nz - 40)
    assert_bit_equal(np.concatenate([lo, hi]), full)
    # (c)
    assert_bit_equal(run(2.0, z0, nz), full * np.float32(2.0))


def test_full_size_2048_cube_properties(oracle):
    """BASELINE config 3 at its full size -- the whole 2048^3 volume (32 GiB) from 2048^2 projections -- through
    size-independent properties checked on the device (nothing this large goes through the oracle):
    (a) z-slab identity: slab 5 of 8 computed on its own equals slices 1280..1535 of the full run, bit for bit;
    (b) the fused kernel (three projections in one launch) equals the three single launches, bit for bit, everywhere;
    (c) negation: adding p and then -p returns every voxel to exactly +0 (each product and sum is sign-symmetric);
    (d) a 64 x 64 x 8 crop at the far corner equals the oracle's ROI run."""
    import torch
    n = 2048
    free, _ = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GiB of free HBM")
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 0.25)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l_vx = float(np.float32(nat.l_vx_x))
    vg, ovg = B.VolumeGeometry(n, n, n, l_vx, l_vx, l_vx), oracle.VolumeGeometry(n, n, n, l_vx, l_vx, l_vx)
    dev = torch.device("cuda", 0)
    idxs = (3, 500, 1111)
    frames = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    # stream 0 = the legacy default stream: torch's fills / comparisons and the library's kernels are ordered on it (a ctx
    # with a private stream would race the 32 GiB torch.zeros below)
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        stack = torch.from_numpy(np.stack(frames)).to(dev)                       # (3, n, n), pitch n * 4
        projs = [abe.wrap_projection(stack[j].data_ptr(), n * 4, n, n, idx=i, owner=stack) for j, i in enumerate(idxs)]
        full = torch.zeros((n, n, n), device=dev)
        d_full = abe.wrap_volume(full.data_ptr(), n, n, n, owner=full)
        for p in projs:
            B.backproject(abe, p, d_full, 0, det, vg, False, False, None)
        # (a)
        slab = torch.zeros((256, n, n), device=dev)
        d_slab = abe.wrap_volume(slab.data_ptr(), n, n, 256, owner=slab)
        for p in projs:
            B.backproject(abe, p, d_slab, 1280, det, vg, False, False, None)
        abe.synchronize()
        assert torch.equal(slab.view(torch.int32), full[1280:1536].view(torch.int32))
        del slab, d_slab
        # (b)
        fused = torch.zeros((n, n, n), device=dev)
        d_fused = abe.wrap_volume(fused.data_ptr(), n, n, n, owner=fused)
        sc = [B.stage_angle(det, i) for i in idxs]
        abe.backproject_batch(stack.data_ptr(), n * 4, n * n * 4, 3, n, n, d_fused, 0, det, vg, False, None,
                              [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        abe.synchronize()
        assert torch.equal(fused.view(torch.int32), full.view(torch.int32))
        assert int(torch.count_nonzero(full)) > 0.5 * n ** 3
        # (d)
        crop = full[n - 8:, n - 64:, n - 64:].cpu().numpy()
        roi = oracle.RegionOfInterest(n - 64, n, n - 64, n, n - 8, n)
        want = np.zeros((8, 64, 64), np.float32)
        for i, f in zip(idxs, frames):
            s, c, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want, f, 0, odet, ovg, s, c, ds, dt, roi)
        assert_bit_equal(crop, want)
        # (c) on the fused copy: one projection and its negative
        fused.zero_()
        neg = (-stack[1:2]).contiguous()
        p_neg = abe.wrap_projection(neg.data_ptr(), n * 4, n, n, idx=idxs[1], owner=neg)
        B.backproject(abe, projs[1], d_fused, 0, det, vg, False, False, None)
        abe.synchronize()
        assert int(torch.count_nonzero(fused)) > 0.5 * n ** 3
        B.backproject(abe, p_neg, d_fused, 0, det, vg, False, False, None)
        abe.synchronize()
        assert int(torch.count_nonzero(fused.view(torch.int32))) == 0            # every voxel is +0, not -0
        del fused, full


@pytest.mark.parametrize("overlap", [True, False])
def test_deferred_groups_on_the_second_stream(oracle, kat_golden, overlap):
    """paris_hip_set_backproject_overlap: the fused launch of a full ring runs on the ctx's second stream while the caller keeps
    enqueuing the next group's copies on the ctx stream. 42 projections through a ring of 4 -- ten full groups, so both halves
    of the ring are rewritten many times while launches that read them may still be running, and a last partial group -- from
    ONE device buffer that is overwritten for every call, nothing synchronising in between; then an observer (volume copy) and
    more calls. Bit-equal to the oracle either way."""
    import torch
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    filtered = kat_golden["filtered"]
    n = 42
    frames = [np.ascontiguousarray(filtered[i % 8] * np.float32(1.0 + 0.03125 * (i // 8))) for i in range(n)]
    idx = [(5 * i) % 29 for i in range(n)]  # any angles
    want = np.zeros((61, 67, 67), np.float32)
    checkpoints = {}
    for i in range(n):
        s, c, ds, dt = oracle.backproject_constants(odet, idx[i])
        oracle.backproject(want, frames[i], 0, odet, ovg, s, c, ds, dt)
        if i in (16, n - 1):
            checkpoints[i] = want.copy()
    with B.Backend(0, synchronous=False) as abe:
        L, ctx = abe._L, abe._ctx
        abe.set_backproject_overlap(overlap)
        pinned = [torch.from_numpy(f).pin_memory() for f in frames]
        d_p = abe.make_projection_device(64, 48)
        d_v = abe.make_volume_device(67, 67, 61)
        abe.set_backproject_deferral(4)
        for i in range(n):
            assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, pinned[i].data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = idx[i]
            B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
            if i == 16:  # an observer in mid-stream: one projection pending, a launch possibly still running on the other stream
                assert_bit_equal(volume_to_host(abe, d_v), checkpoints[16])
        assert_bit_equal(volume_to_host(abe, d_v), checkpoints[n - 1])
        # freeing the volume with work pending / in flight must be safe (and a new volume may land on the same address)
        for i in range(6):
            assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, pinned[i].data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = idx[i]
            B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
        abe.free(d_v)
        d_w = abe.make_volume_device(67, 67, 61)
        for i in range(5):
            assert L.paris_hip_memcpy_projection_h2d(ctx, d_p.ptr, d_p.pitch, pinned[i].data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = idx[i]
            B.backproject(abe, d_p, d_w, 0, det, vg, False, False, None)
        want5 = np.zeros((61, 67, 67), np.float32)
        for i in range(5):
            s, c, ds, dt = oracle.backproject_constants(odet, idx[i])
            oracle.backproject(want5, frames[i], 0, odet, ovg, s, c, ds, dt)
        assert_bit_equal(volume_to_host(abe, d_w), want5)


@pytest.mark.parametrize("n,rows,band", [(512, 37, (6, 20)), (1024, 16, (0, 16)), (700, 9, (2, 7))])
def test_weight_filter_batch_equals_per_frame_calls(be, oracle, n, rows, band):
    """paris_hip_stage_weight_filter_batch: a group of frames weighted and filtered by ONE launch (grid.y = frame) is bit-identical
    to the per-frame paris_hip_stage_weight_filter_rows calls -- fp32 in place and the half-precision store, a row band with an odd
    first / last pair, frames a non-trivial stride apart; rows outside the band keep their values; against the oracle to the
    filter tolerance."""
    g = (n, rows, 0.2, 0.25, 0.5, -0.25, 400, 300, 1.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    n_frames = 5
    frames = [oracle.lcg_projection(n, rows, 11 + i) for i in range(n_frames)]
    first, count = band
    pad = 3  # rows of slack between frames
    stack_rows = (rows + pad) * n_frames
    host = np.full((stack_rows, n), np.float32(-7.0))
    for i, f in enumerate(frames):
        host[i * (rows + pad):i * (rows + pad) + rows] = f
    a = be.make_projection_device(n, stack_rows)
    b = be.make_projection_device(n, stack_rows)
    for d in (a, b):
        be.copy_h2d(B.Projection(host.copy(), n, stack_rows), d)
    stride = a.pitch * (rows + pad)
    half_pitch = n * 2
    half_stride = half_pitch * (rows + pad)
    import torch
    dev = torch.device("cuda", 0)
    ha = torch.zeros((stack_rows, n), dtype=torch.float16, device=dev)
    hb = torch.zeros((stack_rows, n), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()
    # one launch for the group
    B.weight_filter_batch(be, a.ptr, a.pitch, stride, n_frames, n, rows, det, first, count)
    B.weight_filter_batch(be, b.ptr, b.pitch, stride, n_frames, n, rows, det, first, count, ha.data_ptr(), half_pitch, half_stride)
    got_batch = to_host(be, a)
    raw_after_half = to_host(be, b)
    # frame by frame
    c = be.make_projection_device(n, stack_rows)
    be.copy_h2d(B.Projection(host.copy(), n, stack_rows), c)
    for i in range(n_frames):
        p = be.wrap_projection(c.ptr + i * stride, c.pitch, n, rows)
        B.weight_filter_rows(be, p, det, first, count)
        q = be.wrap_projection(b.ptr + i * stride, b.pitch, n, rows)
        B.weight_filter_rows(be, q, det, first, count, hb.data_ptr() + i * half_stride, half_pitch)
    be.synchronize()
    got_single = to_host(be, c)
    assert np.array_equal(got_batch.view(np.uint32), got_single.view(np.uint32))
    assert np.array_equal(raw_after_half, host)  # the half store leaves the fp32 rows alone
    assert torch.equal(ha.view(torch.int16), hb.view(torch.int16))
    fs = oracle.filter_size(n)
    k = oracle.make_filter(fs, det.l_px_row)
    for i, f in enumerate(frames):
        r0 = i * (rows + pad)
        want = oracle.apply_filter(oracle.weight(f.copy(), odet), k, fs)
        blk = got_batch[r0:r0 + rows]
        assert np.max(np.abs(blk[first:first + count] - want[first:first + count])) <= FILTER_TOL * np.abs(want).max()
        assert np.array_equal(blk[:first], f[:first]) and np.array_equal(blk[first + count:], f[first + count:])
        assert np.all(got_batch[r0 + rows:r0 + rows + pad] == np.float32(-7.0))
    for d in (a, b, c):
        be.free(d)


def device_view(torch, v, dev):
    """a torch tensor over a library-allocated volume (z, y, x): the same memory, no copy"""
    class _Mem:
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Mem(v.ptr, (v.dim_z, v.dim_y, v.dim_x)), device=dev)


@pytest.mark.parametrize("n,n_proj,idxs", [(2048, 1440, (100, 470, 1010)), (1024, 720, (50, 235, 505))])
def test_full_size_skip_on_equals_skip_off(oracle, n, n_proj, idxs):
    """VERDICT r02 (weak 2): the bench's headline path at its full size -- a LIBRARY-allocated n^3 volume (known to hold no -0, so
    waves whose tile no ray reaches leave it untouched), the default tile order and depth -- against the same launches with
    paris_hip_set_backproject_skip_invalid(0), compared on the device bit for bit (int32 views: the sign of every zero included);
    then the fused batch with the skip on against the same. Projections from a fast octant (25 deg) and the two slowest
    (117.5, 252.5 deg) of BENCH_r02's per-octant table. The oracle pins a crop at the far corner (outside the field of view:
    skipped tiles) and one at the centre. BASELINE configs 3 and 2."""
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < (3 * 4 * n ** 3 + (4 << 30)):
        pytest.skip("needs %d GiB of free HBM" % ((3 * 4 * n ** 3 >> 30) + 4))
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / n_proj)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    l_vx = float(np.float32(nat.l_vx_x))
    vg, ovg = B.VolumeGeometry(n, n, n, l_vx, l_vx, l_vx), oracle.VolumeGeometry(n, n, n, l_vx, l_vx, l_vx)
    dev = torch.device("cuda", 0)
    frames = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    with B.Backend(0, stream=torch.cuda.current_stream(dev).cuda_stream, synchronous=False) as abe:
        stack = torch.from_numpy(np.stack(frames)).to(dev)
        projs = [abe.wrap_projection(stack[j].data_ptr(), n * 4, n, n, idx=i, owner=stack) for j, i in enumerate(idxs)]
        v_on = abe.make_volume_device(n, n, n)
        v_off = abe.make_volume_device(n, n, n)
        t_on, t_off = device_view(torch, v_on, dev), device_view(torch, v_off, dev)
        abe.set_backproject_skip_invalid(True)
        for p in projs:
            B.backproject(abe, p, v_on, 0, det, vg, False, False, None)
        abe.set_backproject_skip_invalid(False)
        for p in projs:
            B.backproject(abe, p, v_off, 0, det, vg, False, False, None)
        abe.synchronize()
        assert torch.equal(t_on.view(torch.int32), t_off.view(torch.int32))
        nonzero = int(torch.count_nonzero(t_off))
        assert 0.5 * n ** 3 < nonzero < n ** 3  # the grid's corners lie outside the field of view: there is something to skip
        # the fused batch, skip on, into the re-zeroed (whole fill: clean again) first volume
        abe.set_backproject_skip_invalid(True)
        abe.memset_volume(v_on)
        sc = [B.stage_angle(det, i) for i in idxs]
        abe.backproject_batch(stack.data_ptr(), n * 4, n * n * 4, len(idxs), n, n, v_on, 0, det, vg, False, None,
                              [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        abe.synchronize()
        assert torch.equal(t_on.view(torch.int32), t_off.view(torch.int32))
        for (x1, y1, z1) in ((n - 64, n - 64, n - 8), (n // 2 - 32, n // 2 - 32, n // 2)):
            crop = t_on[z1:z1 + 8, y1:y1 + 64, x1:x1 + 64].cpu().numpy()
            roi = oracle.RegionOfInterest(x1, x1 + 64, y1, y1 + 64, z1, z1 + 8)
            want = np.zeros((8, 64, 64), np.float32)
            for i, f in zip(idxs, frames):
                s, c, ds, dt = oracle.backproject_constants(odet, i)
                oracle.backproject(want, f, 0, odet, ovg, s, c, ds, dt, roi)
            assert_bit_equal(crop, want)
        del t_on, t_off


def test_volume_mark_dirty_and_clean(be, oracle):
    """paris_hip_volume_mark_dirty / _mark_clean: the library cannot see writes that do not go through it.
    A library volume the caller scribbled -0 into through a torch view takes every addition once it is marked dirty (bit-equal
    to the oracle started from the same values: no -0 survives); a whole-volume paris_hip_memset_volume lists it as clean again;
    a torch.zeros volume wrapped with clean=True may skip and equals the oracle; one wrapped without the promise takes every
    addition."""
    import torch
    g = (64, 48, 0.4, 0.4, 1.5, -2.0, 200, 150, 23.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    dims = (96, 72, 192)
    dz, dy, dx = dims
    l_vx = float(nat.l_vx_x)
    vg = B.VolumeGeometry(dx, dy, dz, l_vx * 1.1, l_vx * 2.6, l_vx * 1.7)
    ovg = oracle.VolumeGeometry(dx, dy, dz, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    projs = [oracle.lcg_projection(64, 48, i) - np.float32(0.5) for i in range(5)]
    start = np.zeros(dims, np.float32)
    start[::3, ::5, ::7] = np.float32(-0.0)
    want_zero = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    want_up = start.copy()
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want_up, p, 0, odet, ovg, s, c, ds, dt, None)
    dev = torch.device("cuda", 0)
    minus = torch.from_numpy(start).to(dev)

    def add_all(d_v):
        for i, p in enumerate(projs):
            d_p = to_device(be, p, idx=i)
            B.backproject(be, d_p, d_v, 0, det, vg, False, False, None)
            be.free(d_p)
        return volume_to_host(be, d_v)

    be.set_backproject_skip_invalid(True)
    # (1) foreign write + mark_dirty: every addition is made
    d_v = be.make_volume_device(dx, dy, dz)
    be.synchronize()
    view = device_view(torch, d_v, dev)
    view.copy_(minus)
    torch.cuda.synchronize()
    be.volume_mark_dirty(d_v)
    assert_bit_equal(add_all(d_v), want_up)
    # (2) a whole-volume memset lists it again; the result is the zero-start one either way
    be.memset_volume(d_v)
    assert_bit_equal(add_all(d_v), want_zero)
    # (3) the hazard the call exists for: the same foreign write WITHOUT the call leaves -0 in the tiles nothing reaches
    be.memset_volume(d_v)
    be.synchronize()
    view.copy_(minus)
    torch.cuda.synchronize()
    got = add_all(d_v)
    assert np.signbit(got[got == 0]).sum() > 0, "the skip did not happen (or the tiles were not skipped as a whole)"
    del view
    be.free(d_v)
    # (4) wrapped memory: no promise -> every addition; clean=True -> may skip, same bits as the oracle from zero
    t = minus.clone()
    assert_bit_equal(add_all(be.wrap_volume(t.data_ptr(), dx, dy, dz, owner=t)), want_up)
    z = torch.zeros(dims, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    w = be.wrap_volume(z.data_ptr(), dx, dy, dz, owner=z, clean=True)
    assert_bit_equal(add_all(w), want_zero)
    be.volume_mark_dirty(w)


def test_volume_scan_clean(be, oracle):
    """paris_hip_volume_scan_clean (the device scan for sign-bit zeros): memory of unknown history is read once;
    without a -0 it is listed as clean (tiles no ray reaches are then left alone: shown by the -0 a later unannounced write
    leaves behind), with one it takes every addition and the count is exact -- including words before and after the 16-byte
    aligned body of the range."""
    import torch
    g = (64, 48, 0.4, 0.4, 1.5, -2.0, 200, 150, 23.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    dims = (192, 72, 95)   # odd row length: with the one-word offset below neither end of the range is 16-byte aligned
    dz, dy, dx = dims
    l_vx = float(nat.l_vx_x)
    vg = B.VolumeGeometry(dx, dy, dz, l_vx * 1.1, l_vx * 2.6, l_vx * 1.7)
    ovg = oracle.VolumeGeometry(dx, dy, dz, vg.l_vx_x, vg.l_vx_y, vg.l_vx_z)
    projs = [oracle.lcg_projection(64, 48, i) - np.float32(0.5) for i in range(4)]
    want_zero = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    n = dx * dy * dz
    start = np.zeros(n, np.float32)
    marks = [0, 1, 2, 5, n // 2, n - 3, n - 2, n - 1]
    start[marks] = np.float32(-0.0)
    start[7] = np.float32(-1.5)        # sign bit set but not a zero: not counted
    want_up = start.reshape(dims).copy()
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want_up, p, 0, odet, ovg, s, c, ds, dt, None)
    dev = torch.device("cuda", 0)

    def add_all(d_v):
        for i, p in enumerate(projs):
            d_p = to_device(be, p, idx=i)
            B.backproject(be, d_p, d_v, 0, det, vg, False, False, None)
            be.free(d_p)
        return volume_to_host(be, d_v)

    be.set_backproject_skip_invalid(True)
    pool = torch.zeros(n + 8, dtype=torch.float32, device=dev)
    t = pool[1:n + 1]
    assert t.data_ptr() % 16 == 4
    torch.cuda.synchronize()
    # (1) nothing found: listed as clean, same bits as the oracle from zero ...
    w = be.wrap_volume(t.data_ptr(), dx, dy, dz, owner=pool)
    assert be.volume_scan_clean(w) == 0
    assert_bit_equal(add_all(w), want_zero)
    # ... and the range really is skipped from now on: an unannounced write of -0 survives where no ray reaches
    t.copy_(torch.from_numpy(start).to(dev))
    torch.cuda.synchronize()
    got = add_all(w)
    assert np.signbit(got[got == 0]).sum() > 0
    # (2) the same values scanned: every -0 counted (the -1.5 is none), the promise withdrawn, every addition made
    t.copy_(torch.from_numpy(start).to(dev))
    torch.cuda.synchronize()
    assert be.volume_scan_clean(w) == len(marks)
    assert_bit_equal(add_all(w), want_up)
    # (3) through wrap_volume; a range that is no whole number of floats or no device memory is refused
    t.zero_()
    torch.cuda.synchronize()
    assert_bit_equal(add_all(be.wrap_volume(t.data_ptr(), dx, dy, dz, owner=pool, clean="scan")), want_zero)
    L = be._L
    assert L.paris_hip_volume_scan_clean(be._ctx, t.data_ptr(), 4 * n - 2, None) == 10001  # PARIS_HIP_ERROR_INVALID_ARGUMENT
    host = np.zeros(64, np.float32)
    assert L.paris_hip_volume_scan_clean(be._ctx, host.ctypes.data, host.nbytes, None) == 10001  # PARIS_HIP_ERROR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        be.wrap_volume(t.data_ptr(), dx, dy, dz, owner=pool, clean="maybe")
    be.volume_mark_dirty(w)


def test_ctx_destroy_runs_pending_work_only_into_its_own_volumes(oracle):
    """projections still deferred when the ctx is destroyed are run into a volume the ctx allocated (and has
    not freed), and dropped for any other address -- the library cannot know whether foreign memory still belongs to the
    caller. Raw C ABI, no Python-side flush."""
    import torch
    L = _lib.load()
    det, odet = B.DetectorGeometry(*KAT), oracle.DetectorGeometry(*KAT)
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    dims = (vg.dim_z, vg.dim_y, vg.dim_x)
    projs = [oracle.lcg_projection(64, 48, i) - np.float32(0.5) for i in range(3)]
    want = oracle_backproject_all(oracle, projs, odet, ovg, dims)
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.stack(projs)).to(dev)
    foreign = torch.zeros(dims, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    roi = B.RegionOfInterest()

    def pending_into(ctx, d_v):
        assert L.paris_hip_set_backproject_deferral(ctx, 8) == 0
        for i in range(3):
            s, c = B.stage_angle(det, i)
            assert L.paris_hip_backproject(ctx, frames[i].data_ptr(), 64 * 4, 64, 48, d_v, vg.dim_x, vg.dim_y, vg.dim_z, 0,
                                           C.byref(det), C.byref(vg), 0, C.byref(roi), s, c, det.delta_s * det.l_px_row,
                                           det.delta_t * det.l_px_col) == 0

    ctx = C.c_void_p()
    assert L.paris_hip_ctx_create(0, None, 0, C.byref(ctx)) == 0
    own = C.c_void_p()
    assert L.paris_hip_malloc_volume(ctx, vg.dim_x, vg.dim_y, vg.dim_z, C.byref(own)) == 0
    pending_into(ctx, own)
    assert L.paris_hip_ctx_destroy(ctx) == 0  # the volume outlives the ctx (never freed through it)
    got = device_view(torch, B.Volume(own.value, vg.dim_x, vg.dim_y, vg.dim_z, on_device=True), dev).cpu().numpy()
    assert_bit_equal(got, want)

    ctx = C.c_void_p()
    assert L.paris_hip_ctx_create(0, None, 0, C.byref(ctx)) == 0
    pending_into(ctx, C.c_void_p(foreign.data_ptr()))
    assert L.paris_hip_ctx_destroy(ctx) == 0
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(foreign.view(torch.int32))) == 0  # dropped, nothing written
    # an explicit flush is the caller's way to get them (what Backend.close does for wrapped volumes whose owner is alive)
    ctx = C.c_void_p()
    assert L.paris_hip_ctx_create(0, None, 0, C.byref(ctx)) == 0
    pending_into(ctx, C.c_void_p(foreign.data_ptr()))
    assert L.paris_hip_flush(ctx) == 0
    assert L.paris_hip_free(ctx, own) == 0  # the first ctx's volume: an address this ctx does not know is simply released
    assert L.paris_hip_ctx_destroy(ctx) == 0
    torch.cuda.synchronize()
    assert_bit_equal(foreign.cpu().numpy(), want)


def test_shallow_slab_on_a_wide_plane(be, oracle):
    """2048 x 2048 x 8 -- one z tile under tile order 12, whose chunk of 256 slices used to pad the grid to 32 times
    the tile count. Default path and fused batch against oracle crops."""
    n = 2048
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 0.25)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    ovg = oracle.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    v_offset, nz = 1000, 8
    idxs = (7, 400)
    projs = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    stack = be.make_projection_device(n, n * len(idxs))
    be.copy_h2d(B.Projection(np.ascontiguousarray(np.concatenate(projs)), n, n * len(idxs)), stack)
    sc = [B.stage_angle(det, i) for i in idxs]
    for fused in (False, True):
        d_v = be.make_volume_device(n, n, nz)
        if fused:
            be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * n, len(idxs), n, n, d_v, v_offset, det, vg, False, None,
                                 [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        else:
            for j, i in enumerate(idxs):
                p = be.wrap_projection(stack.ptr + j * stack.pitch * n, stack.pitch, n, n, idx=i)
                B.backproject(be, p, d_v, v_offset, det, vg, False, False, None)
        got = volume_to_host(be, d_v)
        be.free(d_v)
        for (x1, y1) in ((0, 0), (1984, 1000), (960, 1984)):
            roi = oracle.RegionOfInterest(x1, x1 + 64, y1, y1 + 64, 0, n)
            want = np.zeros((nz, 64, 64), np.float32)
            for i, p in zip(idxs, projs):
                s, c, ds, dt = oracle.backproject_constants(odet, i)
                oracle.backproject(want, p, v_offset, odet, ovg, s, c, ds, dt, roi)
            assert_bit_equal(got[:, y1:y1 + 64, x1:x1 + 64], want)
    be.free(stack)


def test_config3_4_geometry_slab_crops(be, oracle):
    """BASELINE configs 3 / 4 geometry (2048^2 detector, 2048^3 grid): part of slab 7 of 8 (v_offset 1792, 24 slices =
    one and a half tiles deep) through the default kernel path (y-band tile order, 4-pixel staging, fast division),
    then the same through the fused batch entry; oracle crops at a corner, an edge and the middle."""
    n = 2048
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / 1440)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    ovg = oracle.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    v_offset, nz = 1792, 24
    idxs = (3, 500, 1111)
    projs = [oracle.lcg_projection(n, n, i) - np.float32(0.5) for i in idxs]
    crops = {}
    for (x1, y1, z1) in ((0, 0, 0), (1984, 700, 8), (1000, 1984, 16)):
        roi = oracle.RegionOfInterest(x1, x1 + 64, y1, y1 + 64, 0, n)
        want = np.zeros((8, 64, 64), np.float32)
        for i, p in zip(idxs, projs):
            s, c, ds, dt = oracle.backproject_constants(odet, i)
            oracle.backproject(want, p, v_offset + z1, odet, ovg, s, c, ds, dt, roi)
        crops[(x1, y1, z1)] = want

    stack = be.make_projection_device(n, n * len(idxs))
    be.copy_h2d(B.Projection(np.ascontiguousarray(np.concatenate(projs)), n, n * len(idxs)), stack)
    sc = [B.stage_angle(det, i) for i in idxs]
    for fused in (False, True):
        d_v = be.make_volume_device(n, n, nz)
        if fused:
            be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * n, len(idxs), n, n, d_v, v_offset, det, vg, False, None,
                                 [s for s, _ in sc], [c for _, c in sc], 0.0, 0.0)
        else:
            for j, i in enumerate(idxs):
                p = be.wrap_projection(stack.ptr + j * stack.pitch * n, stack.pitch, n, n, idx=i)
                B.backproject(be, p, d_v, v_offset, det, vg, False, False, None)
        got = volume_to_host(be, d_v)
        be.free(d_v)
        for (x1, y1, z1), want in crops.items():
            assert_bit_equal(got[z1:z1 + 8, y1:y1 + 64, x1:x1 + 64], want)
    be.free(stack)


BAND_STATS = {"cases": 0, "partial": 0}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARIS_FUZZ_SEEDS", "24"))))
def test_backproject_random_geometries_bit_exact(be, oracle, seed):
    """Seeded random geometries: detector size and pitch, offsets of either sign, source / detector distances from a
    narrow to a very wide cone, anisotropic voxels, a volume larger or smaller than the field of view, random ROI and
    slab offset, random angles. Default kernel path (and the fused one for aligned volumes) vs the oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    big = 4 if os.environ.get("PARIS_FUZZ_BIG") else 1
    n_row = int(rng.integers(24, 200 * big))
    n_col = int(rng.integers(16, 160 * big))
    l_r, l_c = float(rng.choice([0.1, 0.2, 0.127, 0.4, 0.074])), float(rng.choice([0.1, 0.2, 0.25, 0.4]))
    d_so = float(rng.uniform(40, 600))
    d_od = float(rng.uniform(20, 600))
    g = (n_row, n_col, l_r, l_c, float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6)), d_so, d_od, float(rng.uniform(0.5, 40)))
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    hi = 400 if os.environ.get("PARIS_FUZZ_BIG") else 120
    full = [int(rng.integers(20, hi)) for _ in range(3)]               # z, y, x of the full grid
    if seed % 3 == 0:
        full[2] = (full[2] + 3) // 4 * 4                               # 16-byte lanes + fused kernel
    scale = [float(nat.l_vx_x * rng.uniform(0.4, 2.5)) for _ in range(3)]
    # keep the grid inside the source orbit (the reference divides by s + d_so)
    half_diag = 0.5 * np.hypot(full[2] * scale[0], full[1] * scale[1])
    if half_diag > 0.8 * d_so:
        f = 0.8 * d_so / half_diag
        scale[0] *= f
        scale[1] *= f
    vg = B.VolumeGeometry(full[2], full[1], full[0], *scale)
    ovg = oracle.VolumeGeometry(full[2], full[1], full[0], *scale)
    use_roi = bool(seed % 2)
    if use_roi:
        x1, y1, z1 = (int(rng.integers(1, d // 3)) for d in (full[2], full[1], full[0]))
        x2, y2, z2 = (int(rng.integers(2 * d // 3, d)) for d in (full[2], full[1], full[0]))
        if seed % 3 == 0:
            x2 = x1 + (x2 - x1) // 4 * 4
        roi, oroi = B.RegionOfInterest(x1, x2, y1, y2, z1, z2), oracle.RegionOfInterest(x1, x2, y1, y2, z1, z2)
        out = (z2 - z1, y2 - y1, x2 - x1)
    else:
        roi = oroi = None
        out = tuple(full)
    v_offset = int(rng.integers(0, max(1, out[0] // 2)))
    dims = (out[0] - v_offset, out[1], out[2])
    n_proj = 5
    angles = [float(rng.uniform(0, 360)) for _ in range(n_proj)]
    projs = [oracle.lcg_projection(n_row, n_col, 100 * seed + i) - np.float32(0.5) for i in range(n_proj)]
    want = np.zeros(dims, np.float32)
    for i, p in enumerate(projs):
        s, c, ds, dt = oracle.backproject_constants(odet, i, True, angles[i])
        oracle.backproject(want, p, v_offset, odet, ovg, s, c, ds, dt, oroi)

    # every other seed forces one of the workgroup -> tile orders and a tile depth (the defaults pick by volume shape, and these
    # volumes are small: without this the band orders 8 / 9 / 12 and shallow tiles would only be fuzzed at full size)
    forced = seed % 2 == 1
    if forced:
        # (the product build has orders 5 and 14 .. 18; the experiments build adds the ones that lost: the draw is made either way, so
        # that both builds see the same geometries)
        drawn = int(rng.choice([0, 1, 5, 8, 9, 12, 14, 15, 16, 17, 18]))
        if not _lib.has_experiments() and drawn in (0, 1, 8, 9, 12):
            drawn = {0: 5, 1: 14, 8: 15, 9: 16, 12: 17}[drawn]
        be.set_backproject_order(drawn, -1)
        be.set_backproject_tuning(tz=int(rng.choice([2, 5, 8, 16])))
    try:
        d_v = be.make_volume_device(dims[2], dims[1], dims[0])
        for i, p in enumerate(projs):
            d_p = to_device(be, p, idx=i, phi=angles[i])
            B.backproject(be, d_p, d_v, v_offset, det, vg, True, use_roi, roi)
            be.free(d_p)
        assert_bit_equal(volume_to_host(be, d_v), want)
        be.free(d_v)
    finally:
        if forced:
            be.set_backproject_order()
            be.set_backproject_tuning()

    # f4: rows outside paris_hip_slab_row_band never reach the slab -- poison them with NaN. A thin sub-slab, so that
    # the band is a real subset of the detector in most cases.
    dz2 = max(1, dims[0] // int(rng.integers(3, 9)))
    z2 = int(rng.integers(0, dims[0] - dz2 + 1))
    first, count = B.slab_row_band(det, vg, dims[2], dims[1], dz2, v_offset + z2, roi)
    BAND_STATS["cases"] += 1
    if count < n_col:
        BAND_STATS["partial"] += 1
        d_v = be.make_volume_device(dims[2], dims[1], dz2)
        for i, p in enumerate(projs):
            q = np.full_like(p, np.nan)
            q[first:first + count] = p[first:first + count]
            d_p = to_device(be, q, idx=i, phi=angles[i])
            B.backproject(be, d_p, d_v, v_offset + z2, det, vg, True, use_roi, roi)
            be.free(d_p)
        assert_bit_equal(volume_to_host(be, d_v), want[z2:z2 + dz2])
        be.free(d_v)

    if True:  # the fused batch entry on the same case (lane width 2 or 1 by the volume's alignment)
        stack = be.make_projection_device(n_row, n_col * n_proj)
        be.copy_h2d(B.Projection(np.ascontiguousarray(np.concatenate(projs)), n_row, n_col * n_proj), stack)
        sc = [B.stage_angle(det, i, True, angles[i]) for i in range(n_proj)]
        d_v = be.make_volume_device(dims[2], dims[1], dims[0])
        be.backproject_batch(stack.ptr, stack.pitch, stack.pitch * n_col, n_proj, n_row, n_col, d_v, v_offset, det, vg, use_roi,
                             roi, [s for s, _ in sc], [c for _, c in sc], det.delta_s * det.l_px_row, det.delta_t * det.l_px_col)
        assert_bit_equal(volume_to_host(be, d_v), want)
        be.free(d_v)
        be.free(stack)


def test_random_geometries_exercised_partial_row_bands():
    """Runs after the fuzz above: the NaN-poison pass must have met real sub-bands, not only whole detectors."""
    assert BAND_STATS["cases"] > 0 and BAND_STATS["partial"] >= BAND_STATS["cases"] // 3, BAND_STATS


def test_empty_and_degenerate_arguments(be, oracle, kat_golden):
    """Empty volumes are a no-op, empty or inconsistent projections are rejected, nothing is touched."""
    import ctypes as C
    L = _lib.load()
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    d_p = to_device(be, kat_golden["filtered"][0])
    d_v = be.make_volume_device(8, 8, 8)
    args = lambda pdx, pdy, vdx, vdy, vdz, pitch: L.paris_hip_backproject(  # noqa: E731
        be._ctx, d_p.ptr, pitch, pdx, pdy, d_v.ptr, vdx, vdy, vdz, 0, C.byref(det), C.byref(vg), 0, None, 0.0, 1.0, 0.0, 0.0)
    assert args(64, 48, 0, 8, 8, d_p.pitch) == _lib.SUCCESS      # empty volume: nothing to do
    assert args(64, 48, 8, 8, 0, d_p.pitch) == _lib.SUCCESS
    assert args(0, 48, 8, 8, 8, d_p.pitch) == _lib.ERROR_INVALID_ARGUMENT   # empty projection
    assert args(64, 48, 8, 8, 8, 100) == _lib.ERROR_INVALID_ARGUMENT        # pitch shorter than a row
    assert args(64, 48, 8, 8, 8, 258) == _lib.ERROR_INVALID_ARGUMENT        # pitch not a multiple of the pixel size
    assert L.paris_hip_backproject(be._ctx, d_p.ptr, d_p.pitch, 64, 48, d_v.ptr, 8, 8, 8, 0, C.byref(det), C.byref(vg), 1, None,
                                   0.0, 1.0, 0.0, 0.0) == _lib.ERROR_INVALID_ARGUMENT  # ROI enabled without a ROI
    assert np.count_nonzero(volume_to_host(be, d_v)) == 0
    assert L.paris_hip_weight(be._ctx, d_p.ptr, d_p.pitch, 0, 48, 0, 0, 1, 1, 1) == _lib.SUCCESS
    assert L.paris_hip_backproject_batch(be._ctx, d_p.ptr, d_p.pitch, d_p.pitch * 48, 0, 64, 48, d_v.ptr, 8, 8, 8, 0, C.byref(det),
                                         C.byref(vg), 0, None, (C.c_float * 1)(), (C.c_float * 1)(), 0.0, 0.0) == _lib.SUCCESS
    be.free(d_p)
    be.free(d_v)


def test_beyond_2_to_32_voxels(be, oracle):
    """64-bit voxel indexing (the reference's 32-bit arithmetic breaks here): a 2048 x 2048 x 1040 slab holds
    2^32 + 67 M voxels; its last 16 slices (entirely beyond element 2^32) must equal a small slab run with the matching
    offset, and an oracle crop."""
    n = 2048
    g = (n, n, 0.2, 0.2, 0, 0, 500, 500, 0.25)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    ovg = oracle.VolumeGeometry(n, n, n, nat.l_vx_x, nat.l_vx_x, nat.l_vx_x)
    nz, z0 = 1040, 500
    p = oracle.lcg_projection(n, n, 9) - np.float32(0.5)
    d_p = to_device(be, p, idx=77)
    big = be.make_volume_device(n, n, nz)
    small = be.make_volume_device(n, n, 16)
    B.backproject(be, d_p, big, z0, det, vg, False, False, None)
    B.backproject(be, d_p, small, z0 + nz - 16, det, vg, False, False, None)
    tail = be.wrap_volume(big.ptr + (nz - 16) * n * n * 4, n, n, 16)  # byte offset 4 * (2^32 + ...): past 16 GiB
    a, b = volume_to_host(be, tail), volume_to_host(be, small)
    assert_bit_equal(a, b)
    roi = oracle.RegionOfInterest(900, 964, 1000, 1064, 0, n)
    want = np.zeros((4, 64, 64), np.float32)
    s, c, ds, dt = oracle.backproject_constants(odet, 77)
    oracle.backproject(want, p, z0 + nz - 16 + 5, odet, ovg, s, c, ds, dt, roi)
    assert_bit_equal(a[5:9, 1000:1064, 900:964], want)
    for v in (big, small, d_p):
        be.free(v)


def test_torch_memory_interop(be, oracle, kat_golden):
    """PyTorch as plumbing: volume and projections owned by torch tensors, kernels enqueued on torch's stream."""
    torch = pytest.importorskip("torch")
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream   # 0 (the legacy default stream) unless the caller set one
    tb = B.Backend(0, stream=stream, synchronous=False)
    assert tb.stream == stream                             # the ctx enqueues where torch does: both stay ordered
    vol = torch.zeros((61, 67, 67), dtype=torch.float32, device=dev)
    v = tb.wrap_volume(vol.data_ptr(), 67, 67, 61, owner=vol)
    for i in range(8):
        t = torch.from_numpy(kat_golden["filtered"][i]).to(dev)
        p = tb.wrap_projection(t.data_ptr(), t.stride(0) * 4, 64, 48, idx=i, owner=t)
        B.backproject(tb, p, v, 0, det, vg, False, False, None)
    torch.cuda.synchronize()
    assert_bit_equal(vol.cpu().numpy(), kat_golden["volume"])
    tb.close()
    # the same on an explicit torch stream: tensor fills, copies and the kernels are ordered on it without any host sync
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        tb = B.Backend(0, stream=side.cuda_stream, synchronous=False)
        assert tb.stream == side.cuda_stream != 0
        big = torch.full((61, 67, 67), 7.0, dtype=torch.float32, device=dev)
        big.zero_()                                       # still running when the first kernel is enqueued
        v = tb.wrap_volume(big.data_ptr(), 67, 67, 61, owner=big)
        frames = torch.from_numpy(kat_golden["filtered"]).pin_memory().to(dev, non_blocking=True)
        for i in range(8):
            p = tb.wrap_projection(frames[i].data_ptr(), 64 * 4, 64, 48, idx=i, owner=frames)
            B.backproject(tb, p, v, 0, det, vg, False, False, None)
        out = big.cpu()
    assert_bit_equal(out.numpy(), kat_golden["volume"])
    tb.close()


# ---- round 4: the per-projection loop beside the fused launches ------------------------------------------------------

@pytest.mark.parametrize("how", ["free", "rebuild"])
def test_freeing_the_filter_with_a_deferred_group_pending(oracle, how):
    """with filter deferral the ring slots of the pending group keep pointers to K's permuted copy until the group
    launches. Deferral 8 + filter deferral, three projections pending, then K goes away -- paris_hip_free of the K the caller
    made, or the stage wrapper rebuilding its cached K because the window changed -- and only then is the volume read: the
    group's filter must have run with the K it was asked for. Equal to the run without filter deferral bit for bit."""
    g = (512, 24, 0.2, 0.2, 0.75, -0.5, 300, 200, 11.0)
    det = B.DetectorGeometry(*g)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(80, 72, 20, nat.l_vx_x * 5.0, nat.l_vx_x * 5.0, nat.l_vx_z * 1.1)
    raws = [oracle.lcg_projection(512, 24, 70 + i) for i in range(3)]
    fs = B.filter_size(512)

    def run(hold):
        with B.Backend(0, synchronous=False) as abe:
            abe.set_stage_fusion(True)
            abe.set_backproject_deferral(8)
            abe.set_filter_deferral(hold)
            abe.set_backproject_overlap(True)
            d_v = abe.make_volume_device(80, 72, 20)
            k = abe.make_filter(fs, det.l_px_row) if how == "free" else None
            for i, raw in enumerate(raws):
                d_p = to_device(abe, raw, idx=i)
                B.weight(abe, d_p, det)
                if how == "free":
                    abe.apply_filter(d_p, k, fs, det.n_col)
                else:
                    B.filter(abe, d_p, det)
                B.backproject(abe, d_p, d_v, 2, det, vg, False, False, None)
                abe.free(d_p)
            if how == "free":
                abe.free(k)                                   # the pending slots still refer to it
                junk = [abe.make_filter(fs, 0.37) for _ in range(4)]   # ... and its memory may be handed out again at once
            else:
                abe.set_filter_window(1)                      # the next stage filter rebuilds (frees) the cached K
                d_q = to_device(abe, raws[0], idx=0)
                B.weight(abe, d_q, det)
                B.filter(abe, d_q, det)
                abe.free(d_q)
            return volume_to_host(abe, d_v)

    want = run(False)
    got = run(True)
    assert np.abs(want).max() > 0
    assert_bit_equal(got, want)


def test_close_flushes_only_into_a_wrapped_volume_whose_owner_lives(oracle, kat_golden):
    """Backend.close() decides per pending volume. Projections deferred into a wrapped tensor that is still alive are
    run at close, even though ANOTHER volume was wrapped without an owner earlier; projections deferred into an owner-less wrap
    are not run."""
    import torch
    det = B.DetectorGeometry(*KAT)
    vg = B.calculate_volume_geometry(det)
    dev = torch.device("cuda", 0)
    filtered = kat_golden["filtered"]

    def run(owned):
        vol = torch.zeros((61, 67, 67), dtype=torch.float32, device=dev)
        other = torch.zeros((4, 8, 8), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        abe = B.Backend(0, synchronous=False)
        abe.wrap_volume(other.data_ptr(), 8, 8, 4)                       # no owner: must not decide for `vol`
        v = abe.wrap_volume(vol.data_ptr(), 67, 67, 61, owner=vol if owned else None)
        abe.set_backproject_deferral(16)
        pinned = [torch.from_numpy(np.ascontiguousarray(f)).pin_memory() for f in filtered]
        d_p = abe.make_projection_device(64, 48)
        for i in range(5):   # (Backend.copy_h2d waits and would flush: the raw asynchronous copy keeps the group pending; a new
            #                   sequence launches its first group after 8 calls: 5 stay pending)
            assert abe._L.paris_hip_memcpy_projection_h2d(abe._ctx, d_p.ptr, d_p.pitch, pinned[i].data_ptr(), 64 * 4, 64, 48) == 0
            d_p.idx = i
            B.backproject(abe, d_p, v, 0, det, vg, False, False, None)
        n, ptr = C.c_uint32(0), C.c_void_p()
        assert abe._L.paris_hip_pending_backprojections(abe._ctx, C.byref(n), C.byref(ptr)) == 0
        assert n.value == 5 and ptr.value == vol.data_ptr()
        abe.close()
        torch.cuda.synchronize()
        return vol.cpu().numpy()

    odet = oracle.DetectorGeometry(*KAT)
    ovg = oracle.calculate_volume_geometry(odet)
    want = np.zeros((61, 67, 67), np.float32)
    for i in range(5):
        sn, cs, ds, dt = oracle.backproject_constants(odet, i)
        oracle.backproject(want, filtered[i], 0, odet, ovg, sn, cs, ds, dt)
    assert_bit_equal(run(True), want)
    assert not run(False).any()


@pytest.mark.parametrize("overlap", [True, False])
def test_per_projection_buffers_through_the_pools_beside_the_fused_launch(oracle, overlap):
    """PARIS's loop (src/main.cpp:98-105, src/loader.cpp:28-33) over the C ABI: a pinned host buffer and a device buffer are
    allocated, filled, uploaded (upload stream), weighted, filtered, backprojected (deferral 4) and FREED per projection, nothing
    synchronising in between; with overlap the fused launches run on the second stream. Released buffers come back from the
    pools only when their last user has finished: a pinned buffer after its own H2D copy, a device buffer after its snapshot.
    The host overwrites every pinned buffer it gets at once -- if one came back early the frame in flight would be corrupted.
    1024 x 64 frames into a 1024 x 1024 x 24 slab (launches long enough to still run when the next buffers are taken), 70
    projections (the rotations hold at most 16 pinned and 56 device buffers of this size). Equal bit for bit to the run that synchronises after every call."""
    n, rows, n_proj = 1024, 64, 70   # (more projections than either rotation can hold: buffers MUST come back from the pools)
    det = B.DetectorGeometry(n, rows, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 360.0 / n_proj)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(n, n, 24, nat.l_vx_x, nat.l_vx_x, nat.l_vx_z)
    frames = [oracle.lcg_projection(n, rows, 300 + i) - np.float32(0.5) for i in range(n_proj)]

    def run(synchronise):
        with B.Backend(0, synchronous=False) as abe:
            L, ctx = abe._L, abe._ctx
            abe.set_stage_fusion(True)
            abe.set_backproject_deferral(4)
            abe.set_backproject_overlap(overlap)
            d_v = abe.make_volume_device(n, n, 24)
            seen_host, seen_dev = set(), set()
            for i, f in enumerate(frames):
                h = C.c_void_p()
                assert L.paris_hip_malloc_host(ctx, n * rows * 4, C.byref(h)) == 0
                seen_host.add(h.value)
                C.memmove(h.value, f.ctypes.data, n * rows * 4)          # the host fills the buffer it was just given
                d_p = abe.make_projection_device(n, rows)
                seen_dev.add(d_p.ptr)
                B._lib.check(L.paris_hip_upload_projection(ctx, d_p.ptr, d_p.pitch, h.value, n * 4, n, rows), "upload")
                d_p.idx = i
                B.weight(abe, d_p, det)
                B.filter(abe, d_p, det)
                B.backproject(abe, d_p, d_v, 0, det, vg, False, False, None)
                abe.free(d_p)
                assert L.paris_hip_free_host(ctx, h) == 0
                if synchronise:
                    abe.synchronize()
            vol = volume_to_host(abe, d_v)
            return vol, len(seen_host), len(seen_dev)

    want, _, _ = run(True)
    got, n_host, n_dev = run(False)
    assert np.abs(want).max() > 0
    assert_bit_equal(got, want)
    assert n_host < n_proj and n_dev < n_proj                           # buffers did rotate through the pools


def test_lean_ieee_sequences_are_validated_on_the_device(be, oracle):
    """the shared-reciprocal form of the two per-column divisions (bp_device.h) and the short sqrt / divide of
    the fused weighting (filter_fused.hip) are used only after validate.hip has compared them with the compiler's correctly
    rounded `/` and sqrtf for EVERY fp32 operand of the launch's range. The checks pass on this toolchain for ordinary
    geometries (so the kernels do run the short forms), cost well under a millisecond of device time once cached or not, refuse
    ranges they cannot vouch for, and the volume / weighted frame is the oracle's bit for bit with the validation on and with the
    rounds 2-3 behaviour (range test alone)."""
    import time
    L, ctx = be._L, be._ctx
    exact = C.c_int(-1)
    for d_sd, d_so in ((1000.0, 500.0), (300.0, 100.0), (1.0e6, 7.5e5), (3.0e-3, 1.0e-3), (12345.678, 2345.6789)):
        t0 = time.perf_counter()
        assert L.paris_hip_lean_division_is_exact(ctx, d_sd, d_so, C.byref(exact)) == 0
        first = time.perf_counter() - t0
        assert exact.value == 1, (d_sd, d_so)
        t0 = time.perf_counter()
        assert L.paris_hip_lean_division_is_exact(ctx, d_sd, d_so, C.byref(exact)) == 0
        assert exact.value == 1 and time.perf_counter() - t0 < 1e-3       # cached
        assert first < 0.25                                                # ~3.7e7 denominators; stream + scratch set-up included
    # operands the check refuses: a negative or vanishing source distance, a range that reaches the denormals
    for d_sd, d_so in ((1000.0, -500.0), (1000.0, 0.0), (1.0e-37, 1.0e-38)):
        assert L.paris_hip_lean_division_is_exact(ctx, d_sd, d_so, C.byref(exact)) == 0
        assert exact.value == 0
    for d_sd, q_lo, q_hi in ((1000.0, 1.0e6, 1.09e6), (300.0, 9.0e4, 9.3e4), (0.02, 4.0e-4, 5.0e-4), (5.0e7, 2.5e15, 2.6e15)):
        assert L.paris_hip_lean_weighting_is_exact(ctx, d_sd, q_lo, q_hi, C.byref(exact)) == 0
        assert exact.value == 1, (d_sd, q_lo, q_hi)
    assert L.paris_hip_lean_weighting_is_exact(ctx, 1.0, 1.0e-30, 1.0e30, C.byref(exact)) == 0
    assert exact.value == 0                                                # 200 binades: not validated, the IEEE forms run
    assert L.paris_hip_lean_weighting_is_exact(ctx, 1.0, 2.0, 1.0, C.byref(exact)) == 0 and exact.value == 0

    # same bits with the device validation deciding and with the range test alone: weighting + filter of a 512-wide frame ...
    g = (512, 20, 0.2, 0.25, 1.5, -0.75, 300, 200, 9.0)
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    raw = oracle.lcg_projection(512, 20, 5)
    outs = []
    for validate in (1, 0):
        assert L.paris_hip_set_lean_validation(ctx, validate) == 0
        d_a, d_b = to_device(be, raw), to_device(be, raw)
        B.weight_filter_rows(be, d_a, det, 0, 20)                          # one launch: the short sqrt / divide in its load
        be.set_stage_fusion(False)
        B.weight(be, d_b, det)                                             # the weighting kernel: plain IEEE operations
        B.filter(be, d_b, det)
        a, b = to_host(be, d_a), to_host(be, d_b)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        outs.append(a)
        be.free(d_a)
        be.free(d_b)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    # ... and a backprojection against the oracle
    vg, ovg = B.calculate_volume_geometry(det), oracle.calculate_volume_geometry(odet)
    frame = oracle.lcg_projection(512, 20, 6) - np.float32(0.5)
    want = np.zeros((vg.dim_z, vg.dim_y, vg.dim_x), np.float32)
    sn, cs, ds, dt = oracle.backproject_constants(odet, 3)
    oracle.backproject(want, frame, 0, odet, ovg, sn, cs, ds, dt)
    for validate in (1, 0):
        assert L.paris_hip_set_lean_validation(ctx, validate) == 0
        d_v = be.make_volume_device(vg.dim_x, vg.dim_y, vg.dim_z)
        d_p = to_device(be, frame, idx=3)
        B.backproject(be, d_p, d_v, 0, det, vg, False, False, None)
        assert_bit_equal(volume_to_host(be, d_v), want)
        be.free(d_p)
        be.free(d_v)
    assert L.paris_hip_set_lean_validation(ctx, 1) == 0
