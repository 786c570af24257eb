"""Pins the CPU oracle (oracle/) to the reference: known-answer values of the reference's OpenMP path recorded in
SURVEY.md section 8c (tests/golden/survey_kat.json), the structural identities the reference satisfies (slab ==
slices, ROI == crop, thread-count independence), and the committed golden fixtures."""
import json
import os

import numpy as np
import pytest

import mkl_fftw


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "survey_kat.json")) as f:
        return json.load(f)


def survey_fnv(a, kat):
    """FNV-1a-64 with the survey driver's (non-standard) offset basis."""
    h = kat["fnv_basis"]
    prime = kat["fnv_prime"]
    mask = (1 << 64) - 1
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * prime) & mask
    return "%016x" % h


def kat_det(O, kat):
    g = kat["geometry"]
    return O.DetectorGeometry(g["n_row"], g["n_col"], g["l_px_row"], g["l_px_col"], g["delta_s"], g["delta_t"],
                              g["d_so"], g["d_od"], g["delta_phi"])


def close(a, b, rel=2e-6):
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)


def test_volume_geometry(oracle, kat):
    vg = oracle.calculate_volume_geometry(kat_det(oracle, kat))
    assert (vg.dim_x, vg.dim_y, vg.dim_z) == (67, 67, 61)
    assert np.float32(vg.l_vx_x) == np.float32(kat["vol_geo"]["l_vx"])
    assert vg.l_vx_y == vg.l_vx_x == vg.l_vx_z


def test_apply_roi_rule(oracle, kat):
    vg = oracle.calculate_volume_geometry(kat_det(oracle, kat))
    r = oracle.apply_roi(vg, oracle.RegionOfInterest(8, 40, 4, 36, 10, 30))
    assert (r.dim_x, r.dim_y, r.dim_z) == (32, 32, 20)
    r = oracle.apply_roi(vg, oracle.RegionOfInterest(0, 40, 0, 36, 0, 30))  # +1 when x1 == 0
    assert (r.dim_x, r.dim_y, r.dim_z) == (41, 37, 31)
    r = oracle.apply_roi(vg, oracle.RegionOfInterest(40, 8, 4, 36, 10, 30))  # invalid: unchanged
    assert (r.dim_x, r.dim_y, r.dim_z) == (67, 67, 61)
    r = oracle.apply_roi(vg, oracle.RegionOfInterest(0, 67, 4, 36, 10, 30))  # 68 > 67: unchanged
    assert (r.dim_x, r.dim_y, r.dim_z) == (67, 67, 61)


def test_weight_bit_exact(oracle, kat):
    det = kat_det(oracle, kat)
    p = oracle.lcg_projection(det.n_row, det.n_col, 0)
    oracle.weight(p, det)
    w = kat["weighted_p0"]
    assert p.flat[0] == np.float32(w["0"]) and p.flat[63] == np.float32(w["63"]) and p.flat[-1] == np.float32(w["last"])
    assert survey_fnv(p, kat) == w["fnv"]  # every bit of all 3072 pixels


def test_filter_size(oracle):
    assert [oracle.filter_size(n) for n in (64, 512, 1000, 1024, 2048)] == [128, 1024, 2048, 2048, 4096]


def test_make_filter_values(oracle, kat):
    k = oracle.make_filter(128, 0.2)
    for idx, val in kat["make_filter_128_0.2"].items():
        assert close(float(k[int(idx)]), val, 2e-6), (idx, k[int(idx)], val)
    # analytic cross-check in float64: K = tau * |rFFT(r)|
    r = oracle.make_filter_real(128, 0.2).astype(np.float64)
    ref = 0.2 * np.abs(np.fft.rfft(r))
    assert np.max(np.abs(k - ref)) <= 2e-6 * ref.max()


def test_filtered_projection_values(oracle, kat):
    det = kat_det(oracle, kat)
    p = oracle.lcg_projection(det.n_row, det.n_col, 0)
    oracle.weight(p, det)
    fs = oracle.filter_size(det.n_row)
    oracle.apply_filter(p, oracle.make_filter(fs, det.l_px_row), fs)
    f = kat["filtered_p0_mkl"]
    scale = np.abs(p).max()
    for idx, val in (("0", f["0"]), ("63", f["63"]), ("1000", f["1000"])):
        assert abs(p.flat[int(idx)] - val) <= 1e-5 * scale
    assert abs(p.flat[-1] - f["last"]) <= 1e-5 * scale


def test_full_volume_values(oracle, kat):
    det = kat_det(oracle, kat)
    vg = oracle.calculate_volume_geometry(det)
    vol = oracle.reconstruct(det, vg, 8)
    f = kat["full"]
    amax = np.abs(vol).max()
    assert close(vol.sum(dtype=np.float64), f["sum"], 1e-5)
    assert close(np.abs(vol).sum(dtype=np.float64), f["abssum"], 1e-5)
    assert vol[0, 0, 0] == 0.0
    for (x, y, z), key in (((33, 33, 30), "v_33_33_30"), ((5, 7, 3), "v_5_7_3"), ((8, 4, 10), "v_8_4_10"),
                           ((13, 11, 13), "v_13_11_13")):
        assert abs(vol[z, y, x] - f[key]) <= 1e-5 * amax
    assert abs(vol.flat[-1] - f["v_last"]) <= 1e-5 * amax


def test_slab_and_roi_identities(oracle, kat):
    det = kat_det(oracle, kat)
    vg = oracle.calculate_volume_geometry(det)
    filtered = []
    vol = oracle.reconstruct(det, vg, 8, filtered_out=filtered)
    slab = oracle.reconstruct(det, vg, 8, v_dims=(31, 67, 67), v_offset=30)
    assert np.array_equal(slab, vol[30:61])  # bit-exact, as the reference
    roi = oracle.RegionOfInterest(8, 40, 4, 36, 10, 30)
    rv = oracle.reconstruct(det, vg, 8, v_dims=(20, 32, 32), roi=roi)
    assert np.array_equal(rv, vol[10:30, 4:36, 8:40])
    r = kat["roi"]
    assert abs(rv[10, 16, 16] - r["v_16_16_10"]) <= 1e-5 * np.abs(vol).max()
    assert close(rv.sum(dtype=np.float64), r["sum"], 1e-5)


def test_thread_count_independence(oracle, kat):
    det = kat_det(oracle, kat)
    vg = oracle.calculate_volume_geometry(det)
    n = oracle.lib().po_num_threads()
    try:
        oracle.lib().po_set_num_threads(1)
        a = oracle.reconstruct(det, vg, 3)
        oracle.lib().po_set_num_threads(max(2, n))
        b = oracle.reconstruct(det, vg, 3)
    finally:
        oracle.lib().po_set_num_threads(n)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("case", [0, 1])
def test_cube_geometries(oracle, kat, case):
    c = kat["cubes"][case]
    n, n_proj = c["n"], c["n_proj"]
    det = oracle.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 100, 200, 360.0 / n_proj)
    vg = oracle.calculate_volume_geometry(det)
    assert [vg.dim_x, vg.dim_y, vg.dim_z] == c["dims"]
    vol = oracle.reconstruct(det, vg, n_proj)
    assert close(vol.sum(dtype=np.float64), c["sum"], 1e-5)
    if "centre" in c:
        assert abs(vol[n // 2, n // 2, n // 2] - c["centre"]) <= 1e-5 * np.abs(vol).max()


@pytest.mark.skipif(not mkl_fftw.available(), reason="libmkl_rt.so (the survey's FFTW provider) not present")
def test_bit_exact_checksums_with_the_surveys_fft(oracle, kat):
    """With the same third-party FFT the survey's reference run linked, the oracle reproduces the reference's
    volume checksums bit for bit: full volume and the v_offset=30 slab. This pins weighting,
    filter generation, expand/multiply/shrink/normalise and backprojection exactly."""
    det = kat_det(oracle, kat)
    vg = oracle.calculate_volume_geometry(det)
    fs = oracle.filter_size(det.n_row)
    k = mkl_fftw.make_filter(oracle, fs, det.l_px_row)
    rf = mkl_fftw.RowFilter(fs, det.n_col)

    def recon(v_dims, off):
        vol = np.zeros(v_dims, np.float32)
        for i in range(8):
            p = oracle.lcg_projection(det.n_row, det.n_col, i)
            oracle.weight(p, det)
            rf.apply(p, k)
            s, c, ds, dt = oracle.backproject_constants(det, i)
            oracle.backproject(vol, p, off, det, vg, s, c, ds, dt)
        return vol

    assert survey_fnv(recon((61, 67, 67), 0), kat) == kat["full"]["fnv_mkl"]
    assert survey_fnv(recon((31, 67, 67), 30), kat) == kat["slab"]["fnv_mkl"]


def test_committed_golden_fixtures_are_the_oracles_output(oracle, golden_dir):
    """tests/golden/*.npz must be exactly what tests/golden/make_golden.py produces from the pinned oracle."""
    gold = np.load(os.path.join(golden_dir, "kat.npz"))
    det = oracle.DetectorGeometry(64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    vg = oracle.calculate_volume_geometry(det)
    p0 = oracle.lcg_projection(64, 48, 0)
    oracle.weight(p0, det)
    assert np.array_equal(p0, gold["weighted_p0"])
    filtered = []
    vol = oracle.reconstruct(det, vg, 8, filtered_out=filtered)
    assert np.array_equal(np.stack(filtered), gold["filtered"])
    assert np.array_equal(vol, gold["volume"])
    for n in (128, 1024, 2048, 4096):
        assert np.array_equal(oracle.make_filter(n, 0.2), gold["k_%d" % n])
    cube = np.load(os.path.join(golden_dir, "cube64.npz"))
    d = oracle.DetectorGeometry(64, 64, 0.2, 0.2, 0, 0, 100, 200, 45.0)
    v = oracle.reconstruct(d, oracle.calculate_volume_geometry(d), 8)
    assert np.array_equal(v[31:34], cube["slices"]) and v.sum(dtype=np.float64) == float(cube["sum"])
