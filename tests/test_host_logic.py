"""CPU-side checks of the product's host code: the C-ABI library loads and exports every symbol that
include/paris_hip.h declares, the stage-wrapper constants and geometry agree with the oracle, and the error
behaviour without a GPU is loud. No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from paris_amd import _lib
from paris_amd import backend as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "paris_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paris_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libparis_hip.so lacks %s" % n


def test_python_signatures_cover_the_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_struct_layouts_match_the_reference_types():
    assert C.sizeof(B.DetectorGeometry) == 2 * 4 + 7 * 4  # src/geometry.h:30-46
    assert C.sizeof(B.VolumeGeometry) == 6 * 4            # :48-57
    assert C.sizeof(B.SubvolumeGeometry) == 4 * 4         # :59-69
    assert C.sizeof(B.RegionOfInterest) == 6 * 4          # src/region_of_interest.h:30-38
    assert C.sizeof(B.SubvolumeInfo) == 5 * 4             # src/subvolume_information.h:30-34


GEOMETRIES = [
    (64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45),
    (512, 512, 0.2, 0.2, 0, 0, 500, 500, 1.0),
    (1024, 1024, 0.2, 0.2, 0, 0, 500, 500, 0.5),
    (2048, 2048, 0.2, 0.2, 0, 0, 500, 500, 0.25),
    (333, 217, 0.127, 0.254, -3.25, 2.5, 321.5, 123.25, 0.7),
]


@pytest.mark.parametrize("g", GEOMETRIES)
def test_volume_geometry_matches_oracle(oracle, g):
    a = B.calculate_volume_geometry(B.DetectorGeometry(*g))
    b = oracle.calculate_volume_geometry(oracle.DetectorGeometry(*g))
    assert (a.dim_x, a.dim_y, a.dim_z) == (b.dim_x, b.dim_y, b.dim_z)
    assert (a.l_vx_x, a.l_vx_y, a.l_vx_z) == (b.l_vx_x, b.l_vx_y, b.l_vx_z)


def test_natural_volume_sizes_of_the_bench_configs():
    for n in (1024, 2048):
        vg = B.calculate_volume_geometry(B.DetectorGeometry(n, n, 0.2, 0.2, 0, 0, 500, 500, 360.0 / n))
        assert (vg.dim_x, vg.dim_y) == (n, n)


def test_apply_roi_matches_oracle(oracle):
    vg = B.calculate_volume_geometry(B.DetectorGeometry(*GEOMETRIES[0]))
    og = oracle.calculate_volume_geometry(oracle.DetectorGeometry(*GEOMETRIES[0]))
    for roi in ((8, 40, 4, 36, 10, 30), (0, 40, 0, 36, 0, 30), (40, 8, 4, 36, 10, 30), (0, 67, 4, 36, 10, 30),
                (1, 67, 1, 67, 1, 61)):
        a = B.apply_roi(vg, *roi)
        b = oracle.apply_roi(og, oracle.RegionOfInterest(*roi))
        assert (a.dim_x, a.dim_y, a.dim_z) == (b.dim_x, b.dim_y, b.dim_z)


def test_filter_size_matches_oracle(oracle):
    for n in (1, 2, 3, 64, 65, 512, 1000, 1024, 1025, 2048, 4096):
        assert B.filter_size(n) == oracle.filter_size(n)


@pytest.mark.parametrize("g", GEOMETRIES)
def test_stage_angle_matches_oracle(oracle, g):
    det, odet = B.DetectorGeometry(*g), oracle.DetectorGeometry(*g)
    for idx in (0, 1, 7, 90, 359, 1439):
        s, c = B.stage_angle(det, idx)
        os_, oc, _, _ = oracle.backproject_constants(odet, idx)
        assert (s, c) == (os_, oc)
    s, c = B.stage_angle(det, 5, True, 33.25)
    os_, oc, _, _ = oracle.backproject_constants(odet, 5, True, 33.25)
    assert (s, c) == (os_, oc)


def test_no_device_is_reported_loudly():
    if B.get_devices():
        pytest.skip("a GPU is present")
    with pytest.raises(B.ParisHipError) as e:
        B.Backend(0)
    assert e.value.status == _lib.ERROR_NO_DEVICE


def test_invalid_arguments():
    L = _lib.load()
    assert L.paris_hip_device_count(None) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_ctx_create(0, None, 0, None) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_ctx_destroy(None) == _lib.SUCCESS
    assert L.paris_hip_weight(None, None, 0, 0, 0, 0, 0, 0, 0, 0) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_calculate_volume_geometry(None, None) == _lib.ERROR_INVALID_ARGUMENT
    assert b"invalid argument" in L.paris_hip_strerror(_lib.ERROR_INVALID_ARGUMENT)
    # the extensions: null ctx / null outputs are rejected before anything touches a device
    assert L.paris_hip_upload_projection(None, None, 0, None, 0, 0, 0) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_weight_rows(None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_stage_filter_rows(None, None, 0, 0, 0, 0, 0, None) == _lib.ERROR_INVALID_ARGUMENT
    det = B.DetectorGeometry(64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    vg = B.calculate_volume_geometry(det)
    first, count = C.c_uint32(), C.c_uint32()
    assert L.paris_hip_slab_row_band(C.byref(det), C.byref(vg), 67, 67, 61, 0, 0, None, None, C.byref(count)) == _lib.ERROR_INVALID_ARGUMENT
    assert L.paris_hip_slab_row_band(C.byref(det), C.byref(vg), 67, 67, 61, 0, 1, None, C.byref(first), C.byref(count)) \
        == _lib.ERROR_INVALID_ARGUMENT                       # ROI enabled but not given
    assert B.slab_row_band(det, vg, 67, 67, 0) == (0, 0)    # an empty slab reads nothing
    assert B.slab_row_band(det, vg, 67, 67, 61) == (0, 48)  # the whole field of view reads the whole detector
    # a source inside the volume's circle: no bound exists, the whole detector is reported
    near = B.DetectorGeometry(64, 48, 0.2, 0.25, 0.0, 0.0, 3.0, 200, 45)
    big = B.VolumeGeometry(64, 64, 64, 0.2, 0.2, 0.2)
    assert B.slab_row_band(near, big, 64, 64, 4, 10) == (0, 48)


def test_product_does_not_import_the_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "paris_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                raw = open(os.path.join(dirpath, f), "rb").read()
                if b"\0" in raw:
                    continue  # a built binary (e.g. the paris.hip executable), not source
                assert b"oracle" not in raw.lower(), os.path.join(dirpath, f)


REFERENCE = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree exists only in the build container")
def test_reference_wrappers_compile_against_the_hip_backend(tmp_path):
    """Drop-in check at the source level: the reference's own stage wrappers (the ones that need neither Boost nor
    FFTW) are compiled, unmodified and where they lie, against namespace paris::hip installed as src/hip/backend.h with
    the selector arm of INTEGRATION.md section 1. Syntax/semantic pass only (-fsyntax-only): nothing of the reference
    is copied into this repo, built into the product or run."""
    import subprocess
    src = tmp_path / "src"
    (src / "hip").mkdir(parents=True)
    for name in ("geometry.h", "projection.h", "volume.h", "region_of_interest.h", "subvolume_information.h", "exception.h",
                 "weighting.h", "weighting.cpp", "filtering.h", "filtering.cpp", "loader.h", "loader.cpp", "make_volume.h",
                 "make_volume.cpp", "backprojection.h"):
        os.symlink(os.path.join(REFERENCE, name), src / name)
    os.symlink(os.path.join(ROOT, "paris_amd", "host", "paris", "hip", "backend.h"), src / "hip" / "backend.h")
    # the selector a maintainer gets after adding the PARIS_ENABLE_HIP arm: only the chosen arm matters
    (src / "backend.h").write_text('#ifndef PARIS_BACKEND_H_\n#define PARIS_BACKEND_H_\n#include "hip/backend.h"\n'
                                   "namespace paris { namespace backend = hip; }\n#endif\n")
    for unit in ("weighting.cpp", "filtering.cpp", "loader.cpp", "make_volume.cpp"):
        r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-DPARIS_ENABLE_HIP", "-DPARIS_HIP_INSIDE_PARIS",
                            "-I", os.path.join(ROOT, "include"), str(src / unit)], capture_output=True, text=True)
        assert r.returncode == 0, unit + "\n" + r.stderr
    # backprojection.h declares paris::backproject over backend types: its signature must be expressible too
    probe = src / "probe.cpp"
    probe.write_text('#include "backprojection.h"\n#include "hip/backend.h"\n'
                     "void f(paris::backend::projection_device_type& p, paris::backend::volume_device_type& v,\n"
                     "       const paris::detector_geometry& d, const paris::volume_geometry& g, const paris::region_of_interest& r)\n"
                     "{ paris::backend::backproject(p, v, 0u, d, g, false, r, 0.f, 1.f, 0.f, 0.f); auto k = paris::backend::make_filter(8u, 1.f);\n"
                     "  paris::backend::apply_filter(p, k, 8u, 1u); auto dev = paris::backend::get_devices(); paris::backend::set_device(dev[0]);\n"
                     "  auto s = paris::backend::make_subvolume_information(g, d); (void)s; }\n")
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-DPARIS_ENABLE_HIP", "-DPARIS_HIP_INSIDE_PARIS", "-I",
                        os.path.join(ROOT, "include"), str(probe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _rows_touched(det, vg, dims, v_offset, roi, sin, cos):
    """fp32 restatement of the v coordinate of src/openmp/backprojection.cpp:116-133: (lowest, highest) detector row any
    valid tap of the slab reads for this angle, or None."""
    f32 = np.float32
    dz, dy, dx = dims
    rx, ry, rz = (roi.x1, roi.y1, roi.z1) if roi is not None else (0, 0, 0)

    def centred(coord, dim, size):
        size2 = f32(size) / f32(2)
        return -(f32(dim) * size2) + size2 + coord.astype(f32) * f32(size)

    x = centred(np.arange(dx) + rx, vg.dim_x, vg.l_vx_x)[None, :]
    y = centred(np.arange(dy) + ry, vg.dim_y, vg.l_vx_y)[:, None]
    z = centred(np.array([0, dz - 1]) + rz + v_offset, vg.dim_z, vg.l_vx_z)
    s = x * f32(cos) + y * f32(sin)
    factor = (f32(abs(det.d_so)) + f32(abs(det.d_od))) / (s + f32(det.d_so))
    size2 = f32(det.l_px_col) / f32(2)
    vmin = -(f32(det.n_col) * size2) - f32(det.delta_t) * f32(det.l_px_col)
    lo, hi = None, None
    for zz in z:  # v is monotonic in z for a fixed column: the end slices bound every slice in between
        v = (zz * factor - vmin) / f32(det.l_px_col) - f32(0.5)
        v1 = np.floor(v)
        ok = (v1 >= 0) & (v1 + 1 < det.n_col)
        if ok.any():
            a, b = int(v1[ok].min()), int(v1[ok].max()) + 1
            lo = a if lo is None else min(lo, a)
            hi = b if hi is None else max(hi, b)
    return None if lo is None else (lo, hi)


@pytest.mark.parametrize("seed", range(40))
def test_slab_row_band_covers_every_tap(seed):
    """f4: paris_hip_slab_row_band must contain every detector row a valid tap of the slab reads, for any angle; for a
    slab whose end slices straddle the band the bound must also be useful (not the whole detector for thin slabs)."""
    rng = np.random.default_rng(7000 + seed)
    n_row, n_col = int(rng.integers(24, 300)), int(rng.integers(16, 300))
    l_r, l_c = float(rng.choice([0.1, 0.2, 0.127, 0.4])), float(rng.choice([0.1, 0.2, 0.25, 0.4]))
    d_so, d_od = float(rng.uniform(40, 600)), float(rng.uniform(20, 600))
    det = B.DetectorGeometry(n_row, n_col, l_r, l_c, float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6)), d_so, d_od, 1.0)
    nat = B.calculate_volume_geometry(det)
    full = [int(rng.integers(20, 160)) for _ in range(3)]  # z, y, x
    scale = [float(nat.l_vx_x * rng.uniform(0.4, 2.5)) for _ in range(3)]
    half_diag = 0.5 * np.hypot(full[2] * scale[0], full[1] * scale[1])
    if half_diag > 0.8 * d_so:
        scale[0] *= 0.8 * d_so / half_diag
        scale[1] *= 0.8 * d_so / half_diag
    vg = B.VolumeGeometry(full[2], full[1], full[0], *scale)
    roi = None
    out = tuple(full)
    if seed % 2:
        x1, y1, z1 = (int(rng.integers(1, d // 3)) for d in (full[2], full[1], full[0]))
        x2, y2, z2 = (int(rng.integers(2 * d // 3, d)) for d in (full[2], full[1], full[0]))
        roi = B.RegionOfInterest(x1, x2, y1, y2, z1, z2)
        out = (z2 - z1, y2 - y1, x2 - x1)
    n_slabs = int(rng.integers(1, 9))
    dz = max(1, out[0] // n_slabs)
    for g in range(n_slabs):
        v_offset = g * dz
        dims = (dz if g < n_slabs - 1 else out[0] - v_offset, out[1], out[2])
        first, count = B.slab_row_band(det, vg, dims[2], dims[1], dims[0], v_offset, roi)
        assert first % 2 == 0 and first + count <= n_col and (count == 0 or (first + count) % 2 == 0 or first + count == n_col)
        lo, hi = None, None
        for phi in list(rng.uniform(0, 2 * np.pi, 12)) + [0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi]:
            r = _rows_touched(det, vg, dims, v_offset, roi, np.float32(np.sin(phi)), np.float32(np.cos(phi)))
            if r is not None:
                lo = r[0] if lo is None else min(lo, r[0])
                hi = r[1] if hi is None else max(hi, r[1])
        if lo is None:
            continue
        assert first <= lo and hi < first + count, (seed, g, (first, count), (lo, hi))
    # the z-slabs of the whole field of view at the bench geometry: the band is a fraction of the detector
    det = B.DetectorGeometry(2048, 2048, 0.2, 0.2, 0.0, 0.0, 500.0, 500.0, 0.25)
    nat = B.calculate_volume_geometry(det)
    vg = B.VolumeGeometry(2048, 2048, 2048, *([float(np.float32(nat.l_vx_x))] * 3))
    counts = [B.slab_row_band(det, vg, 2048, 2048, 256, 256 * g)[1] for g in range(8)]
    assert max(counts) < 0.4 * 2048 and counts == counts[::-1]
    assert B.slab_row_band(det, vg, 2048, 2048, 2048, 0) == (0, 2048)


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/paris_hip.h must compile as strict C99 and as C++11, and a C program
    must link against the library by name."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "paris_hip.h"\n'
                   'int main(void) { paris_detector_geometry g; paris_volume_geometry v; g.n_row = 64; g.n_col = 48;\n'
                   '  g.l_px_row = 0.2f; g.l_px_col = 0.25f; g.delta_s = 1.5f; g.delta_t = -0.75f; g.d_so = 100.f; g.d_od = 200.f;\n'
                   '  g.delta_phi = 45.f; if(paris_hip_calculate_volume_geometry(&g, &v)) return 2;\n'
                   '  return (v.dim_x == 67 && v.dim_z == 61 && paris_hip_filter_size(64) == 128) ? 0 : 1; }\n')
    inc = os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o",
                           str(tmp_path / "c.o")])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src),
                           "-o", str(tmp_path / "cxx.o")])
    lib_dir = os.path.join(root, "paris_amd", "lib")
    exe = tmp_path / "hdr"
    subprocess.check_call(["gcc", str(tmp_path / "c.o"), "-o", str(exe), "-L", lib_dir, "-lparis_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.run([str(exe)]).returncode == 0  # geometry entry points need no device


def test_fast_path_row_weight_identity():
    """backproject.hip's fast path uses 1 - (v - floor v) for (floor v + 1) - v (src/openmp/backprojection.cpp:80): the
    two are the same fp32 number for every valid tap, v in [0, 2^24). Every 5th float below 4 and 10 M random ones above."""
    f = np.float32

    def mismatches(v):
        y1 = np.floor(v)
        return int(np.count_nonzero(((y1 + f(1)) - v).view(np.uint32) != (f(1) - (v - y1)).view(np.uint32)))

    bits = np.arange(0, int(np.float32(4).view(np.uint32)), 5, dtype=np.uint32)
    assert mismatches(bits.view(np.float32)) == 0
    rng = np.random.default_rng(1)
    v = (rng.random(10_000_000) * rng.choice([8, 100, 5000, 70000, 2 ** 20, 2 ** 24 - 2], 10_000_000)).astype(np.float32)
    assert mismatches(v) == 0


def test_the_product_build_ships_no_experiments():
    """two builds of one C ABI. The product library holds the kernels its launcher picks by itself; the slice
    kernel, the two-pass variant, the first radix-16 filter kernel and the getenv A/B switches exist only in the experiments build
    (make EXPERIMENTS=1), which exports the same symbols; a PARIS_*TIMING* macro (wrong results, timing only) cannot be compiled
    into the product at all."""
    import ctypes
    import subprocess
    product = os.path.join(ROOT, "paris_amd", "lib", "libparis_hip.so")
    experiments = _lib.EXPERIMENTS_LIB_PATH
    assert os.path.exists(experiments), "make -C paris_amd/csrc EXPERIMENTS=1 (__graft_entry__.build() does)"
    p, x = ctypes.CDLL(product), ctypes.CDLL(experiments)
    assert p.paris_hip_has_experiments() == 0 and x.paris_hip_has_experiments() == 1
    for name in _lib.SIGNATURES:
        getattr(x, name)
    pb, xb = open(product, "rb").read(), open(experiments, "rb").read()
    for marker in (b"bp_slice_kernel", b"bp_column_state_kernel", b"apply_filter_r16_kernel", b"PARIS_TILE_NEST", b"PARIS_FUSED_XFAST",
                   b"PARIS_HIP_UPLOAD_STREAM"):
        assert marker not in pb and marker in xb, marker
    assert len(pb) < 0.8 * len(xb)
    for src, macro in (("backproject_fused.hip", "PARIS_TIMING_ONLY_STAGE_ONCE"), ("filter_fused.hip", "PARIS_FILTER_TIMING_NO_MEMORY")):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-D" + macro, "-I" + os.path.join(ROOT, "include"),
                            "-I" + os.path.join(ROOT, "paris_amd", "csrc"), os.path.join(ROOT, "paris_amd", "csrc", src)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "experiments build only" in r.stderr, (src, r.stderr[-500:])
