"""The experiments build (make -C paris_amd/csrc EXPERIMENTS=1 -> paris_amd/lib/libparis_hip_experiments.so).

The product library ships the kernels the launcher picks by itself; what was measured and lost -- the slice kernel, the two-pass
variant, unroll 3 / 4 bodies, tile orders 0 / 1 / 8 / 9 / 12, the first radix-16 filter kernel, the getenv A/B switches -- is compiled
into a second library with the same C ABI, for tools/ and for these tests. The variant tests of tests/test_gpu_parity.py assert
PARIS_HIP_ERROR_UNSUPPORTED against the product and bit-equality with the oracle against the experiments build; here they are run
against the latter (a child pytest with PARIS_HIP_LIBRARY set)."""
import os
import re
import subprocess
import sys

import pytest

from paris_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_product_build_has_no_experiments():
    assert "PARIS_HIP_LIBRARY" in os.environ or not _lib.has_experiments()


def test_variant_tests_against_the_experiments_build():
    assert os.path.exists(_lib.EXPERIMENTS_LIB_PATH), "build it: make -C paris_amd/csrc EXPERIMENTS=1 (__graft_entry__.build() does)"
    env = dict(os.environ, PARIS_HIP_LIBRARY=_lib.EXPERIMENTS_LIB_PATH)
    pick = ("apply_filter or kat_full or every_kernel_shape or every_tile_order or two_pass or slice_kernel or random_geometries "
            "or batch_equals_sequence or fused_batch_bit_exact or deferred_backprojection_is_bit_identical")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", pick,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1700, env=env, cwd=ROOT)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) > 150 and "skipped" not in r.stdout.splitlines()[-1], tail
    # ... and it really was the other library
    r = subprocess.run([sys.executable, "-c", "from paris_amd import _lib; print(_lib.has_experiments(), _lib.LIB_PATH)"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.stdout.split()[0] == "True" and r.stdout.split()[1].endswith("libparis_hip_experiments.so"), r.stdout + r.stderr
