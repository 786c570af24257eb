"""The oracle's reference pin, re-run on the GPU box.

The `-m gpu` tests trust oracle/libparis_oracle.so as the checker, and that library is (re)built by whatever gcc
the GPU box has. This module selects the MKL-free known-answer checks of tests/test_oracle_kat.py -- SURVEY.md 8c's
values of the reference's OpenMP path: weighted-projection FNV over every pixel, K values, filtered values, volume
sums and spot values, slab == slices, ROI == crop, thread-count independence, the cube geometries -- under the `gpu`
marker too, so the box that runs the parity tests also records that its oracle build reproduces the pin. (The same
functions run in the CPU suite under `-m "not gpu"`; nothing here touches the GPU.) The one check that goes through MKL's
FFTW3 interface stays in the CPU suite only: MKL picks its FFT code path by host CPU, so its last bits -- and with them the
survey's two volume FNVs -- reproduce only on the CPU type the survey ran on (this build container), not on the GPU box.)"""
import pytest

import test_oracle_kat as K

pytestmark = pytest.mark.gpu

kat = K.kat  # module-scoped fixture: tests/golden/survey_kat.json

test_volume_geometry = K.test_volume_geometry
test_apply_roi_rule = K.test_apply_roi_rule
test_weight_bit_exact = K.test_weight_bit_exact
test_filter_size = K.test_filter_size
test_make_filter_values = K.test_make_filter_values
test_filtered_projection_values = K.test_filtered_projection_values
test_full_volume_values = K.test_full_volume_values
test_slab_and_roi_identities = K.test_slab_and_roi_identities
test_thread_count_independence = K.test_thread_count_independence
test_cube_geometries = K.test_cube_geometries
test_committed_golden_fixtures_are_the_oracles_output = K.test_committed_golden_fixtures_are_the_oracles_output
