"""paris_amd -- MI355X (gfx950) backend for the hzdr/PARIS FDK hot path: cosine weighting, ramp row filter and
voxel-driven cone-beam backprojection as hand-written HIP kernels behind a C ABI (include/paris_hip.h).

`paris_amd.backend` mirrors the reference's backend surface in Python; `paris_amd/host/` does the same in C++.
"""
from . import backend  # noqa: F401

__version__ = "0.1.0"
