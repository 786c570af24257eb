"""Regenerates the golden fixtures in this directory from the CPU oracle (oracle/), which is itself pinned to
the reference by tests/test_oracle_kat.py. Run from the repo root:  python tests/golden/make_golden.py

Fixtures are data only (inputs are regenerated from the LCG; expected outputs are stored):
  kat.npz     KAT geometry of SURVEY.md 8c: weighted projection 0, the 8 filtered projections, the full
              67x67x61 volume, the ramp filter K for N in {128, 1024, 2048, 4096}
  cube64.npz  64^3 / 8 projections: three central slices, sum and abs-sum
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    det = O.DetectorGeometry(64, 48, 0.2, 0.25, 1.5, -0.75, 100, 200, 45)
    vg = O.calculate_volume_geometry(det)
    p0 = O.lcg_projection(64, 48, 0)
    O.weight(p0, det)
    filtered = []
    vol = O.reconstruct(det, vg, 8, filtered_out=filtered)
    ks = {"k_%d" % n: O.make_filter(n, 0.2) for n in (128, 1024, 2048, 4096)}
    np.savez_compressed(os.path.join(HERE, "kat.npz"), weighted_p0=p0, filtered=np.stack(filtered), volume=vol, **ks)

    d = O.DetectorGeometry(64, 64, 0.2, 0.2, 0, 0, 100, 200, 45.0)
    g = O.calculate_volume_geometry(d)
    v = O.reconstruct(d, g, 8)
    np.savez_compressed(os.path.join(HERE, "cube64.npz"), slices=v[31:34].copy(),
                        sum=np.float64(v.sum(dtype=np.float64)), abssum=np.float64(np.abs(v).sum(dtype=np.float64)))
    for f in ("kat.npz", "cube64.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
