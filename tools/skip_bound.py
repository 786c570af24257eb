#!/usr/bin/env python3
"""How much of a launch's volume traffic could finer skipping still remove?

For a BASELINE geometry and a sample of projection angles: the share of voxels no ray reaches (the reference adds an exact 0
there: src/openmp/backprojection.cpp:52-84) voxel by voxel, and the share of the volume the kernel could leave untouched at the
granularities a kernel can test wave-uniformly -- a wave's 64 x 4 columns over the whole tile depth (what bp_tile_kernel skips
today: Column::none for all its columns), the same columns per group of 2 slices (one UNROLL step), and whole workgroup tiles
(64 x 16 x depth). Geometry in float64 with the reference's formulas (src/openmp/backprojection.cpp:116-139); a column's valid
slices are an interval (v is monotone in z), so the count is exact per column and cheap.

  python tools/skip_bound.py [c3|c2|c4|c5] [n_angles=48]
"""
import math
import sys

import numpy as np

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_angles = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n = {"c3": 2048, "c4": 2048, "c2": 1024, "c5": 2048}[which]
n_proj = {"c3": 1440, "c4": 1440, "c2": 720, "c5": 3600}[which]
l_px, d_so, d_od = 0.2, 500.0, 500.0
d_sd = d_so + d_od
alpha = math.atan((n * l_px / 2) / d_sd)
l_nat = d_so * math.sin(alpha) / ((n * l_px / 2) / l_px)
grid = 4096 if which == "c5" else n                      # the grid the coordinates refer to
l_vx = l_nat * n / grid
x0, y0, z0 = (1024, 1024, 1024) if which == "c5" else (0, 0, 0)   # ROI offset (config 5)
dx = dy = 2048 if which == "c5" else n
dz = 256 if which == "c4" else (2048 if which == "c5" else n)
slabs = range(8) if which == "c4" else [0]
tz = 8 if which == "c4" else 16                          # the tile kernel's tile depth for that slab shape

centre = lambda k, dim: -(dim * l_vx / 2) + l_vx / 2 + k * l_vx
xs = centre(np.arange(dx) + x0, grid)
ys = centre(np.arange(dy) + y0, grid)
min_h = -(n * l_px / 2)
min_v = -(n * l_px / 2)

tot = {"voxel": 0.0, "wave_2": 0.0, "wave_tile": 0.0, "wg_2": 0.0, "wg_tile": 0.0}
cnt = 0
for a in range(n_angles):
    phi = 2 * math.pi * (a * (n_proj // n_angles)) / n_proj
    s = xs[None, :] * math.cos(phi) + ys[:, None] * math.sin(phi)
    t = -xs[None, :] * math.sin(phi) + ys[:, None] * math.cos(phi)
    f = d_sd / (s + d_so)
    h = (t * f - min_h) / l_px - 0.5
    h_ok = (np.floor(h) >= 0) & (np.floor(h) + 1 < n)
    # v(z) = (z f - min_v) / l_px - 1/2 valid iff 0 <= floor(v) and floor(v) + 1 < n  <=>  0 <= v < n - 1
    # z = centre(m + z0 + off): m in [m_lo, m_hi)
    zc0 = centre(z0, grid)                              # z of slice index 0 of the allocated volume (before the slab offset)
    m_lo = np.ceil(((0.5 * l_px + min_v) / f - zc0) / l_vx - 1e-9)
    m_hi = np.ceil((((n - 1) + 0.5) * l_px + min_v) / f / l_vx - zc0 / l_vx - 1e-9)   # first invalid slice above
    for slab in slabs:
        off = slab * dz
        lo = np.clip(m_lo - off, 0, dz)
        hi = np.clip(m_hi - off, 0, dz)
        hi = np.where(h_ok, np.maximum(hi, lo), lo)      # columns with an invalid h have no valid slice
        lo = np.where(h_ok, lo, lo)
        valid = (hi - lo).sum()
        tot["voxel"] += 1.0 - valid / (dx * dy * dz)

        def touched(gy, gz):
            """share of the volume in groups of 64 x gy columns x gz slices that contain a valid voxel"""
            L = lo.reshape(dy // gy, gy, dx // 64, 64)
            H = hi.reshape(dy // gy, gy, dx // 64, 64)
            has = (H > L)
            any_ = has.any(axis=(1, 3))
            gl = np.where(has, L, dz).min(axis=(1, 3))
            gh = np.where(has, H, 0).max(axis=(1, 3))
            b0 = np.floor(gl / gz)
            b1 = np.ceil(gh / gz)
            blocks = np.where(any_, b1 - b0, 0).sum()
            return blocks * gz * gy * 64 / (dx * dy * dz)

        tot["wave_2"] += 1.0 - touched(4, 2)
        tot["wave_tile"] += 1.0 - touched(4, tz)
        tot["wg_2"] += 1.0 - touched(16, 2)
        tot["wg_tile"] += 1.0 - touched(16, tz)
        cnt += 1

print("# tools/skip_bound.py %s %d: share of the slab's voxels a projection's rays do not reach, mean over %d angles%s"
      % (which, n_angles, n_angles, " and the 8 slabs" if which == "c4" else ""))
print("voxel by voxel (the bound)                                  %.4f" % (tot["voxel"] / cnt))
print("a wave's 64 x 4 columns, per 2 slices                        %.4f" % (tot["wave_2"] / cnt))
print("a wave's 64 x 4 columns over the tile depth (%2d slices)      %.4f   <- what the kernel's wave-level skip can reach today" % (tz, tot["wave_tile"] / cnt))
print("a workgroup's 64 x 16 columns, per 2 slices                  %.4f" % (tot["wg_2"] / cnt))
print("a workgroup's 64 x 16 x %2d tile                              %.4f" % (tz, tot["wg_tile"] / cnt))
