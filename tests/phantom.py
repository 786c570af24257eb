"""Analytic cone-beam projections of the 3-D Shepp-Logan head phantom (test input generator, no reference counterpart).

Geometry conventions are those of the backprojector (src/openmp/backprojection.cpp:116-133): for view angle phi a point
(x, y, z) has s = x cos + y sin, t = -x sin + y cos; the source sits at s = -d_so, the detector plane at s = +d_od, and
detector pixel (i, j) is centred at t = (i + 0.5) * l_px_row - n_row * l_px_row / 2 (+ offsets), z likewise.
"""
import numpy as np

# (value, a, b, c, x0, y0, z0, rotation about z in degrees) -- the usual 10-ellipsoid table, unit-sphere coordinates
ELLIPSOIDS = [
    (1.00, .6900, .920, .810, 0., 0., 0., 0.),
    (-.80, .6624, .874, .780, 0., -.0184, 0., 0.),
    (-.20, .1100, .310, .220, .22, 0., 0., -18.),
    (-.20, .1600, .410, .280, -.22, 0., 0., 18.),
    (.10, .2100, .250, .410, 0., .35, -.15, 0.),
    (.10, .0460, .046, .050, 0., .1, .25, 0.),
    (.10, .0460, .046, .050, 0., -.1, .25, 0.),
    (.10, .0460, .023, .050, -.08, -.605, 0., 0.),
    (.10, .0230, .023, .020, 0., -.606, 0., 0.),
    (.10, .0230, .046, .020, .06, -.605, 0., 0.),
]


def projection(n_row, n_col, l_px_row, l_px_col, d_so, d_od, phi_deg, radius_mm, delta_s=0.0, delta_t=0.0):
    """Line integrals (mm) for one view; returns float32 (n_col, n_row)."""
    phi = np.deg2rad(phi_deg)
    c, s_ = np.cos(phi), np.sin(phi)
    t = (np.arange(n_row) + 0.5) * l_px_row - n_row * l_px_row / 2 - delta_s * l_px_row
    z = (np.arange(n_col) + 0.5) * l_px_col - n_col * l_px_col / 2 - delta_t * l_px_col
    T, Z = np.meshgrid(t, z)                    # (n_col, n_row)
    # ray: P(l) = S + l * D in (s, t, z); to (x, y, z): x = s c - t s_, y = s s_ + t c
    D = np.stack([np.full_like(T, d_so + d_od), T, Z], -1)
    D /= np.linalg.norm(D, axis=-1, keepdims=True)
    Dx = D[..., 0] * c - D[..., 1] * s_
    Dy = D[..., 0] * s_ + D[..., 1] * c
    Dz = D[..., 2]
    Sx, Sy, Sz = -d_so * c, -d_so * s_, 0.0
    out = np.zeros_like(T)
    for val, a, b, cc, x0, y0, z0, rot in ELLIPSOIDS:
        r = np.deg2rad(rot)
        cr, sr = np.cos(r), np.sin(r)
        # into the ellipsoid's frame, scaled to a unit sphere
        px, py, pz = Sx - x0 * radius_mm, Sy - y0 * radius_mm, Sz - z0 * radius_mm
        ox, oy = (px * cr + py * sr), (-px * sr + py * cr)
        dx, dy = (Dx * cr + Dy * sr), (-Dx * sr + Dy * cr)
        ox, oy, oz = ox / (a * radius_mm), oy / (b * radius_mm), pz / (cc * radius_mm)
        dx, dy, dz = dx / (a * radius_mm), dy / (b * radius_mm), Dz / (cc * radius_mm)
        A = dx * dx + dy * dy + dz * dz
        Bq = ox * dx + oy * dy + oz * dz
        Cq = ox * ox + oy * oy + oz * oz - 1.0
        disc = Bq * Bq - A * Cq
        out += val * np.where(disc > 0, 2.0 * np.sqrt(np.maximum(disc, 0)) / A, 0.0)
    return out.astype(np.float32)
