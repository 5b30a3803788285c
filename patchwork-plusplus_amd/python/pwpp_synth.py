"""Seeded synthetic LiDAR frames in KITTI Velodyne layout (float32 x, y, z, intensity).

Used by tests and bench.py when the workload calls for frames other than the six KITTI
samples (dense 128-beam ~500k-point clouds) and as a source of
adversarial inputs for parity tests.  Pure numpy, deterministic for a given seed.
"""
import numpy as np


def make_cloud(seed, beams=64, azimuth_steps=2000, elev_deg=(-24.8, 2.0), sensor_height=1.723,
               n_boxes=40, max_range=120.0, range_noise=0.02, reflect_frac=0.02, undulation=0.15):
    """Ray-cast a spinning multi-beam sensor against an undulating ground plane and boxes.

    Returns an (N, 4) float32 array, N ~ 0.9 * beams * azimuth_steps.
    """
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], beams))
    az = np.linspace(0.0, 2.0 * np.pi, azimuth_steps, endpoint=False) + rng.uniform(0, 2 * np.pi / azimuth_steps)
    el_g, az_g = np.meshgrid(el, az, indexing="ij")
    el_g = el_g.ravel() + rng.normal(0.0, 2e-4, el_g.size)
    az_g = az_g.ravel()
    dx, dy, dz = np.cos(el_g) * np.cos(az_g), np.cos(el_g) * np.sin(az_g), np.sin(el_g)

    # ground: z = -h + A sin(2 pi x / L1) cos(2 pi y / L2) + slope
    slope = rng.normal(0.0, 0.01, 2)
    ph = rng.uniform(0, 2 * np.pi, 2)

    def ground_z(x, y):
        return (-sensor_height + undulation * np.sin(2 * np.pi * x / 20.0 + ph[0]) * np.cos(2 * np.pi * y / 27.0 + ph[1])
                + slope[0] * x + slope[1] * y)

    t = np.full(dx.size, np.inf)
    down = dz < -1e-3
    t0 = np.where(down, -sensor_height / np.where(down, dz, -1.0), np.inf)
    for _ in range(3):  # fixed-point refinement on the undulating surface
        tf0 = np.where(down, t0, 0.0)
        gz = ground_z(tf0 * dx, tf0 * dy)
        t0 = np.where(down, gz / np.where(down, dz, -1.0), np.inf)
    t = np.where(down & (t0 > 0), t0, np.inf)

    # axis-aligned boxes (cars / walls / poles)
    for _ in range(n_boxes):
        r = rng.uniform(5.0, 70.0)
        a = rng.uniform(0, 2 * np.pi)
        cx, cy = r * np.cos(a), r * np.sin(a)
        w, d, hgt = rng.uniform(0.3, 10.0), rng.uniform(0.3, 10.0), rng.uniform(1.0, 4.0)
        lo = np.array([cx - w / 2, cy - d / 2, -sensor_height - 0.2])
        hi = np.array([cx + w / 2, cy + d / 2, -sensor_height + hgt])
        with np.errstate(divide="ignore", invalid="ignore"):
            t1x, t2x = lo[0] / dx, hi[0] / dx
            t1y, t2y = lo[1] / dy, hi[1] / dy
            t1z, t2z = lo[2] / dz, hi[2] / dz
        tn = np.maximum(np.maximum(np.minimum(t1x, t2x), np.minimum(t1y, t2y)), np.minimum(t1z, t2z))
        tf = np.minimum(np.minimum(np.maximum(t1x, t2x), np.maximum(t1y, t2y)), np.maximum(t1z, t2z))
        hit = (tf >= tn) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)

    ok = np.isfinite(t) & (t < max_range)
    t = t[ok] + rng.normal(0.0, range_noise, ok.sum())
    x, y, z = t * dx[ok], t * dy[ok], t * dz[ok]
    inten = np.round(rng.uniform(0.0, 1.0, x.size), 2)

    # reflected-noise points: below the ground, close, dim (exercise RNR, ref :377-400)
    k = int(reflect_frac * x.size)
    if k:
        sel = rng.choice(x.size, k, replace=False)
        rr = rng.uniform(3.0, 8.0, k)
        aa = rng.uniform(0, 2 * np.pi, k)
        x[sel], y[sel] = rr * np.cos(aa), rr * np.sin(aa)
        z[sel] = -sensor_height - rng.uniform(0.9, 2.5, k)
        inten[sel] = np.round(rng.uniform(0.0, 0.15, k), 2)
    pts = np.stack([x, y, z, inten], axis=1).astype(np.float32)
    return pts


def make_dense_cloud(seed):
    """Config 5 of BASELINE.json: 128 beams x 4000 azimuth steps (~460-500k returns)."""
    return make_cloud(seed, beams=128, azimuth_steps=4000, elev_deg=(-25.0, 3.0))


def add_edge_cases(pts, seed=0):
    """Append points that sit exactly on the decision boundaries of pc2czm / RNR.

    Axis and diagonal directions (atan2 is an exact multiple of pi/4), points exactly at
    min_range / max_range and at the zone boundaries, the FLT_MIN tombstone value, far points
    and points at the origin.
    """
    rng = np.random.default_rng(seed + 12345)
    extra = []
    for r in (2.7, 2.7000001, 5.0, 12.3625, 22.025, 41.35, 79.99999, 80.0, 80.00001, 100.0, 1.0, 0.0):
        for ux, uy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)):
            s = np.hypot(ux, uy)
            x = np.float32(r * ux / s)
            y = np.float32(r * uy / s)
            if ux != 0 and uy != 0:
                y = np.float32(np.sign(uy)) * abs(x)  # |x| == |y| exactly in float32
            extra.append([x, y, -1.7 + rng.normal(0, 0.02), rng.uniform(0, 1)])
    extra.append([10.0, -0.0, -1.7, 0.5])
    extra.append([-10.0, -0.0, -1.7, 0.5])
    extra.append([0.0, 0.0, -1.7, 0.5])
    extra.append([5.0, 5.0, np.finfo(np.float32).tiny, 0.5])  # == FLT_MIN: dropped by the reference (:591)
    extra.append([4.0, 1.0, -3.5, 0.05])                      # RNR hit
    extra.append([4.0, 1.0, -3.5, 0.5])                       # same place, too bright for RNR
    extra = np.asarray(extra, np.float32)
    out = np.concatenate([pts, extra], axis=0)
    rng.shuffle(out, axis=0)
    return np.ascontiguousarray(out)
