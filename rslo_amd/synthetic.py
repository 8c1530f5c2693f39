"""Synthetic KITTI-shaped LiDAR scans (measurement input).

An HDL-64-like sensor (64 rings x 2083 azimuth steps, elevation +2..-24.8 deg, 1.73 m above
a ground plane) looks at a fixed random street canyon (world seed 123): piecewise walls at
|y| in [8,14] m in 10 m slabs and 40 boxes on the ground.  Rays keep the first hit, the range
gate is 2.5..80 m and each coordinate gets N(0, 1 cm) noise.  Points come out ring-major,
which is what the first-come voxel numbering and the max_voxels cap depend on.

Per point 7 floats, the layout the reference's HDF5 reader produces
(rslo/data/kitti_dataset_hdf5.py:253-261): x, y, z, intensity, nx, ny, nz -- with the
dataset's rule that components of exactly-vertical normals are zeroed (:261).

`scan()` reproduces the counts SURVEY.md quotes: 131 484 points for the default sensor and
263 129 for n_el=128.
"""
import numpy as np

SENSOR_H = 1.73
PC_RANGE = np.array([-70.4, -38.4, -3.0, 70.4, 38.4, 5.0], np.float32)
VOXEL_SIZE = np.array([0.1, 0.1, 0.2], np.float32)
VOXEL_SIZE_DENSE = np.array([0.1, 0.1, 0.1], np.float32)
MAX_POINTS_PER_VOXEL = 10
MAX_VOXELS = 40000


def _rays(n_az, n_el, yaw):
    el = np.deg2rad(np.linspace(2.0, -24.8, n_el))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    E, A = np.meshgrid(el, az, indexing="ij")
    d_sensor = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    if yaw == 0.0:
        return d_sensor, d_sensor
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return d_sensor, d_sensor @ Rz.T


def scan(n_az=2083, n_el=64, pose_xy=(0.0, 0.0), yaw=0.0, scan_seed=0, world_seed=123, h=SENSOR_H,
         with_features=True):
    """One scan in the sensor frame.  Returns [P,7] float32 (or [P,3] if not with_features)."""
    d_s, d = _rays(n_az, n_el, yaw)
    n = len(d)
    o = np.array([pose_xy[0], pose_xy[1], 0.0])
    t = np.full(n, np.inf)
    nrm = np.zeros((n, 3))
    # ground plane z = -h
    tg = np.where(d[:, 2] < -1e-3, -h / np.where(d[:, 2] < -1e-3, d[:, 2], -1.0), np.inf)
    hit = tg < t
    t = np.where(hit, tg, t)
    nrm[hit] = (0.0, 0.0, 1.0)
    w = np.random.default_rng(world_seed)  # the world is drawn in this fixed order
    for ysign in (-1, 1):
        for x0 in np.arange(-80, 80, 10.0):
            yw = ysign * (8 + 6 * w.random())
            top = 2 + 6 * w.random()
            dy = np.where(np.abs(d[:, 1]) > 1e-6, d[:, 1], 1e-6)
            tw = (yw - o[1]) / dy
            px = o[0] + tw * d[:, 0]
            pz = tw * d[:, 2]
            ok = (tw > 0) & (px >= x0) & (px < x0 + 10) & (pz > -h) & (pz < top) & (tw < t)
            t = np.where(ok, tw, t)
            nrm[ok] = (0.0, -float(ysign), 0.0)
    for _ in range(40):
        cx, cy = w.uniform(-60, 60), w.uniform(-7, 7)
        sx, sy, sz = w.uniform(1.5, 4.5), w.uniform(1.5, 2), w.uniform(1.3, 2)
        lo = np.array([cx - sx / 2, cy - sy / 2, -h]) - o
        hi = np.array([cx + sx / 2, cy + sy / 2, -h + sz]) - o
        inv = 1 / np.where(np.abs(d) > 1e-9, d, 1e-9)
        t0, t1 = lo * inv, hi * inv
        tmin = np.minimum(t0, t1)
        tn = tmin.max(1)
        tf = np.maximum(t0, t1).min(1)
        ok = (tf >= tn) & (tn > 0.5) & (tn < t)
        t = np.where(ok, tn, t)
        ax = tmin.argmax(1)
        fn = np.zeros((n, 3))
        fn[np.arange(n), ax] = -np.sign(d[np.arange(n), ax])
        nrm[ok] = fn[ok]
    keep = np.isfinite(t) & (t < 80) & (t > 2.5)
    rng = np.random.default_rng(scan_seed)
    xyz = (d_s[keep] * t[keep, None] + rng.normal(0, 0.01, (int(keep.sum()), 3))).astype(np.float32)
    if not with_features:
        return xyz
    inten = rng.random(len(xyz)).astype(np.float32)
    nk = nrm[keep]
    if yaw != 0.0:  # normals into the sensor frame
        c, s = np.cos(-yaw), np.sin(-yaw)
        Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        nk = nk @ Rz.T
    nk = nk.astype(np.float32)
    # dataset rule (kitti_dataset_hdf5.py:261): components equal to |[0,0,1]| element-wise are zeroed
    vert = np.abs(nk) == np.array([0, 0, 1], np.float32)
    nk = np.where(vert, np.float32(0), nk)
    return np.concatenate([xyz, inten[:, None], nk], axis=1).astype(np.float32)


def frame_pair(sample_idx=0, n_el=64, n_az=2083):
    """Two consecutive scans of the same world: frame 2 is advanced by
    t = (0.8 + 0.4u, 0.02u, 0) m and yaw 0.5 deg * u, u ~ U(-1, 1) (seed 1000 + sample_idx).
    Returns (points0, points1, motion) with motion = (tx, ty, yaw)."""
    r = np.random.default_rng(1000 + sample_idx)
    u = r.uniform(-1, 1, 3)
    tx, ty, yaw = 0.8 + 0.4 * u[0], 0.02 * u[1], np.deg2rad(0.5) * u[2]
    x0 = 3.0 * sample_idx
    p0 = scan(n_az, n_el, (x0, 0.0), 0.0, scan_seed=2 * (1000 + sample_idx))
    p1 = scan(n_az, n_el, (x0 + tx, ty), yaw, scan_seed=2 * (1000 + sample_idx) + 1)
    return p0, p1, (tx, ty, yaw)


def small_cloud(n=4000, seed=0, extent=(20.0, 10.0, 2.0)):
    """A cheap random surface-ish cloud for unit tests: [n,7] (xyz, intensity, unit normal)."""
    r = np.random.default_rng(seed)
    xyz = (r.random((n, 3)) * 2 - 1) * np.asarray(extent)
    xyz[:, 2] = 0.3 * np.sin(xyz[:, 0] * 0.7) + 0.2 * np.cos(xyz[:, 1] * 1.3) + 0.05 * r.normal(size=n)
    nrm = r.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.concatenate([xyz, r.random((n, 1)), nrm], 1).astype(np.float32)
