"""Seeded synthetic inputs of the shapes SURVEY.md section 8(d) lists (there is no dataset and no
network): textured frames, depth maps, camera poses, descriptor databases with planted matches and
PnP correspondence sets.  Pure NumPy; used by tests/ and bench.py."""
from __future__ import annotations

import numpy as np

FX = FY = 320.0
CX, CY = 320.0, 240.0


def textured_frame(rng, w=640, h=480, n_shapes=400, noise=4.0):
    """mid-grey canvas + random filled rectangles/discs + N(0, noise): many FAST corners."""
    img = np.full((h, w, 3), 128, np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for k in range(n_shapes):
        col = rng.integers(0, 256, 3).astype(np.float32)
        sz = rng.integers(6, 60, 2)
        x0 = int(rng.integers(0, w - 6)); y0 = int(rng.integers(0, h - 6))
        if k % 3 == 0:
            r = int(sz[0]) // 2 + 3
            m = (xx[max(y0 - r, 0):y0 + r + 1, max(x0 - r, 0):x0 + r + 1] - x0) ** 2 + \
                (yy[max(y0 - r, 0):y0 + r + 1, max(x0 - r, 0):x0 + r + 1] - y0) ** 2 <= r * r
            img[max(y0 - r, 0):y0 + r + 1, max(x0 - r, 0):x0 + r + 1][m] = col
        else:
            img[y0:y0 + sz[1], x0:x0 + sz[0]] = col
    if noise > 0:
        img += rng.normal(0, noise, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def ground_depth_mm(rng, w=640, h=480, zeros=0.02):
    """tilted ground plane 0.8..14 m + N(0, 10 mm), a few zero holes; uint16 millimetres."""
    v = np.linspace(14.0, 0.8, h)[:, None] * np.ones((1, w))
    d = v * 1000.0 + rng.normal(0, 10.0, (h, w))
    d = np.clip(d, 0, 65535).astype(np.uint16)
    d[rng.random((h, w)) < zeros] = 0
    return d


def random_descriptors(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def perturb_descriptors(rng, base, flip_p=0.08):
    """copies of `base` with Binomial(256, flip_p) flipped bits per row"""
    bits = np.unpackbits(base, axis=1)
    flips = rng.random(bits.shape) < flip_p
    return np.packbits(bits ^ flips, axis=1)


def quat_from_yaw_pitch_roll(yaw, pitch=0.0, roll=0.0):
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def rodrigues(rvec):
    rvec = np.asarray(rvec, np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pnp_problem(rng, m=200, outlier_ratio=0.4, noise_px=0.0, rvec=None, tvec=None):
    """3-D points in the frustum z in [0.5, 15], a pose with |t| < 2 m and angle < 20 deg, pixel noise
    and gross outliers.  Returns obj(m,3) f32, img(m,2) f32, rvec, tvec, inlier mask."""
    if rvec is None:
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        rvec = ax * np.deg2rad(rng.uniform(0, 20))
    if tvec is None:
        tvec = rng.uniform(-1, 1, 3) * np.array([1.0, 0.5, 1.0])
    R = rodrigues(rvec)
    obj = []
    while len(obj) < m:
        z = rng.uniform(0.5, 15.0)
        x = rng.uniform(-1, 1) * z; y = rng.uniform(-0.75, 0.75) * z
        p = np.array([x, y, z])
        pc = R @ p + tvec
        if pc[2] > 0.3:
            u = FX * pc[0] / pc[2] + CX; v = FY * pc[1] / pc[2] + CY
            if 0 <= u < 640 and 0 <= v < 480:
                obj.append(p)
    obj = np.array(obj, np.float32)
    pc = (R @ obj.astype(np.float64).T).T + tvec
    uv = np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], 1)
    if noise_px > 0:
        uv += rng.normal(0, noise_px, uv.shape)
    out = rng.random(m) < outlier_ratio
    shift = rng.uniform(20, 200, (m, 2)) * rng.choice([-1, 1], (m, 2))
    uv[out] += shift[out]
    return obj, uv.astype(np.float32), np.asarray(rvec, np.float64), np.asarray(tvec, np.float64), ~out


def descriptor_db(rng, n_records, rows="fixed64", cur=None, planted_records=(), flip_p=0.08):
    """Packed database arena: desc (T,32) u8, pts3d (T,3) f32, offsets (L+1) i64, poses (L,7) f64.
    rows: "fixed64" | "ragged" (clip(round(N(60,25)),30,500)) | int.  Records listed in
    planted_records get rows that are noisy copies of rows of `cur` (true matches)."""
    if rows == "fixed64":
        n = np.full(n_records, 64, np.int64)
    elif rows == "ragged":
        n = np.clip(np.rint(rng.normal(60, 25, n_records)), 30, 500).astype(np.int64)
    else:
        n = np.full(n_records, int(rows), np.int64)
    off = np.zeros(n_records + 1, np.int64)
    off[1:] = np.cumsum(n)
    T = int(off[-1])
    desc = random_descriptors(rng, T)
    pts3d = np.stack([rng.uniform(-4, 4, T), rng.uniform(-2, 2, T), rng.uniform(1, 12, T)], 1).astype(np.float32)
    poses = np.zeros((n_records, 7))
    poses[:, 0] = np.arange(n_records) * 2.0          # a straight teach route, 2 m apart
    poses[:, 2] = 0.3
    poses[:, 3:] = np.array([0.5, -0.5, 0.5, -0.5])   # camera optical frame looking along world +x
    if cur is not None:
        for r in planted_records:
            k = int(min(n[r], len(cur)))
            src = rng.choice(len(cur), k, replace=False)
            desc[off[r]:off[r] + k] = perturb_descriptors(rng, cur[src], flip_p)
    return desc, pts3d, off, poses


# ---------------------------------------------------------------------------------------------------
# A renderable scene: one large textured wall in front of the robot, seen by a pinhole camera mounted
# on base_link (0.35 m forward, 0.18 m up, optical frame right-down-forward).
class WallScene:
    """world: x forward, y left, z up.  The wall is the plane x = wall_x; its texture is metric
    (`m_per_px` metres per texel).  render(base_pose) -> (bgr (H,W,3) u8, depth_mm (H,W) u16)."""

    # optical axes expressed in base_link coordinates (columns): x_cam = -y, y_cam = -z, z_cam = +x
    R_BASE_CAM = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    T_BASE_CAM = np.array([0.35, 0.0, 0.18])

    def __init__(self, seed=20260501, wall_x=16.0, m_per_px=0.02, tex_w=2400, tex_h=1200, w=640, h=480, noise=2.0):
        rng = np.random.default_rng(seed)
        self.seed = seed
        self.tex = textured_frame(rng, tex_w, tex_h, n_shapes=2600, noise=0.0).astype(np.float32)
        self.wall_x, self.m_per_px, self.w, self.h, self.noise = wall_x, m_per_px, w, h, noise
        v, u = np.mgrid[0:h, 0:w]
        self.rays = np.stack([(u - CX) / FX, (v - CY) / FY, np.ones_like(u, dtype=np.float64)], axis=-1)  # camera frame

    @staticmethod
    def _quat_to_rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def render(self, base_pose):
        R_wb = self._quat_to_rot(base_pose[3:7])
        c = np.asarray(base_pose[:3], np.float64) + R_wb @ self.T_BASE_CAM
        R_wc = R_wb @ self.R_BASE_CAM
        d = self.rays @ R_wc.T                                  # ray directions in the world
        lam = (self.wall_x - c[0]) / np.where(np.abs(d[..., 0]) < 1e-9, 1e-9, d[..., 0])
        hit = lam > 0
        py = c[1] + lam * d[..., 1]
        pz = c[2] + lam * d[..., 2]
        th, tw = self.tex.shape[:2]
        tu = (tw / 2.0) - py / self.m_per_px                    # +y (left) -> smaller texture u
        tv = (th / 2.0) - (pz - 1.0) / self.m_per_px            # wall centred 1 m above ground
        inside = hit & (tu >= 0) & (tu < tw - 1) & (tv >= 0) & (tv < th - 1)
        tu = np.clip(tu, 0, tw - 1.001); tv = np.clip(tv, 0, th - 1.001)
        u0 = tu.astype(np.int64); v0 = tv.astype(np.int64)
        a = (tu - u0)[..., None]; b = (tv - v0)[..., None]
        t = self.tex
        img = (t[v0, u0] * (1 - a) * (1 - b) + t[v0, u0 + 1] * a * (1 - b) + t[v0 + 1, u0] * (1 - a) * b + t[v0 + 1, u0 + 1] * a * b)
        img = np.where(inside[..., None], img, 90.0)
        if self.noise > 0:   # noise is a function of the pose, so a render does not depend on call order
            import zlib
            key = zlib.crc32(np.asarray(base_pose, np.float64).tobytes())
            img = img + np.random.default_rng([self.seed, key]).normal(0, self.noise, img.shape)
        depth = np.where(inside, lam, 0.0) * 1000.0             # z along the optical axis = lam (rays have z = 1)
        return (np.clip(np.rint(img), 0, 255).astype(np.uint8), np.clip(np.rint(depth), 0, 65535).astype(np.uint16))


def base_pose(x, y, yaw_deg=0.0, z=0.0):
    q = quat_from_yaw_pitch_roll(np.deg2rad(yaw_deg))
    return (float(x), float(y), float(z), float(q[0]), float(q[1]), float(q[2]), float(q[3]))
