"""
Scaled synthetic generator for the benchmark configurations.

Keeps the geometry and noise model of the reference's harness
(`Work/triangulation_comparison/triangulation_comparison.py`: points in a radius-4 ball :30-33,
640x480 cameras with f = 480, c = (320, 240) :97-98, camera centre (sideways, 0, -40 + towards)
with R = Rot_y(angle) :109-123, N(0, 0.8 px) noise followed by rint :155-160,277-278, normalised
as (p - c) / f :168-172) but makes the number of landmarks a parameter -- the reference's
generator is fixed at 257 grid points -- and adds cameras 3 and 4 for the 4-view configuration.
"""
from math import asin
import numpy as np

RSEED = 123456789
POSE_OFFSET = 40.0
RESOLUTION = (640, 480)
FOCAL = float(min(RESOLUTION))
CENTRE = np.array(RESOLUTION, dtype=np.float64) / 2.0
NOISE_SIGMA = 0.8


def camera_matrix(sideways=0.0, towards=0.0, angle=0.0, offset=POSE_OFFSET):
    """3x4 world->camera matrix [R | -R C] of a camera at (sideways, 0, -offset + towards)."""
    ca, sa = np.cos(angle), np.sin(angle)
    R = np.array([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])
    C = np.array([sideways, 0.0, -offset + towards])
    return np.concatenate([R, (-R.dot(C)).reshape(3, 1)], axis=1)


def benchmark_cameras(C):
    """
    C = 2: reference camera + last pose of trajectory 4 (circle, sideways 12).
    C = 4: + last pose of trajectory 1 (sideways 12) + last pose of trajectory 5 (90 degrees),
    which gives unequal depths so that iterative-LS really iterates.  C in (3, 5..8) fills in
    further circle poses.
    """
    a4 = asin(12.0 / POSE_OFFSET)
    poses = [
        (0.0, 0.0, 0.0),
        (POSE_OFFSET * np.sin(a4), POSE_OFFSET * (1 - np.cos(a4)), a4),
        (12.0, 0.0, 0.0),
        (POSE_OFFSET, POSE_OFFSET, np.pi / 2),
    ]
    for k in range(4, 8):
        a = -(k - 3) * 0.2
        poses.append((POSE_OFFSET * np.sin(a), POSE_OFFSET * (1 - np.cos(a)), a))
    if not (2 <= C <= 8):
        raise ValueError("C must be in [2, 8]")
    return np.stack([camera_matrix(*p) for p in poses[:C]])


def ball_points(N, radius=4.0, seed=RSEED):
    """N i.i.d. points uniform in the ball of the given radius (PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = rng.standard_normal((N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = radius * rng.random(N) ** (1.0 / 3.0)
    return d * r[:, None]


def project_pixels(points, P):
    """Pinhole projection f*(X/Z, Y/Z) + c of world points through the 3x4 matrix P."""
    q = points.dot(P[:, 0:3].T) + P[:, 3]
    return FOCAL * q[:, 0:2] / q[:, 2:3] + CENTRE


def make_observations(points, P, sigma=NOISE_SIGMA, discretized=True, seed=RSEED):
    """
    Normalised noisy observations u (C, N, 2) float64 of `points` (N, 3) in cameras P (C, 3, 4);
    camera c's noise stream is PCG64(seed + c).
    """
    C = P.shape[0]
    u = np.empty((C, len(points), 2), dtype=np.float64)
    for c in range(C):
        px = project_pixels(points, P[c])
        if sigma:
            rng = np.random.Generator(np.random.PCG64(seed + c))
            px = px + sigma * rng.standard_normal(px.shape)
        if discretized:
            px = np.rint(px)
        u[c] = (px - CENTRE) / FOCAL
    return u


def triangulation_problem(N, C, sigma=NOISE_SIGMA, discretized=True, seed=RSEED):
    """Returns (u (C,N,2), P (C,3,4), points_true (N,3)) of the benchmark scene."""
    P = benchmark_cameras(C)
    pts = ball_points(N, seed=seed)
    return make_observations(pts, P, sigma, discretized, seed), P, pts


# ---------------------------------------------------------------------------------------------------
# Rendered image sequence for the end-to-end loop (BASELINE configs[4]; the reference's Blender-rendered set is
# not in the repository): a textured plane seen by a moving pinhole camera with OpenCV-model distortion.
# ---------------------------------------------------------------------------------------------------
def plane_texture(size=2048, blobs=6000, seed=4):
    """Sum of Gaussian blobs of mixed scale on a size x size grid, float32 in [0, 255]."""
    rng = np.random.default_rng(seed)
    tex = np.zeros((size, size), dtype=np.float32)
    for _ in range(blobs):
        cx, cy = rng.uniform(0, size, 2)
        s = rng.choice([2.0, 3.5, 6.0, 10.0])
        a = rng.uniform(-1, 1)
        r = int(4 * s)
        x0, x1, y0, y1 = int(max(0, cx - r)), int(min(size, cx + r + 1)), int(max(0, cy - r)), int(min(size, cy + r + 1))
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        tex[y0:y1, x0:x1] += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    tex -= tex.min()
    return tex * (255.0 / tex.max())


class PlaneSequence:
    """World plane z = 0 textured over [-extent, extent]^2; camera poses are world -> camera (R, t)."""

    def __init__(self, image_size=(640, 480), f=480.0, dist=(-0.06, 0.01, 0.0005, -0.0003), extent=10.0, frames=40, seed=4):
        W, H = image_size
        self.W, self.H, self.extent = W, H, extent
        self.K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
        self.dist = np.array(dist, dtype=np.float64)
        self.tex = plane_texture(seed=seed)
        # per-pixel undistorted normalised ray directions (fixed-point inverse of the distortion model)
        v, u = np.mgrid[0:H, 0:W].astype(np.float64)
        xd, yd = (u - W / 2.0) / f, (v - H / 2.0) / f
        k1, k2, p1, p2 = self.dist
        x, y = xd.copy(), yd.copy()
        for _ in range(20):
            r2 = x * x + y * y
            g = 1 + k1 * r2 + k2 * r2 * r2
            x = (xd - (2 * p1 * x * y + p2 * (r2 + 2 * x * x))) / g
            y = (yd - (p1 * (r2 + 2 * y * y) + 2 * p2 * x * y)) / g
        self.rays = np.stack([x, y, np.ones_like(x)], axis=-1)
        # trajectory: sideways sweep with a slow yaw and a tilt so that the plane is not fronto-parallel
        self.poses = []
        for k in range(frames):
            s = k / max(1, frames - 1)
            C = np.array([-3.0 + 6.0 * s, 0.8 - 0.6 * s, -9.0 + 0.8 * np.sin(2.5 * s)])
            yaw, pitch = 0.25 - 0.5 * s, 0.18
            Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
            Rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
            R = Rx @ Ry
            self.poses.append((R, -R @ C))

    def centres(self):
        return np.array([-R.T @ t for R, t in self.poses])

    def render(self, k):
        R, t = self.poses[k]
        C = -R.T @ t
        d = self.rays @ R                                      # ray directions in the world frame (R^T applied to each ray)
        lam = -C[2] / d[..., 2]
        X = C[0] + lam * d[..., 0]
        Y = C[1] + lam * d[..., 1]
        n = self.tex.shape[0]
        tx = (X + self.extent) * ((n - 1) / (2 * self.extent))
        ty = (Y + self.extent) * ((n - 1) / (2 * self.extent))
        inside = (lam > 0) & (tx >= 0) & (tx < n - 1) & (ty >= 0) & (ty < n - 1)
        tx, ty = np.clip(tx, 0, n - 1.001), np.clip(ty, 0, n - 1.001)
        ix, iy = tx.astype(np.int64), ty.astype(np.int64)
        fx, fy = (tx - ix).astype(np.float32), (ty - iy).astype(np.float32)
        T = self.tex
        val = (T[iy, ix] * (1 - fx) * (1 - fy) + T[iy, ix + 1] * fx * (1 - fy) + T[iy + 1, ix] * (1 - fx) * fy
               + T[iy + 1, ix + 1] * fx * fy)
        return np.where(inside, np.clip(np.rint(val), 0, 255), 0).astype(np.uint8)

    def project(self, k, pts3d):
        """Exact pixel projections of world points in frame k (pinhole + distortion)."""
        R, t = self.poses[k]
        q = np.asarray(pts3d, dtype=np.float64) @ R.T + t
        x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
        k1, k2, p1, p2 = self.dist
        r2 = x * x + y * y
        g = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        f = self.K[0, 0]
        return np.stack([f * xd + self.W / 2.0, f * yd + self.H / 2.0], axis=1)
