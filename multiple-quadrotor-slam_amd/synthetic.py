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
