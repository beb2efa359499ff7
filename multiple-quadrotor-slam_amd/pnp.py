"""
Camera pose from 3D-2D correspondences on the gfx950 kernels of csrc/pnp.hip -- the pose step of the
reference's per-frame loop, with OpenCV's call surface:

  solvePnP(objp, imgp, cameraMatrix, distCoeffs[, rvec, tvec, useExtrinsicGuess])  -> (ret, rvec, tvec)
      = cv2.solvePnP as called at Work/SLAM/application/own/slam2.py:489-490, 576-577, 1156
  solvePnPRansac(objp, imgp, cameraMatrix, distCoeffs, minInliersCount=.., reprojectionError=..)
      -> (rvec, tvec, inliers)     = cv2.solvePnPRansac as called at slam2.py:453-454
  Rodrigues(rvec_or_R)             = Work/python_libs/cv2_helpers.py:30-31

rvec / tvec are (3, 1) float64 like OpenCV's; `inliers` is an (n, 1) int32 array of indices or None when
there are none (slam2.py:457 tests `inliers == None`).  No CPU fallback: the library must be loadable.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import c_f64p, c_i32p, c_i64, c_u8p
from .camera import _intr, rodrigues

DEFAULT_MAX_ITER = 100          # LM iterations: run to convergence (OpenCV 2.4 stops at 20 or on a relative step of FLT_EPSILON)
DEFAULT_EPS = 1e-10            # step size (relative to 1 + |t|) below which LM stops; OpenCV 2.4: FLT_EPSILON relative step
RANSAC_HYPOTHESES = 256         # all evaluated in one launch
RANSAC_SAMPLE_SIZE = 6          # direct linear transform needs 6 points (OpenCV 2.4 draws 5 and calls solvePnP on them)
RANSAC_SAMPLE_ITERS = 5


def rotation_vector(R):
    """Rotation matrix -> rotation vector (cv2.Rodrigues(R)[0])."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    if np.pi - th < 1e-6:
        A = 0.5 * (R + np.eye(3))
        d = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(d))
        ax = A[k] / d[k]
        ax /= np.linalg.norm(ax)
        if ax @ v < 0:
            ax = -ax
        return th * ax
    return th / (2.0 * np.sin(th)) * v


def Rodrigues(rvec_or_R):
    a = np.asarray(rvec_or_R, dtype=np.float64)
    if a.shape == (3, 3):
        return rotation_vector(a).reshape(3, 1)
    if a.size != 3:
        raise ValueError("Rodrigues expects a 3-vector or a 3x3 matrix")
    return rodrigues(a)


def _points(objp, imgp):
    o = np.asarray(objp)
    m = np.asarray(imgp)
    if o.ndim != 2 or o.shape[1] != 3:
        raise ValueError("objp must have shape (N, 3)")
    if m.ndim != 2 or m.shape[1] != 2 or len(m) != len(o):
        raise ValueError("imgp must have shape (N, 2) with the same N as objp")
    return np.ascontiguousarray(o, dtype=np.float64), np.ascontiguousarray(m, dtype=np.float64)


def _pose12(rvec, tvec):
    R = rodrigues(np.asarray(rvec, dtype=np.float64).reshape(3))
    return np.ascontiguousarray(np.concatenate([R, np.asarray(tvec, dtype=np.float64).reshape(3, 1)], axis=1))


def _rt(pose):
    return rotation_vector(pose[:, :3]).reshape(3, 1), pose[:, 3].reshape(3, 1).copy()


def solve_pnp_pose(objp, imgp, intr, pose=None, max_iter=DEFAULT_MAX_ITER, eps=DEFAULT_EPS):
    """Array-level form: returns (P (3, 4) = [R | t], info = [sqerr, iterations, N, flags])."""
    o, m = _points(objp, imgp)
    use_guess = pose is not None
    P = np.ascontiguousarray(pose, dtype=np.float64).reshape(3, 4).copy() if use_guess else np.eye(3, 4)
    if len(o) < (3 if use_guess else 6):
        raise ValueError("solvePnP needs at least 6 points (3 with a starting pose)")
    info = np.zeros(4)
    intr = np.ascontiguousarray(intr, dtype=np.float64)
    _lib.check(_lib.lib().mqs_solve_pnp(_lib.default_context().handle, o.ctypes.data_as(c_f64p), m.ctypes.data_as(c_f64p),
                                        c_i64(len(o)), intr.ctypes.data_as(c_f64p), P.ctypes.data_as(c_f64p),
                                        int(use_guess), int(max_iter), ctypes.c_double(eps), info.ctypes.data_as(c_f64p)))
    return P, info


KF_DROPPED = -128           # keyframe_step status of a point the first triangulation pass did not keep


def keyframe_step(objp, imgp, intr, P_prev, p0=None, p1=None, P0=None, tolerance=3.e-5, max_iter=DEFAULT_MAX_ITER,
                  eps=DEFAULT_EPS):
    """
    One keyframe step of the per-frame loop in ONE library call / launch (`mqs_keyframe_step`; slam2.py handle_new_frame
    :453-490, 541-590): pose from the tracked landmarks (objp (n,3), imgp (n,2) pixels, start P_prev (3,4)), triangulation of
    the new pixel tracks p0 (base keyframe, pose P0) / p1 (this frame) -- undistorted as cv2.undistortPoints does --, refined
    pose on old + kept (status 1, float32) points, re-triangulation with it.  Without new tracks: the pose alone.
    Returns (P_first (3,4), P_refined (3,4), x (m,3), status (m,) int32: second-pass status or KF_DROPPED, info (8,)).
    """
    o, m = _points(objp, imgp)
    if len(o) < 3:
        raise ValueError("a pose needs at least 3 tracked landmarks")
    intr = np.ascontiguousarray(intr, dtype=np.float64)
    Pp = np.ascontiguousarray(P_prev, dtype=np.float64).reshape(3, 4)
    n_new = 0 if p0 is None else len(p0)
    if n_new:
        a = np.ascontiguousarray(p0, dtype=np.float64).reshape(-1, 2)
        b = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 2)
        if len(a) != len(b) or P0 is None:
            raise ValueError("p0 and p1 must have the same length and P0 must be given")
        Pb = np.ascontiguousarray(P0, dtype=np.float64).reshape(3, 4)
    else:
        a = b = np.zeros((0, 2))
        Pb = None
    poses = np.empty((2, 3, 4))
    x = np.empty((n_new, 3))
    st = np.empty(n_new, dtype=np.int32)
    info = np.zeros(8)
    fp = lambda arr: arr.ctypes.data_as(c_f64p) if arr is not None and arr.size else None
    _lib.check(_lib.lib().mqs_keyframe_step(
        _lib.default_context().handle, fp(o), fp(m), c_i64(len(o)), fp(a), fp(b), c_i64(n_new), fp(intr), fp(Pp), fp(Pb),
        ctypes.c_double(tolerance), int(max_iter), ctypes.c_double(eps), poses.ctypes.data_as(c_f64p), fp(x),
        st.ctypes.data_as(_lib.c_i32p) if n_new else None, info.ctypes.data_as(c_f64p)))
    return poses[0], poses[1], x, st, info


class KeyframeStepper:
    """`keyframe_step` for a loop: the intrinsics, the output buffers and the ctypes plumbing are set up once, so that a
    frame costs the library call and little else.  Inputs must be float64, C-contiguous (they are passed as they are)."""

    def __init__(self, intr, capacity=1024, tolerance=3.e-5, max_iter=DEFAULT_MAX_ITER, eps=DEFAULT_EPS):
        self._fn = _lib.lib().mqs_keyframe_step
        self._ctx = _lib.default_context().handle
        self._intr = np.ascontiguousarray(intr, dtype=np.float64)
        self._pintr = self._intr.ctypes.data_as(c_f64p)
        self._poses = np.empty((2, 3, 4))
        self._pposes = self._poses.ctypes.data_as(c_f64p)
        self._tol, self._it, self._eps = ctypes.c_double(tolerance), int(max_iter), ctypes.c_double(eps)
        self._grow(capacity)

    def _grow(self, n):
        self._cap = n
        self._x = np.empty((n, 3))
        self._st = np.empty(n, dtype=np.int32)
        self._px, self._pst = self._x.ctypes.data_as(c_f64p), self._st.ctypes.data_as(_lib.c_i32p)

    def __call__(self, objp, imgp, P_prev, p0=None, p1=None, P0=None):
        """Returns (refined pose (3,4), x (m,3), status (m,)); x / status are views of buffers reused by the next call."""
        n_new = 0 if p0 is None else len(p0)
        if n_new > self._cap:
            self._grow(2 * n_new)
        rc = self._fn(self._ctx, objp.ctypes.data_as(c_f64p), imgp.ctypes.data_as(c_f64p), len(objp),
                      p0.ctypes.data_as(c_f64p) if n_new else None, p1.ctypes.data_as(c_f64p) if n_new else None, n_new,
                      self._pintr, P_prev.ctypes.data_as(c_f64p), P0.ctypes.data_as(c_f64p) if n_new else None,
                      self._tol, self._it, self._eps, self._pposes, self._px if n_new else None,
                      self._pst if n_new else None, None)
        if rc != 0:
            _lib.check(rc)
        return self._poses[1].copy(), self._x[:n_new], self._st[:n_new]


def solvePnP(objp, imgp, cameraMatrix, distCoeffs, rvec=None, tvec=None, useExtrinsicGuess=False,
             max_iter=DEFAULT_MAX_ITER, eps=DEFAULT_EPS):
    intr = _intr(cameraMatrix, distCoeffs)
    pose = None
    if useExtrinsicGuess:
        if rvec is None or tvec is None:
            raise ValueError("useExtrinsicGuess needs rvec and tvec")
        pose = _pose12(rvec, tvec)
    P, info = solve_pnp_pose(objp, imgp, intr, pose, max_iter, eps)
    r, t = _rt(P)
    return bool(np.isfinite(P).all()), r, t


def draw_samples(n_points, hypotheses=RANSAC_HYPOTHESES, sample_size=RANSAC_SAMPLE_SIZE, seed=0):
    """Minimal samples without replacement, [hypotheses][sample_size] int32 (numpy PCG64, repeatable)."""
    rng = np.random.default_rng(seed)
    if n_points < 8 * sample_size:
        # few points: the head of a random permutation per hypothesis
        keys = rng.random((hypotheses, n_points))
        return np.ascontiguousarray(np.argsort(keys, axis=1)[:, :sample_size].astype(np.int32))
    # many points: independent draws, rows with a repeated index drawn again (a permutation per hypothesis costs a
    # millisecond of host time per frame at 300 points -- more than the GPU spends on the whole RANSAC)
    smp = rng.integers(0, n_points, (hypotheses, sample_size))
    for _ in range(64):
        srt = np.sort(smp, axis=1)
        bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
        if len(bad) == 0:
            break
        smp[bad] = rng.integers(0, n_points, (len(bad), sample_size))
    return np.ascontiguousarray(smp.astype(np.int32))


def solve_pnp_ransac_pose(objp, imgp, intr, reproj_error, samples=None, hypotheses=RANSAC_HYPOTHESES, seed=0,
                          sample_iters=RANSAC_SAMPLE_ITERS, max_iter=DEFAULT_MAX_ITER, eps=DEFAULT_EPS):
    """Array-level form: returns (P (3, 4), mask (N,) bool, chosen hypothesis, info)."""
    o, m = _points(objp, imgp)
    n = len(o)
    if samples is None:
        if n < RANSAC_SAMPLE_SIZE:
            raise ValueError("solvePnPRansac needs at least %d points" % RANSAC_SAMPLE_SIZE)
        samples = draw_samples(n, hypotheses, RANSAC_SAMPLE_SIZE, seed)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    if samples.ndim != 2 or samples.shape[1] < 6 or samples.min() < 0 or samples.max() >= n:
        raise ValueError("samples must be [hypotheses][>= 6] indices into the points")
    P = np.zeros((3, 4))
    sel = np.zeros(2, dtype=np.int32)
    mask = np.zeros(n, dtype=np.uint8)
    info = np.zeros(4)
    intr = np.ascontiguousarray(intr, dtype=np.float64)
    _lib.check(_lib.lib().mqs_solve_pnp_ransac(
        _lib.default_context().handle, o.ctypes.data_as(c_f64p), m.ctypes.data_as(c_f64p), c_i64(n),
        intr.ctypes.data_as(c_f64p), samples.ctypes.data_as(c_i32p), int(samples.shape[0]), int(samples.shape[1]),
        ctypes.c_double(reproj_error), int(sample_iters), int(max_iter), ctypes.c_double(eps), P.ctypes.data_as(c_f64p),
        sel.ctypes.data_as(c_i32p), mask.ctypes.data_as(c_u8p), info.ctypes.data_as(c_f64p)))
    return P, mask.astype(bool), int(sel[0]), info


def solvePnPRansac(objp, imgp, cameraMatrix, distCoeffs, minInliersCount=100, reprojectionError=8.0,
                   iterationsCount=RANSAC_HYPOTHESES, seed=0):
    """minInliersCount only ends OpenCV's serial loop early; every hypothesis is evaluated here, so it has no effect."""
    del minInliersCount
    P, mask, best, _ = solve_pnp_ransac_pose(objp, imgp, _intr(cameraMatrix, distCoeffs), float(reprojectionError),
                                             hypotheses=int(iterationsCount), seed=seed)
    r, t = _rt(P)
    idx = np.nonzero(mask)[0].astype(np.int32)
    return r, t, (idx.reshape(-1, 1) if best >= 0 and idx.size else None)
