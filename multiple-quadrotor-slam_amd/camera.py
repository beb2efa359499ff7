"""
Camera-model steps either side of triangulation, on the gfx950 kernels of csrc/camera.hip:

  undistort_points(points_2D, cameraMatrix, distCoeffs)
      = cv2.undistortPoints(np.array([points_2D]), cameraMatrix, distCoeffs)[0]
        as called at Work/SLAM/application/own/slam2.py:551-552 and
        Work/triangulation_comparison/triangulation_comparison.py:173
  reprojection_error(objp, imgp, cameraMatrix, distCoeffs, rvec, tvec)
      = Work/python_libs/calibration_tools.py:116-124 (RMS error of one image + reprojected points)
  project_points(points, cameraMatrix, distCoeffs, P)
      world points through a 3x4 / 4x4 world->camera matrix and the distortion model.

distCoeffs: (k1, k2, p1, p2[, k3]) -- OpenCV order.  No CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import c_f64p, c_i64


def _intr(cameraMatrix, distCoeffs):
    K = np.asarray(cameraMatrix, dtype=np.float64)
    if K.shape != (3, 3):
        raise ValueError("cameraMatrix must be 3x3")
    d = np.zeros(5) if distCoeffs is None else np.asarray(distCoeffs, dtype=np.float64).reshape(-1)
    if d.size not in (4, 5):
        raise ValueError("distCoeffs must hold 4 or 5 coefficients (k1, k2, p1, p2[, k3])")
    out = np.zeros(9)
    out[0], out[1], out[2], out[3] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out[4:4 + d.size] = d
    return out


def rodrigues(rvec):
    """Rotation matrix of a rotation vector (cv2.Rodrigues(rvec)[0])."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = np.linalg.norm(r)
    K = np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K


def undistort_points(points_2D, cameraMatrix, distCoeffs):
    p = np.asarray(points_2D)
    if p.ndim != 2 or p.shape[1] != 2:
        raise ValueError("points_2D must have shape (N, 2)")
    if np.finfo(p.dtype).dtype not in (np.float32, np.float64):
        raise TypeError("points_2D must be float32 or float64")
    intr = _intr(cameraMatrix, distCoeffs)
    pin = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(pin)
    _lib.check(_lib.lib().mqs_undistort_points(_lib.default_context().handle, pin.ctypes.data_as(c_f64p),
                                               intr.ctypes.data_as(c_f64p), c_i64(len(pin)), out.ctypes.data_as(c_f64p)))
    return out.astype(p.dtype, copy=False)            # cv2 returns the input's depth


def project_points(points, cameraMatrix, distCoeffs, P, imgp=None):
    """Returns (uv (N,2), depth (N,), sum of squared errors vs imgp or None)."""
    x = np.ascontiguousarray(points, dtype=np.float64)
    if x.ndim != 2 or x.shape[1] != 3:
        raise ValueError("points must have shape (N, 3)")
    Pm = np.ascontiguousarray(np.asarray(P, dtype=np.float64)[0:3, 0:4])
    intr = _intr(cameraMatrix, distCoeffs)
    uv = np.empty((len(x), 2))
    z = np.empty(len(x))
    m = None if imgp is None else np.ascontiguousarray(imgp, dtype=np.float64)
    if m is not None and m.shape != uv.shape:
        raise ValueError("imgp must have shape (N, 2)")
    err = np.zeros(1)
    _lib.check(_lib.lib().mqs_project_points(
        _lib.default_context().handle, x.ctypes.data_as(c_f64p), Pm.ctypes.data_as(c_f64p), intr.ctypes.data_as(c_f64p),
        None if m is None else m.ctypes.data_as(c_f64p), c_i64(len(x)), uv.ctypes.data_as(c_f64p),
        z.ctypes.data_as(c_f64p), None if m is None else err.ctypes.data_as(c_f64p)))
    return uv, z, (None if m is None else float(err[0]))


def reprojection_error(objp, imgp, cameraMatrix, distCoeffs, rvec, tvec):
    """RMS reprojection error of one image and the reprojected points (calibration_tools.py:116-124)."""
    P = np.concatenate([rodrigues(rvec), np.asarray(tvec, dtype=np.float64).reshape(3, 1)], axis=1)
    imgp = np.asarray(imgp).reshape(-1, 2)
    uv, _, sq = project_points(np.asarray(objp).reshape(-1, 3), cameraMatrix, distCoeffs, P, imgp)
    return np.sqrt(sq / float(len(imgp))), uv
