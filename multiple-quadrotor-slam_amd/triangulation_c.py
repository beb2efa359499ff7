"""
Thin numpy wrappers over the C ABI -- the counterpart of the reference's
`Work/python_libs/triangulation_c/__init__.py:18-86`: cast `u` to float64 iff it is not
already, force contiguity, allocate the outputs, call the native function in place, return
`(x, status)`.  The N-view forms (`*_nview`) are this build's generalisation (SURVEY.md
Appendix C); the 2-view forms keep the reference's names, argument order and dtype rules.

There is no Python fallback: a missing library raises RuntimeError (see _lib.lib()).
"""
import ctypes
import numpy as np

from . import _lib
from ._lib import c_f64p, c_i32p, c_u8p, c_i64

loaded = _lib.loaded        # same flag name as triangulation_c/__init__.py:1


def _as_f64_u(u):
    # triangulation_c/__init__.py:32-33,36-37 -- cast iff not float64, then the reshape trick
    u = np.asarray(u)
    if np.finfo(u.dtype).dtype != np.float64:
        u = u.astype(np.float64)
    if u.ndim != 2 or u.shape[1] != 2:
        raise ValueError("u must have shape (N, 2), got %r" % (u.shape,))
    return np.ascontiguousarray(u.reshape(u.size).reshape(u.shape))


def _as_u(u1, u2):
    """The reference widens non-float64 observations on the host (__init__.py:32-33).  float32 pairs -- what slam2.py passes,
    slam2.py:19 -- go over the link as they are and are widened inside the kernel: the same doubles, no host pass, half the
    bytes.  Returns (u1, u2, is_float32)."""
    a, b = np.asarray(u1), np.asarray(u2)
    if a.dtype == np.float32 and b.dtype == np.float32:
        for u in (a, b):
            if u.ndim != 2 or u.shape[1] != 2:
                raise ValueError("u must have shape (N, 2), got %r" % (u.shape,))
        return np.ascontiguousarray(a), np.ascontiguousarray(b), True
    return _as_f64_u(a), _as_f64_u(b), False


def _as_P(P):
    # the reference does NOT cast P (weave raises TypeError on a non-float64 P) and accepts
    # 3x4 or 4x4: the kernel reads the first 12 doubles only (triangulation.c:24-25).
    P = np.asarray(P)
    if P.dtype != np.float64:
        raise TypeError("camera matrix must be float64 (the reference ABI does not cast P), got %s" % P.dtype)
    if P.ndim != 2 or P.shape[1] != 4 or P.shape[0] < 3:
        raise ValueError("camera matrix must be 3x4 or 4x4, got %r" % (P.shape,))
    return np.ascontiguousarray(P.reshape(P.size).reshape(P.shape))


def _ptr(a, t):
    return a.ctypes.data_as(t)


def linear_LS_triangulation(u1, P1, u2, P2):
    """
    Linear Least Squares based triangulation (reference: triangulation_c/__init__.py:18-47).
    (u1, P1) is the reference pair of normalized image coordinates (x, y) and camera matrix,
    (u2, P2) the second pair.  The status-vector is True for all points.
    """
    (u1, u2, f32), P1, P2 = _as_u(u1, u2), _as_P(P1), _as_P(P2)
    if len(u1) != len(u2):
        raise ValueError("u1 and u2 must have the same number of points")
    x = np.empty((len(u1), 3), dtype=np.float64)
    if f32:
        _lib.check(_lib.lib().mqs_triangulation_2view_f32(
            _lib.default_context().handle, 0, _ptr(u1, _lib.c_f32p), _ptr(P1, c_f64p), _ptr(u2, _lib.c_f32p), _ptr(P2, c_f64p),
            c_i64(len(u1)), ctypes.c_double(0.0), ctypes.c_double(0.0), _ptr(x, c_f64p), None, None))
        return x, np.ones(len(u1), dtype=bool)
    _lib.check(_lib.lib().mqs_linear_LS_triangulation(
        _lib.default_context().handle, _ptr(u1, c_f64p), _ptr(P1, c_f64p), _ptr(u2, c_f64p), _ptr(P2, c_f64p),
        c_i64(len(u1)), _ptr(x, c_f64p)))
    return x, np.ones(len(u1), dtype=bool)


def iterative_LS_triangulation(u1, P1, u2, P2, tolerance=3.e-5):
    """
    Iterative (Linear) Least Squares based triangulation, Hartley & Sturm 1997
    (reference: triangulation_c/__init__.py:51-86, kernel triangulation.c:104-161).
    Returns x (N,3) float64 and x_status (N,) int32:
        1 inlier in front of both cameras; 0 not converged but in front of both;
        -1 behind 1st camera; -2 behind 2nd camera; -3 behind both.
    """
    (u1, u2, f32), P1, P2 = _as_u(u1, u2), _as_P(P1), _as_P(P2)
    if len(u1) != len(u2):
        raise ValueError("u1 and u2 must have the same number of points")
    x = np.empty((len(u1), 3), dtype=np.float64)
    x_status = np.empty(len(u1), dtype=np.int32)
    if f32:
        _lib.check(_lib.lib().mqs_triangulation_2view_f32(
            _lib.default_context().handle, 1, _ptr(u1, _lib.c_f32p), _ptr(P1, c_f64p), _ptr(u2, _lib.c_f32p), _ptr(P2, c_f64p),
            c_i64(len(u1)), ctypes.c_double(tolerance), ctypes.c_double(0.0), _ptr(x, c_f64p), _ptr(x_status, c_i32p), None))
        return x, x_status
    _lib.check(_lib.lib().mqs_iterative_LS_triangulation(
        _lib.default_context().handle, _ptr(u1, c_f64p), _ptr(P1, c_f64p), _ptr(u2, c_f64p), _ptr(P2, c_f64p),
        c_i64(len(u1)), ctypes.c_double(tolerance), _ptr(x, c_f64p), _ptr(x_status, c_i32p)))
    return x, x_status


# ---------------------------------------------------------------------------------------
# N-view forms: u (C, N, 2), P (C, >=3, 4)
# ---------------------------------------------------------------------------------------

def _pack(u, P):
    u = np.asarray(u)
    if np.finfo(u.dtype).dtype != np.float64:
        u = u.astype(np.float64)
    if u.ndim != 3 or u.shape[2] != 2:
        raise ValueError("u must have shape (C, N, 2), got %r" % (u.shape,))
    P = np.asarray(P)
    if P.dtype != np.float64:
        raise TypeError("camera matrices must be float64, got %s" % P.dtype)
    if P.ndim != 3 or P.shape[0] != u.shape[0] or P.shape[1] < 3 or P.shape[2] != 4:
        raise ValueError("P must have shape (C, 3|4, 4) matching u, got %r" % (P.shape,))
    C = u.shape[0]
    if not (2 <= C <= 8):
        raise ValueError("number of cameras must be in [2, 8], got %d" % C)
    return np.ascontiguousarray(u), np.ascontiguousarray(P[:, 0:3, :]), C, u.shape[1]


def linear_LS_triangulation_nview(u, P):
    u, P, C, N = _pack(u, P)
    x = np.empty((N, 3), dtype=np.float64)
    _lib.check(_lib.lib().mqs_triangulate_linear_ls(_lib.default_context().handle, _ptr(u, c_f64p), _ptr(P, c_f64p),
                                                    C, c_i64(N), _ptr(x, c_f64p)))
    return x, np.ones(N, dtype=bool)


def iterative_LS_triangulation_nview(u, P, tolerance=3.e-5, max_iter=10):
    u, P, C, N = _pack(u, P)
    x = np.empty((N, 3), dtype=np.float64)
    st = np.empty(N, dtype=np.int32)
    _lib.check(_lib.lib().mqs_triangulate_iterative_ls(_lib.default_context().handle, _ptr(u, c_f64p),
                                                       _ptr(P, c_f64p), C, c_i64(N), ctypes.c_double(tolerance),
                                                       int(max_iter), _ptr(x, c_f64p), _ptr(st, c_i32p)))
    return x, st


def linear_eigen_triangulation_nview(u, P, max_coordinate_value=1.e16):
    u, P, C, N = _pack(u, P)
    x = np.empty((N, 3), dtype=np.float64)
    ok = np.empty(N, dtype=np.uint8)
    _lib.check(_lib.lib().mqs_triangulate_linear_eigen(_lib.default_context().handle, _ptr(u, c_f64p),
                                                       _ptr(P, c_f64p), C, c_i64(N),
                                                       ctypes.c_double(max_coordinate_value), _ptr(x, c_f64p),
                                                       _ptr(ok, c_u8p)))
    return x, ok.astype(bool)
