"""
Drop-in counterpart of the reference module `Work/python_libs/triangulation.py`: same public
names, signatures and `(x, status)` returns, backed by the gfx950 library.

    linear_eigen_triangulation(u1, P1, u2, P2, max_coordinate_value=1.e16)   triangulation.py:6-25
    linear_LS_triangulation(u1, P1, u2, P2)                                  triangulation.py:240-244
    iterative_LS_triangulation(u1, P1, u2, P2, tolerance=3.e-5)              triangulation.py:248-252
    set_triangl_output_dtype(dtype) / output_dtype                           triangulation.py:256-267
    polynomial_triangulation                                                 out of scope (needs
        cv2.correctMatches' sextic solver) -> NotImplementedError

The reference silently falls back to pure Python when its C extension fails to build
(triangulation.py:254-256).  This module has no such fallback: without the HIP library every
call raises RuntimeError.
"""
import numpy as np

from . import triangulation_c

output_dtype = float        # triangulation.py:259


def set_triangl_output_dtype(output_dtype_):
    """Set the datatype of the triangulated 3D point positions (default "float")."""
    global output_dtype
    output_dtype = output_dtype_


def _cast(x):
    # triangulation.py:242-243 -- cast only when the float format differs
    if np.finfo(x.dtype) != np.finfo(output_dtype):
        x = x.astype(output_dtype)
    return x


def linear_LS_triangulation(*args):
    x, x_status = triangulation_c.linear_LS_triangulation(*args)
    return _cast(x), x_status


linear_LS_triangulation.__doc__ = triangulation_c.linear_LS_triangulation.__doc__


def iterative_LS_triangulation(*args, **kwargs):
    x, x_status = triangulation_c.iterative_LS_triangulation(*args, **kwargs)
    return _cast(x), x_status


iterative_LS_triangulation.__doc__ = triangulation_c.iterative_LS_triangulation.__doc__


def linear_eigen_triangulation(u1, P1, u2, P2, max_coordinate_value=1.e16):
    """
    Linear Eigenvalue based (SVD null-vector) triangulation, the arithmetic of OpenCV's
    triangulatePoints (reference: triangulation.py:6-25).  The status-vector is False for
    points with a non-finite or huge (> max_coordinate_value) coordinate.
    """
    u1 = np.asarray(u1)
    u2 = np.asarray(u2)
    if len(u1) != len(u2):
        raise ValueError("u1 and u2 must have the same number of points")
    P = np.stack([np.asarray(P1, dtype=np.float64)[0:3, 0:4], np.asarray(P2, dtype=np.float64)[0:3, 0:4]])
    u = np.stack([u1.astype(np.float64, copy=False), u2.astype(np.float64, copy=False)])
    x, status = triangulation_c.linear_eigen_triangulation_nview(u, P, max_coordinate_value)
    return x.astype(output_dtype), status


def polynomial_triangulation(u1, P1, u2, P2):
    raise NotImplementedError("polynomial_triangulation is outside the accelerated hot path "
                              "(depends on cv2.correctMatches; see DESIGN.md 'Out of scope')")


# N-view generalisations (not in the reference; C == 2 reduces to the functions above)
def linear_LS_triangulation_nview(u, P):
    x, s = triangulation_c.linear_LS_triangulation_nview(u, P)
    return _cast(x), s


def iterative_LS_triangulation_nview(u, P, tolerance=3.e-5, max_iter=10):
    x, s = triangulation_c.iterative_LS_triangulation_nview(u, P, tolerance, max_iter)
    return _cast(x), s


def linear_eigen_triangulation_nview(u, P, max_coordinate_value=1.e16):
    x, s = triangulation_c.linear_eigen_triangulation_nview(u, P, max_coordinate_value)
    return x.astype(output_dtype), s
