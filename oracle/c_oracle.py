"""
ORACLE (test infrastructure, not product code) -- ctypes binding of oracle/c/liboracle.so,
the plain-C restatement of the reference's native kernels (see oracle/c/*.c headers for the
reference file:line each function follows).  Built by `make -C oracle/c` (gcc only).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle.so")
_lib = None

_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _prep(u, P):
    u = np.ascontiguousarray(u, dtype=np.float64)
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64)[:, 0:3, :])
    assert u.ndim == 3 and u.shape[2] == 2 and P.shape == (u.shape[0], 3, 4)
    return u, P, u.shape[0], u.shape[1]


def linear_LS_triangulation(u, P, use_omp=False):
    u, P, C, N = _prep(u, P)
    x = np.empty((N, 3))
    rc = lib().orc_linear_ls(_p(u, _f64p), _p(P, _f64p), C, ctypes.c_int64(N), _p(x, _f64p), int(use_omp))
    assert rc == 0
    return x, np.ones(N, dtype=bool)


def iterative_LS_triangulation(u, P, tolerance=3.e-5, max_iter=10, use_omp=False):
    u, P, C, N = _prep(u, P)
    x = np.empty((N, 3))
    st = np.empty(N, dtype=np.int32)
    rc = lib().orc_iterative_ls(_p(u, _f64p), _p(P, _f64p), C, ctypes.c_int64(N), ctypes.c_double(tolerance),
                                int(max_iter), _p(x, _f64p), _p(st, _i32p), int(use_omp))
    assert rc == 0
    return x, st


def linear_eigen_triangulation(u, P, max_coordinate_value=1.e16, use_omp=False):
    u, P, C, N = _prep(u, P)
    x = np.empty((N, 3))
    ok = np.empty(N, dtype=np.uint8)
    rc = lib().orc_linear_eigen(_p(u, _f64p), _p(P, _f64p), C, ctypes.c_int64(N),
                                ctypes.c_double(max_coordinate_value), _p(x, _f64p), _p(ok, _u8p), int(use_omp))
    assert rc == 0
    return x, ok.astype(bool)
