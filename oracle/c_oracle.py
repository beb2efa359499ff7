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


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    cdir = os.path.join(_HERE, "c")
    return any(os.path.getmtime(os.path.join(cdir, f)) > t for f in os.listdir(cdir) if f.endswith((".c", ".h")) or f == "Makefile")


def build():
    """Builds liboracle.so when a source is newer than it.  No child process at all when it is up to date; when one is
    needed it runs without a profiler preload (LD_PRELOAD / ROCP_*): under `rocprofv3 --pmc` every child would otherwise
    initialise the GPU before exec'ing gcc, which the GPU pool forbids."""
    if not _stale():
        return
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_"))}
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")], env=env)


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


def _opt(a, t):
    return None if a is None else _p(a, t)


def ba_linearize(poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None, lam=0.0, use_omp=False):
    """C restatement of oracle/ba_np.linearize: returns S, g, cost, nvalid."""
    C, N = obs.shape[0], obs.shape[1]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (poses, calib, sigma, points, obs)]
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    pw = None if prior_w is None else np.ascontiguousarray(prior_w, dtype=np.float64)
    px = None if prior_xyz is None else np.ascontiguousarray(prior_xyz, dtype=np.float64)
    n6 = 6 * C
    out = np.zeros(n6 * n6 + n6 + 2)
    rc = lib().orc_ba_linearize(_p(arrs[0], _f64p), _p(arrs[1], _f64p), _p(arrs[2], _f64p), C, _p(arrs[3], _f64p),
                                _p(arrs[4], _f64p), _opt(m, _u8p), _opt(pw, _f64p), _opt(px, _f64p), ctypes.c_int64(N),
                                ctypes.c_double(lam), _p(out, _f64p), int(use_omp))
    assert rc == 0
    return out[:n6 * n6].reshape(n6, n6), out[n6 * n6:n6 * n6 + n6], out[-2], int(out[-1])


def ba_backsub(poses, calib, sigma, points, obs, dpose, mask=None, prior_w=None, prior_xyz=None, lam=0.0, use_omp=False):
    C, N = obs.shape[0], obs.shape[1]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (poses, calib, sigma, points, obs, dpose)]
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    pw = None if prior_w is None else np.ascontiguousarray(prior_w, dtype=np.float64)
    px = None if prior_xyz is None else np.ascontiguousarray(prior_xyz, dtype=np.float64)
    out = np.empty((N, 3))
    rc = lib().orc_ba_backsub(_p(arrs[0], _f64p), _p(arrs[1], _f64p), _p(arrs[2], _f64p), C, _p(arrs[3], _f64p),
                              _p(arrs[4], _f64p), _opt(m, _u8p), _opt(pw, _f64p), _opt(px, _f64p), ctypes.c_int64(N),
                              ctypes.c_double(lam), _p(arrs[5], _f64p), _p(out, _f64p), int(use_omp))
    assert rc == 0
    return out
