"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's
triangulators, generalised from 2 views to C views such that C == 2
is exactly the reference.

Follows (all paths relative to /root/reference):
  * T1 linear_LS      Work/python_libs/triangulation_c/triangulation.c:24-42,65-83
                      (Python twin Work/python_libs/triangulation.py:31-94)
  * T2 iterative_LS   Work/python_libs/triangulation_c/triangulation.c:104-161
                      (Python twin triangulation.py:100-195; the C kernel's status semantics
                      are canonical: `i < 10` can be false, zero-depth break exists)
  * T3 linear_eigen   Work/python_libs/triangulation.py:6-25 -> cv2.triangulatePoints
                      (OpenCV 2.4.x, external: per point a 6x4 system with rows
                      x*P[2]-P[0], y*P[2]-P[1], x*P[1]-y*P[0] per camera, right singular
                      vector of the smallest singular value).

The third-party solver `cvSolve(..., DECOMP_SVD)` (OpenCV 2.4.x, not vendored) is restated as
its published algorithm: SVD followed by back-substitution in which singular values
w_i <= 2*DBL_EPSILON*sum(w) are dropped (minimum-norm solution).

Pinned against the reference's known-answer file test_3.mat by tests/test_oracle_golden.py.

All functions take
    u : (C, N, 2) float64  normalised image coordinates per camera
    P : (C, 3, 4) float64  camera matrices (top three rows)
"""
import numpy as np

MAX_ITER = 10            # triangulation.c:125 "Hartley suggests 10 iterations at most"
DEFAULT_TOL = 3.e-5      # triangulation_c/__init__.py:51
MAX_COORD = 1.e16        # triangulation.py:6 max_coordinate_value


def build_A_b(u, P):
    """Rows 2c,2c+1 of A (N,2C,3) and b (N,2C): triangulation.c:30-40 (construct_A/_b)."""
    u = np.asarray(u, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    C, N, _ = u.shape
    A = np.empty((N, 2 * C, 3))
    b = np.empty((N, 2 * C))
    for c in range(C):
        for k in range(2):
            A[:, 2 * c + k, :] = u[c, :, k:k + 1] * P[c, 2, 0:3] - P[c, k, 0:3]
            b[:, 2 * c + k] = -(u[c, :, k] * P[c, 2, 3] - P[c, k, 3])
    return A, b


def svd_solve(A, b):
    """
    Batched restatement of cvSolve(A, b, x, DECOMP_SVD) for tall A (N, m, 3), b (N, m):
    x = V diag(1/w if w > thr else 0) U^T b, thr = 2*eps*sum(w).
    """
    U, w, Vt = np.linalg.svd(A, full_matrices=False)
    thr = 2.0 * np.finfo(np.float64).eps * np.sum(w, axis=-1, keepdims=True)
    with np.errstate(divide="ignore"):
        winv = np.where(w > thr, 1.0 / w, 0.0)
    Utb = np.einsum("nmk,nm->nk", U, b)
    return np.einsum("nkj,nk->nj", Vt, winv * Utb)


def linear_LS_triangulation(u, P):
    """T1.  Returns x (N,3) float64, status (N,) bool all True (triangulation_c/__init__.py:47)."""
    A, b = build_A_b(u, P)
    x = svd_solve(A, b)
    return x, np.ones(len(x), dtype=bool)


def iterative_LS_triangulation(u, P, tolerance=DEFAULT_TOL, max_iter=MAX_ITER, return_iters=False):
    """
    T2.  Returns x (N,3) float64 and status (N,) int32 with the C kernel's codes generalised to
    C cameras:  s = (iters < max_iter and all d_c > 0);  for c: if d_c <= 0: s -= 2**c.
    (triangulation.c:154-159 for C == 2: {1, 0, -1, -2, -3}).
    """
    u = np.asarray(u, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    C, N, _ = u.shape
    A, b = build_A_b(u, P)
    x = np.zeros((N, 3))
    d = np.ones((N, C))                     # triangulation.c:122  d1 = d2 = 1.
    d_new = np.ones((N, C))
    it = np.zeros(N, dtype=np.int64)        # value of `i` when the loop is left
    active = np.ones(N, dtype=bool)
    for i in range(max_iter):
        idx = np.nonzero(active)[0]
        if idx.size == 0:
            break
        xa = svd_solve(A[idx], b[idx])      # triangulation.c:130
        x[idx] = xa
        dn = np.einsum("cj,nj->nc", P[:, 2, 0:3], xa) + P[:, 2, 3]   # :133-134
        d_new[idx] = dn
        conv = np.all(np.abs(dn - d[idx]) <= tolerance, axis=1) | np.any(dn == 0, axis=1)   # :137-140
        it[idx[conv]] = i
        cont = idx[~conv]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            s = 1.0 / dn[~conv]             # :143-146 rows of camera c and b scaled by 1/d_c_new
        A[cont] *= np.repeat(s, 2, axis=1)[:, :, None]
        b[cont] *= np.repeat(s, 2, axis=1)
        d[cont] = dn[~conv]                 # :149-150
        active[idx[conv]] = False
    it[active] = max_iter                   # loop ran to completion: i == 10
    front = d_new > 0
    status = ((it < max_iter) & np.all(front, axis=1)).astype(np.int32)
    for c in range(C):
        status -= ((~front[:, c]).astype(np.int32) << c)
    if return_iters:
        return x, status, it
    return x, status


def build_A_eigen(u, P):
    """(N, 3C, 4) homogeneous system of cv2.triangulatePoints (OpenCV 2.4), see module doc."""
    u = np.asarray(u, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    C, N, _ = u.shape
    A = np.empty((N, 3 * C, 4))
    for c in range(C):
        x = u[c, :, 0:1]
        y = u[c, :, 1:2]
        A[:, 3 * c + 0, :] = x * P[c, 2] - P[c, 0]
        A[:, 3 * c + 1, :] = y * P[c, 2] - P[c, 1]
        A[:, 3 * c + 2, :] = x * P[c, 1] - y * P[c, 0]
    return A


def linear_eigen_triangulation(u, P, max_coordinate_value=MAX_COORD):
    """T3.  Returns x (N,3) float64 and status (N,) bool (triangulation.py:20-25)."""
    A = build_A_eigen(u, P)
    _, _, Vt = np.linalg.svd(A, full_matrices=False)
    X = Vt[:, 3, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = X[:, 0:3] / X[:, 3:4]
        status = np.max(np.abs(x), axis=1) <= max_coordinate_value    # NaN/Inf -> False
    return x, status


# ---------------------------------------------------------------------------------------
# Per-point loop twin (mirrors the reference's Python fallback line by line in structure;
# used on small cases to cross-check the batched forms above).
# ---------------------------------------------------------------------------------------

def iterative_LS_triangulation_loop(u, P, tolerance=DEFAULT_TOL, max_iter=MAX_ITER):
    u = np.asarray(u, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    C, N, _ = u.shape
    x = np.zeros((N, 3))
    status = np.zeros(N, dtype=np.int32)
    for xi in range(N):
        A, b = build_A_b(u[:, xi:xi + 1], P)
        A = A[0]
        b = b[0]
        d = np.ones(C)
        d_new = np.ones(C)
        i = 0
        broke = False
        for i in range(max_iter):
            x[xi] = svd_solve(A[None], b[None])[0]
            d_new = P[:, 2, 0:3].dot(x[xi]) + P[:, 2, 3]
            if np.all(np.abs(d_new - d) <= tolerance) or np.any(d_new == 0):
                broke = True
                break
            for c in range(C):
                A[2 * c:2 * c + 2] *= 1.0 / d_new[c]
                b[2 * c:2 * c + 2] *= 1.0 / d_new[c]
            d = d_new
        s = int(broke and np.all(d_new > 0))
        for c in range(C):
            if d_new[c] <= 0:
                s -= (1 << c)
        status[xi] = s
    return x, status
