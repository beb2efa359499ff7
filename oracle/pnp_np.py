"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the pose-from-points step of the
reference's per-frame loop:
    cv2.solvePnPRansac   Work/SLAM/application/own/slam2.py:453-454
    cv2.solvePnP         Work/SLAM/application/own/slam2.py:489-490, 576-577, 1156
    Rodrigues            Work/python_libs/cv2_helpers.py (thin wrapper of cv2.Rodrigues)
OpenCV 2.4 is an external dependency of the reference and is not vendored; what is restated here is its
published method (CV_ITERATIVE): Levenberg-Marquardt over (rvec, tvec) on the pixel reprojection error of
cv2.projectPoints, damping by scaling the diagonal of J^T J with (1 + lambda), lambda = 1e-3, /10 on
success, x10 on failure; without a starting pose, a direct linear transform of the undistorted points.

Deliberately independent of the device code: OpenCV's own parametrisation (Rodrigues vector, not a
left-multiplied increment), Jacobians by complex-step differentiation (exact to rounding), DLT by
least squares + SVD instead of normal equations + polar iteration.

PINNED by the reference's own run: on the recorded inlier tracks of the SVO data set the minimiser
reproduces the reference's recorded poses (traj_out.cam0-slam2.txt) to ~1e-7 on plain frames and ~1e-6 /
1e-5 (rotation / translation) on keyframes replayed through the two-pass logic of slam2.py:541-590
(tests/test_pnp.py, tests/test_replay.py).  The RANSAC draw order of OpenCV is not reproducible
(parity unpinned for the choice of hypothesis; the refined pose on a given inlier set is pinned as above).
"""
import numpy as np

from .harness_np import undistort_normalized


def rodrigues(r):
    """Rotation vector -> matrix (works for complex input: used for complex-step derivatives)."""
    r = np.asarray(r)
    th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]
    K = np.array([[0 * r[0], -r[2], r[1]], [r[2], 0 * r[0], -r[0]], [-r[1], r[0], 0 * r[0]]])
    if abs(th2) < 1e-24:
        return np.eye(3) + K
    th = np.sqrt(th2)
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th2) * (K @ K)


def rodrigues_inv(R):
    """Rotation matrix -> vector (angle in [0, pi])."""
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    if np.pi - th < 1e-6:                                   # near pi: from the symmetric part
        A = 0.5 * (R + np.eye(3))
        ax = np.sqrt(np.maximum(np.diag(A), 0))
        k = int(np.argmax(ax))
        ax = A[k] / ax[k]
        ax /= np.linalg.norm(ax)
        if ax @ v < 0:
            ax = -ax
        return th * ax
    return th / (2 * np.sin(th)) * v


def project(rvec, tvec, objp, intr):
    """cv2.projectPoints with K = [[fx,0,cx],[0,fy,cy],[0,0,1]], dist = (k1,k2,p1,p2,k3)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = intr
    q = objp @ rodrigues(rvec).T + np.asarray(tvec)
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], axis=1)


def residuals_and_jacobian(params, objp, imgp, intr):
    r = (project(params[:3], params[3:], objp, intr) - imgp).ravel()
    J = np.empty((r.size, 6))
    h = 1e-30
    for k in range(6):
        p = params.astype(complex)
        p[k] += 1j * h
        J[:, k] = (project(p[:3], p[3:], objp, intr).ravel()).imag / h
    return r, J


def dlt_pose(objp, imgp, intr):
    """Start pose without a guess (>= 6 points; >= 4 when planar): returns (R, t).  Planar point sets (OpenCV:
    third singular value of the centred covariance < 1e-3 of the second) go through the plane-to-image
    homography, the others through the direct linear transform."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = intr
    if k3 != 0:
        raise NotImplementedError("k3 in the oracle's undistortion")
    x, y = undistort_normalized((imgp[:, 0] - cx) / fx, (imgp[:, 1] - cy) / fy, k1, k2, p1, p2)
    n = len(objp)
    c = objp.mean(axis=0)
    U, S, _ = np.linalg.svd((objp - c).T @ (objp - c))
    if S[2] < 1e-3 * S[1]:
        e1, e2 = U[:, 0], U[:, 1]
        nrm = np.cross(e1, e2)
        a, b = (objp - c) @ e1, (objp - c) @ e2
        A = np.zeros((2 * n, 9))
        A[0::2, 0], A[0::2, 1], A[0::2, 2] = a, b, 1
        A[0::2, 6], A[0::2, 7], A[0::2, 8] = -x * a, -x * b, -x
        A[1::2, 3], A[1::2, 4], A[1::2, 5] = a, b, 1
        A[1::2, 6], A[1::2, 7], A[1::2, 8] = -y * a, -y * b, -y
        Hm = np.linalg.svd(A)[2][-1].reshape(3, 3)
        if Hm[2, 2] < 0:
            Hm = -Hm                                           # the plane origin (centroid) lies in front of the camera
        lam = 0.5 * (np.linalg.norm(Hm[:, 0]) + np.linalg.norm(Hm[:, 1]))
        r1, r2 = Hm[:, 0] / np.linalg.norm(Hm[:, 0]), Hm[:, 1] / np.linalg.norm(Hm[:, 1])
        M = np.stack([r1, r2, np.cross(r1, r2)], axis=1)
        Uu, _, Vt = np.linalg.svd(M)
        Rpc = Uu @ Vt
        R = Rpc @ np.stack([e1, e2, nrm])
        return R, Hm[:, 2] / lam - R @ c
    A = np.zeros((2 * n, 12))
    Xh = np.c_[objp, np.ones(n)]
    # condition the problem like any DLT: centroid / mean-distance normalisation of the world points
    s = np.mean(np.linalg.norm(objp - c, axis=1))
    T = np.eye(4)
    T[:3, :3] /= s
    T[:3, 3] = -c / s
    Xn = Xh @ T.T
    A[0::2, 0:4] = Xn; A[0::2, 8:12] = -x[:, None] * Xn
    A[1::2, 4:8] = Xn; A[1::2, 8:12] = -y[:, None] * Xn
    p = np.linalg.svd(A)[2][-1].reshape(3, 4) @ T
    if np.linalg.det(p[:, :3]) < 0:
        p = -p
    U, S, Vt = np.linalg.svd(p[:, :3])
    return U @ Vt, p[:, 3] / S.mean()


def solve_pnp(objp, imgp, intr, rvec=None, tvec=None, max_iter=100, eps=1e-14):
    """Returns (rvec, tvec, sum of squared residuals, iterations)."""
    objp = np.asarray(objp, dtype=np.float64)
    imgp = np.asarray(imgp, dtype=np.float64)
    if rvec is None:
        R, t = dlt_pose(objp, imgp, intr)
        rvec, tvec = rodrigues_inv(R), t
    p = np.concatenate([np.asarray(rvec, float).ravel(), np.asarray(tvec, float).ravel()])
    r, J = residuals_and_jacobian(p, objp, imgp, intr)
    cost = r @ r
    lam = 1e-3
    it = 0
    for it in range(1, max_iter + 1):
        H = J.T @ J
        H[np.diag_indices(6)] *= 1 + lam
        try:
            d = np.linalg.solve(H, -J.T @ r)
        except np.linalg.LinAlgError:
            lam *= 10
            continue
        rn, Jn = residuals_and_jacobian(p + d, objp, imgp, intr)
        cn = rn @ rn
        if cn < cost:
            gain = cost - cn
            p, r, J, cost = p + d, rn, Jn, cn
            lam = max(lam / 10, 1e-16)
            if np.linalg.norm(d) <= eps * (1 + np.linalg.norm(p)) or gain <= 1e-15 * cost:
                break
        else:
            lam *= 10
            if lam > 1e12:
                break
    return p[:3], p[3:], cost, it


def reprojection_errors(rvec, tvec, objp, imgp, intr):
    return np.linalg.norm(project(rvec, tvec, objp, intr) - imgp, axis=1)


def solve_pnp_ransac(objp, imgp, intr, samples, reproj_error, sample_iters=5):
    """Same hypothesis set as the device path (the caller's samples): best = most inliers, lowest index on ties."""
    best, best_n, best_mask = -1, -1, None
    for h, smp in enumerate(samples):
        try:
            rv, tv, _, _ = solve_pnp(objp[smp], imgp[smp], intr, max_iter=sample_iters)
        except np.linalg.LinAlgError:
            continue
        q = objp @ rodrigues(rv).T + tv
        m = (q[:, 2] > 0) & (reprojection_errors(rv, tv, objp, imgp, intr) <= reproj_error)
        if m.sum() > best_n:
            best, best_n, best_mask, best_pose = h, int(m.sum()), m, (rv, tv)
    rv, tv, cost, _ = solve_pnp(objp[best_mask], imgp[best_mask], intr, *best_pose)
    return rv, tv, best_mask, best
