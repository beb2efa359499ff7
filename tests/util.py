"""Helpers shared by the parity tests."""
import numpy as np


def rel_err(x, x_ref):
    """Per-point |x - x_ref| / max(|x_ref|, 1)  (SURVEY.md 8(d) parity gate)."""
    return np.linalg.norm(x - x_ref, axis=1) / np.maximum(np.linalg.norm(x_ref, axis=1), 1.0)


def stable_mask(fn, u, P, x_ref, s_ref=None, eps=1e-13, tol=1e-7, seed=0):
    """
    Points whose ORACLE answer is itself stable: re-run the oracle on inputs perturbed by a
    relative 1e-13 and keep the points whose output moves by < 1e-7 relative and whose status
    does not change.  The parity bar (1e-5) is asserted on these; the rest are threshold- or
    rank-adjacent (SURVEY.md section 7 'Status-flag parity' / 'Rank-deficient systems') and
    only their count is bounded.
    """
    rng = np.random.default_rng(seed)
    up = u * (1.0 + eps * rng.standard_normal(u.shape)) + 1e-16 * rng.standard_normal(u.shape)
    xp, sp = fn(up, P)
    with np.errstate(invalid="ignore"):
        ok = rel_err(xp, x_ref) < tol
    ok &= np.isfinite(x_ref).all(axis=1)
    if s_ref is not None:
        ok &= (np.asarray(sp) == np.asarray(s_ref))
    return ok


def random_scene(N, C, seed=0, behind_frac=0.05, noise=1e-3):
    """Random well-posed C-view scene with a few points behind some camera."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-3, 3, (N, 3))
    P = np.empty((C, 3, 4))
    for c in range(C):
        ang = rng.uniform(-0.5, 0.5, 3)
        cx, sx = np.cos(ang[0]), np.sin(ang[0])
        cy, sy = np.cos(ang[1]), np.sin(ang[1])
        cz, sz = np.cos(ang[2]), np.sin(ang[2])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = Rz @ Ry @ Rx
        centre = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-25, -12)])
        P[c] = np.concatenate([R, (-R @ centre).reshape(3, 1)], axis=1)
    nb = int(behind_frac * N)
    if nb:
        pts[:nb, 2] -= 40.0                     # far behind the cameras
    u = np.empty((C, N, 2))
    for c in range(C):
        q = pts @ P[c, :, :3].T + P[c, :, 3]
        u[c] = q[:, :2] / q[:, 2:3] + noise * rng.standard_normal((N, 2))
    return u, P, pts


def ate_rmse(traj_est, traj_gt, max_dt=0.02):
    """Absolute trajectory error as the TUM benchmark's evaluate_ate.py computes it (restated:
    nearest-timestamp association within max_dt, rigid Horn alignment without scale, RMSE of the
    translational residuals).  traj_*: lists of (timestamp, xyz)."""
    gt_t = np.array([t for t, _ in traj_gt])
    pairs = []
    for t, p in traj_est:
        k = int(np.argmin(np.abs(gt_t - t)))
        if abs(gt_t[k] - t) < max_dt:
            pairs.append((p, traj_gt[k][1]))
    A = np.array([a for a, _ in pairs]).T        # estimated, 3 x n
    B = np.array([b for _, b in pairs]).T        # ground truth
    Ac, Bc = A - A.mean(1, keepdims=True), B - B.mean(1, keepdims=True)
    U, _, Vt = np.linalg.svd(Ac @ Bc.T)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt                               # rotates ground-truth-centred onto estimate: est ~ R^T ...
    R = R.T
    t = B.mean(1, keepdims=True) - R @ A.mean(1, keepdims=True)
    err = R @ A + t - B
    return float(np.sqrt(np.mean(np.sum(err ** 2, axis=0)))), len(pairs)
