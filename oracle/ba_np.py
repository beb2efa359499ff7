"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the bundle-adjustment
arithmetic that the reference delegates to GTSAM 3.2.1 (NOT vendored under /root/reference,
cannot be built here): parity of per-iteration normal equations is therefore UNPINNED.  The CONVERGED
output is pinned: with the odometry BetweenFactors (:301-309, the tool's default) the optimiser built on
this arithmetic lands on the reference's committed GTSAM result under
Work/SLAM/tools/bundle_adjustment/example (map 2e-4, trajectory 3e-4 = the files' 6 digits;
tests/test_ba_files.py::test_odometry_between_factors), and on the SVO data set it reproduces the
reference's post-BA accuracy (ATE 0.0214 m vs 0.021598 m).

Graph structure followed: /root/reference/Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp
  :268-282  PriorFactor<Pose3> on first-frame poses, PriorFactor<Point3> on step-0 landmarks
  :289-298  GenericProjectionFactor<Pose3, Point3, Cal3DS2>(uv, sigma_c, pose, point, K_c)
  :301-309  BetweenFactor<Pose3>(from, to, odometry, sigma[from cam][to cam])   (between_error below)
  :323-324  LevenbergMarquardtOptimizer(graph, initialEstimate).optimize()
Conventions (IO.hpp:221-236): pose = camera-to-world (R, t), line "tx ty tz qx qy qz qw";
calibration = fx fy s u0 v0 k1 k2 p1 p2 (Cal3DS2 constructor order).

Published GTSAM 3.2.1 factor maths restated:
  q = R^T (p - t);  (x, y) = (X/Z, Y/Z);  r2 = x^2 + y^2;  g = 1 + k1 r2 + k2 r2^2
  x' = g x + 2 p1 x y + p2 (r2 + 2 x^2);   y' = g y + 2 p2 x y + p1 (r2 + 2 y^2)
  uv_hat = (fx x' + s y' + u0, fy y' + v0);  e = uv_hat - uv;  whitened e / sigma
  pose tangent [omega, v], right perturbation T*Exp(xi):  dq/dxi = [ [q]x | -I ],  dq/dp = R^T
  cheirality (Z <= 0): zero Jacobians, constant residual 2*fx*(1,1)  (GenericProjectionFactor
  with throwCheirality = false).
  cost = 0.5 * sum |e / sigma|^2   (= graph.error(values))

Dense layouts shared with the library (include/mqslam.h):
  poses [C][12] = R row-major (9) + t (3);  calib [C][9];  sigma [C];  points [N][3];
  obs [C][N][2];  mask [C][N] uint8 or None;  prior_w [N] (1/sigma_p^2, 0 = no prior) and
  prior_xyz [N][3] or None.
"""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def so3_exp(w):
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K


def so3_log(R):
    c = np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    return v * (th / (2 * np.sin(th)))


def retract_pose(pose12, xi):
    """R <- R Exp(omega), t <- t + R v   (first-order-equivalent retraction, Appendix B)."""
    R = pose12[:9].reshape(3, 3)
    t = pose12[9:]
    out = np.empty(12)
    out[:9] = (R @ so3_exp(xi[:3])).reshape(-1)
    out[9:] = t + R @ xi[3:]
    return out


def project(pose12, K9, p):
    """Returns (uv_hat (2,), D (2,3) = d uv_hat / d q, q (3,), valid)."""
    R = pose12[:9].reshape(3, 3)
    t = pose12[9:]
    fx, fy, s, u0, v0, k1, k2, p1, p2 = K9
    q = R.T @ (p - t)
    X, Y, Z = q
    if not (Z > 0):
        return None, None, q, False
    x, y = X / Z, Y / Z
    r2 = x * x + y * y
    g = 1 + k1 * r2 + k2 * r2 * r2
    xd = g * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = g * y + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    uv = np.array([fx * xd + s * yd + u0, fy * yd + v0])
    dg = k1 + 2 * k2 * r2
    Dd = np.array([[g + 2 * x * x * dg + 2 * p1 * y + 6 * p2 * x, 2 * x * y * dg + 2 * p1 * x + 2 * p2 * y],
                   [2 * x * y * dg + 2 * p2 * y + 2 * p1 * x, g + 2 * y * y * dg + 2 * p2 * x + 6 * p1 * y]])
    Kk = np.array([[fx, s], [0.0, fy]])
    Dp = np.array([[1.0, 0.0, -x], [0.0, 1.0, -y]]) / Z
    return uv, Kk @ Dd @ Dp, q, True


def factor(pose12, K9, sigma, p, uv):
    """Whitened residual e (2,), Jpose (2,6), Jpoint (2,3), valid."""
    uvh, D, q, valid = project(pose12, K9, p)
    if not valid:
        return np.full(2, 2.0 * K9[0]) / sigma, np.zeros((2, 6)), np.zeros((2, 3)), False
    R = pose12[:9].reshape(3, 3)
    e = (uvh - uv) / sigma
    Jp = (D @ np.concatenate([skew(q), -np.eye(3)], axis=1)) / sigma
    Jl = (D @ R.T) / sigma
    return e, Jp, Jl, True


def linearize(poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None, lam=0.0):
    """
    Landmark-eliminated (Schur) normal equations of the projection factors (+ point priors).
    Returns S (6C,6C), g (6C,) with S dpose = g, cost, n_valid, and the per-landmark pieces
    (Hll_inv (N,3,3), gl (N,3), Hpl (N,6C,3)) for the back-substitution.
    """
    C, N = obs.shape[0], obs.shape[1]
    n6 = 6 * C
    S = np.zeros((n6, n6))
    g = np.zeros(n6)
    cost = 0.0
    nvalid = 0
    Hll_inv = np.zeros((N, 3, 3))
    gls = np.zeros((N, 3))
    Hpls = np.zeros((N, n6, 3))
    for i in range(N):
        Hll = np.zeros((3, 3))
        gl = np.zeros(3)
        constrained = False
        Hpl = np.zeros((n6, 3))
        Hpp = np.zeros((n6, n6))
        gp = np.zeros(n6)
        for c in range(C):
            if mask is not None and not mask[c, i]:
                continue
            e, Jp, Jl, valid = factor(poses[c], calib[c], sigma[c], points[i], obs[c, i])
            cost += 0.5 * e.dot(e)
            nvalid += int(valid)
            constrained = constrained or valid
            sl = slice(6 * c, 6 * c + 6)
            Hll += Jl.T @ Jl
            gl -= Jl.T @ e
            Hpl[sl] += Jp.T @ Jl
            Hpp[sl, sl] += Jp.T @ Jp
            gp[sl] -= Jp.T @ e
        if prior_w is not None and prior_w[i] > 0:
            d = points[i] - prior_xyz[i]
            Hll += prior_w[i] * np.eye(3)
            gl -= prior_w[i] * d
            cost += 0.5 * prior_w[i] * d.dot(d)
            constrained = True
        if lam > 0:
            Hll = Hll + lam * np.diag(np.diag(Hll))
        elif lam < 0:                                        # GTSAM 3.2.1's default damping: |lam| * I on every variable
            Hll = Hll - lam * np.eye(3)
        # a landmark whose 3x3 block is not positive definite (fewer than two valid views and no
        # prior; the reference asserts >= 2 factors per landmark, bundle_adjust.cpp:158) is
        # unconstrained: it is left where it is and eliminates nothing (the library applies the
        # same rule through its Cholesky pivot test, ba_math.h point_finish)
        if constrained and np.linalg.eigvalsh(Hll)[0] > 1e-13 * np.trace(Hll):
            Hi = np.linalg.inv(Hll)
        else:
            Hi = np.zeros((3, 3))
        S += Hpp - Hpl @ Hi @ Hpl.T
        g += gp - Hpl @ Hi @ gl
        Hll_inv[i] = Hi
        gls[i] = gl
        Hpls[i] = Hpl
    return S, g, cost, nvalid, (Hll_inv, gls, Hpls)


def backsub(pieces, dpose):
    Hi, gl, Hpl = pieces
    return np.einsum("nij,nj->ni", Hi, gl - np.einsum("nkj,k->nj", Hpl, dpose))


def cost_only(poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None):
    C, N = obs.shape[0], obs.shape[1]
    cost = 0.0
    for i in range(N):
        for c in range(C):
            if mask is not None and not mask[c, i]:
                continue
            e, _, _, _ = factor(poses[c], calib[c], sigma[c], points[i], obs[c, i])
            cost += 0.5 * e.dot(e)
        if prior_w is not None and prior_w[i] > 0:
            d = points[i] - prior_xyz[i]
            cost += 0.5 * prior_w[i] * d.dot(d)
    return cost


def pose_prior_terms(poses, prior_poses, prior_sigmas, prior_mask):
    """
    PriorFactor<Pose3> (bundle_adjust.cpp:273): e = Local_{T0}(T) ~ (Log(R0^T R), R0^T (t - t0)),
    Jacobian ~ I6 (first order), diagonal sigmas in file order (rot x3, trans x3).
    Returns H (6C,6C) block-diagonal, g (6C,), cost.
    """
    C = poses.shape[0]
    H = np.zeros((6 * C, 6 * C))
    g = np.zeros(6 * C)
    cost = 0.0
    for c in range(C):
        if not prior_mask[c]:
            continue
        R0 = prior_poses[c, :9].reshape(3, 3)
        R = poses[c, :9].reshape(3, 3)
        e = np.concatenate([so3_log(R0.T @ R), R0.T @ (poses[c, 9:] - prior_poses[c, 9:])]) / prior_sigmas[c]
        W = np.diag(1.0 / prior_sigmas[c] ** 2)
        H[6 * c:6 * c + 6, 6 * c:6 * c + 6] = W
        g[6 * c:6 * c + 6] = -e / prior_sigmas[c]
        cost += 0.5 * e.dot(e)
    return H, g, cost


def gauss_newton(poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None,
                 pose_prior=None, iters=10, lam=0.0):
    """Plain GN (lam = 0) / fixed-damping loop; returns new state and the cost history
    (cost BEFORE each iteration, then the final cost)."""
    poses = poses.copy()
    points = points.copy()
    hist = []
    for _ in range(iters):
        S, g, cost, _, pieces = linearize(poses, calib, sigma, points, obs, mask, prior_w, prior_xyz, lam)
        if pose_prior is not None:
            Hp, gp, cp = pose_prior_terms(poses, *pose_prior)
            S = S + Hp
            g = g + gp
            cost += cp
        hist.append(cost)
        if lam > 0:
            S = S + lam * np.diag(np.diag(S))
        elif lam < 0:
            S = S - lam * np.eye(len(S))
        dpose = np.linalg.solve(S, g)
        dpts = backsub(pieces, dpose)
        for c in range(poses.shape[0]):
            poses[c] = retract_pose(poses[c], dpose[6 * c:6 * c + 6])
        points = points + dpts
    final = cost_only(poses, calib, sigma, points, obs, mask, prior_w, prior_xyz)
    if pose_prior is not None:
        final += pose_prior_terms(poses, *pose_prior)[2]
    hist.append(final)
    return poses, points, hist


def dense_reference_step(poses, calib, sigma, points, obs):
    """Full (un-eliminated) Gauss-Newton system J^T J d = -J^T e, used to validate the Schur form."""
    C, N = obs.shape[0], obs.shape[1]
    nv = 6 * C + 3 * N
    H = np.zeros((nv, nv))
    b = np.zeros(nv)
    for i in range(N):
        for c in range(C):
            e, Jp, Jl, _ = factor(poses[c], calib[c], sigma[c], points[i], obs[c, i])
            J = np.zeros((2, nv))
            J[:, 6 * c:6 * c + 6] = Jp
            J[:, 6 * C + 3 * i:6 * C + 3 * i + 3] = Jl
            H += J.T @ J
            b -= J.T @ e
    return H, b


# ---------------------------------------------------------------------------------------------
# General (sparse-visibility) form: P poses, each with a camera id; observations listed per landmark
# (CSR).  Same factor maths; this is the graph bundle_adjust.cpp:245-298 builds from the file set.
# ---------------------------------------------------------------------------------------------

def sparse_linearize(poses, pose_cam, calib, sigma, points, obs_ptr, obs_pose, obs_uv, prior_w=None, prior_xyz=None,
                     lam=0.0, additive=False):
    """Returns S (6P,6P), g (6P,), cost, n_valid and per-landmark pieces for the back-substitution.
    lam damps the landmark blocks before they are eliminated: lam * diag(H_ll) (Marquardt), or -- additive=True -- lam * I, what
    GTSAM 3.2.1's LevenbergMarquardtParams default (diagonalDamping = false) adds to every variable; the caller damps the pose block
    of the reduced system the same way."""
    P, N = len(poses), len(points)
    S = np.zeros((6 * P, 6 * P))
    g = np.zeros(6 * P)
    cost = 0.0
    nvalid = 0
    pieces = []
    for i in range(N):
        Hll = np.zeros((3, 3))
        gl = np.zeros(3)
        rows = []
        constrained = False
        for k in range(obs_ptr[i], obs_ptr[i + 1]):
            j = obs_pose[k]
            c = pose_cam[j]
            e, Jp, Jl, valid = factor(poses[j], calib[c], sigma[c], points[i], obs_uv[k])
            cost += 0.5 * e.dot(e)
            nvalid += int(valid)
            constrained = constrained or valid
            Hll += Jl.T @ Jl
            gl -= Jl.T @ e
            rows.append((j, Jp.T @ Jl, Jp.T @ Jp, -Jp.T @ e))
        if prior_w is not None and prior_w[i] > 0:
            d = points[i] - prior_xyz[i]
            Hll += prior_w[i] * np.eye(3)
            gl -= prior_w[i] * d
            cost += 0.5 * prior_w[i] * d.dot(d)
            constrained = True
        if lam:
            Hll = Hll + (lam * np.eye(3) if additive else lam * np.diag(np.diag(Hll)))
        if constrained and np.linalg.eigvalsh(Hll)[0] > 1e-13 * np.trace(Hll):
            Hi = np.linalg.inv(Hll)
        else:
            Hi = np.zeros((3, 3))
        for (j, Hpl, Hpp, gp) in rows:
            S[6 * j:6 * j + 6, 6 * j:6 * j + 6] += Hpp
            g[6 * j:6 * j + 6] += gp - Hpl @ Hi @ gl
            for (j2, Hpl2, _, _) in rows:
                S[6 * j:6 * j + 6, 6 * j2:6 * j2 + 6] -= Hpl @ Hi @ Hpl2.T
        pieces.append((Hi, gl, [(j, Hpl) for (j, Hpl, _, _) in rows]))
    return S, g, cost, nvalid, pieces


def sparse_backsub(pieces, dpose):
    out = np.zeros((len(pieces), 3))
    for i, (Hi, gl, rows) in enumerate(pieces):
        r = gl.copy()
        for (j, Hpl) in rows:
            r -= Hpl.T @ dpose[6 * j:6 * j + 6]
        out[i] = Hi @ r
    return out


def sparse_pose_prior_terms(poses, idx, prior_poses, sigmas):
    """PriorFactor<Pose3> on poses idx[k] with prior prior_poses[k] and sigmas[k] (6)."""
    n = 6 * len(poses)
    H = np.zeros((n, n))
    g = np.zeros(n)
    cost = 0.0
    for k, j in enumerate(idx):
        R0 = prior_poses[k, :9].reshape(3, 3)
        R = poses[j, :9].reshape(3, 3)
        e = np.concatenate([so3_log(R0.T @ R), R0.T @ (poses[j, 9:] - prior_poses[k, 9:])])
        w = 1.0 / sigmas[k] ** 2
        H[6 * j:6 * j + 6, 6 * j:6 * j + 6] += np.diag(w)
        g[6 * j:6 * j + 6] -= w * e
        cost += 0.5 * float((w * e * e).sum())
    return H, g, cost


def sparse_cost(poses, pose_cam, calib, sigma, points, obs_ptr, obs_pose, obs_uv, prior_w=None, prior_xyz=None):
    cost = 0.0
    for i in range(len(points)):
        for k in range(obs_ptr[i], obs_ptr[i + 1]):
            j = obs_pose[k]
            c = pose_cam[j]
            e, _, _, _ = factor(poses[j], calib[c], sigma[c], points[i], obs_uv[k])
            cost += 0.5 * e.dot(e)
        if prior_w is not None and prior_w[i] > 0:
            d = points[i] - prior_xyz[i]
            cost += 0.5 * prior_w[i] * d.dot(d)
    return cost


def between_error(T1, T2, Tm):
    """BetweenFactor<Pose3>::evaluateError of GTSAM 3.2.1 (bundle_adjust.cpp:301-309): h = T1^-1 T2,
    measured.localCoordinates(h) in the first-order chart = (Log(Rm^T Rh), Rm^T (th - tm)); order [omega, v].
    Returns e (6), H1 (6,6), H2 (6,6): the Jacobians of `between` only (-Ad(h^-1), I), as GTSAM uses them."""
    R1, t1 = T1[:9].reshape(3, 3), T1[9:]
    R2, t2 = T2[:9].reshape(3, 3), T2[9:]
    Rm, tm = Tm[:9].reshape(3, 3), Tm[9:]
    Rh, th = R1.T @ R2, R1.T @ (t2 - t1)
    e = np.concatenate([so3_log(Rm.T @ Rh), Rm.T @ (th - tm)])
    Ri, ti = Rh.T, -Rh.T @ th                                  # h^-1
    Ad = np.zeros((6, 6))
    Ad[:3, :3] = Ri
    Ad[3:, :3] = skew(ti) @ Ri
    Ad[3:, 3:] = Ri
    return e, -Ad, np.eye(6)


def sparse_between_terms(poses, odo_from, odo_to, odo_meas, odo_sigmas):
    """Normal-equation contribution (H, g = -J^T r, cost) of the odometry factors."""
    n = 6 * len(poses)
    H = np.zeros((n, n))
    g = np.zeros(n)
    cost = 0.0
    for k in range(len(odo_from)):
        a, b = int(odo_from[k]), int(odo_to[k])
        e, H1, H2 = between_error(poses[a], poses[b], odo_meas[k])
        W = np.diag(1.0 / odo_sigmas[k] ** 2)
        sa, sb = slice(6 * a, 6 * a + 6), slice(6 * b, 6 * b + 6)
        H[sa, sa] += H1.T @ W @ H1
        H[sa, sb] += H1.T @ W @ H2
        H[sb, sa] += H2.T @ W @ H1
        H[sb, sb] += H2.T @ W @ H2
        g[sa] -= H1.T @ W @ e
        g[sb] -= H2.T @ W @ e
        cost += 0.5 * float(e @ W @ e)
    return H, g, cost
