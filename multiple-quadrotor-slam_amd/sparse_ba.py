"""
General sparse-visibility bundle adjustment over the kernels of csrc/ba_sparse.hip -- the solver behind
the CLI-compatible tool `tools/bundle_adjust.py`, i.e. the counterpart of
`performBundleAdjustment` with iSAM_version = 0 (Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:190-330:
build the whole graph, one batch LevenbergMarquardtOptimizer::optimize() at the last step).

Input: a `ba_io.SparseProblem` (poses, per-pose camera id, CSR observations, point / pose priors, and -- with
`useOdometry` -- the odometry BetweenFactors of bundle_adjust.cpp:301-309, added to the reduced camera system by
`mqs_sba_between_dev`).
"""
import ctypes

import numpy as np

from . import _lib
from .bundle_adjustment import (LM_ABS_TOL, LM_LAMBDA_FACTOR, LM_LAMBDA_INITIAL, LM_LAMBDA_UPPER, LM_MAX_ITERATIONS,
                                LM_REL_TOL)


def _torch():
    import torch
    return torch


def _sp():
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def build_pairs(obs_ptr):
    """All (a, b), a <= b, of observation indices within each landmark (observations are sorted by pose
    inside a landmark, so pose(a) <= pose(b))."""
    pa, pb = [], []
    for i in range(len(obs_ptr) - 1):
        k0, k = int(obs_ptr[i]), int(obs_ptr[i + 1] - obs_ptr[i])
        if k == 0:
            continue
        r, c = np.triu_indices(k)
        pa.append(k0 + r)
        pb.append(k0 + c)
    if not pa:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(pa).astype(np.int64), np.concatenate(pb).astype(np.int64)


def group_pairs(pair_a, pair_b, obs_pose, n_poses):
    """Sorts the pair list by (pose of a, pose of b) -- stable, so the order inside a group is the landmark order -- and
    returns (pair_a, pair_b, group_ptr): group k = pairs [group_ptr[k], group_ptr[k + 1]) all feed ONE 6 x 6 block of the
    reduced system (mqs_sba_linearize_grouped_dev)."""
    if len(pair_a) == 0:
        return pair_a, pair_b, np.zeros(1, np.int64)
    op = np.asarray(obs_pose, dtype=np.int64)
    # canonical orientation: the pose of a is never after the pose of b (the grouped stage has ONE writer per block and its
    # mirror image; with observations in any order inside a landmark, (ja, jb) and (jb, ja) would be two writers)
    swap = op[pair_a] > op[pair_b]
    if swap.any():
        pair_a, pair_b = np.where(swap, pair_b, pair_a), np.where(swap, pair_a, pair_b)
    key = op[pair_a] * np.int64(n_poses) + op[pair_b]
    order = np.argsort(key, kind="stable")
    key = key[order]
    cuts = np.nonzero(key[1:] != key[:-1])[0] + 1
    return pair_a[order], pair_b[order], np.concatenate([[0], cuts, [len(key)]]).astype(np.int64)


def sort_observations_by_pose(problem):
    """Returns a copy of the problem whose observations are sorted by pose index within each landmark (stable)."""
    ptr = np.asarray(problem.obs_ptr, dtype=np.int64)
    op = np.asarray(problem.obs_pose)
    landmark = np.repeat(np.arange(len(ptr) - 1, dtype=np.int64), np.diff(ptr))
    order = np.lexsort((op, landmark))                    # by landmark, then pose; lexsort is stable
    return problem._replace(obs_pose=op[order].copy(), obs_uv=np.asarray(problem.obs_uv)[order].copy())


def sort_observations_dev(obs_ptr, obs_pose, obs_uv, n_landmarks, n_poses, want_order=False):
    """`sort_observations_by_pose` on the device (csrc/pair_group.hip: key = landmark * P + pose, the stable radix sort of the pair
    grouping, one gather): device tensors in, (obs_pose, obs_uv[, order]) sorted by pose inside every landmark out."""
    torch = _torch()
    M = int(obs_pose.numel())
    pose_out, uv_out = torch.empty_like(obs_pose), torch.empty_like(obs_uv)
    order = torch.empty(M, dtype=torch.int32, device=obs_pose.device) if want_order else None
    if M:
        ws = torch.empty(int(_lib.lib().mqs_sba_sort_observations_workspace_bytes(M)), dtype=torch.uint8, device=obs_pose.device)
        _lib.check(_lib.lib().mqs_sba_sort_observations_dev(_p(obs_ptr), _p(obs_pose), _p(obs_uv), int(n_landmarks), M, int(n_poses),
                                                            _p(pose_out), _p(uv_out), _p(order), _p(ws), ws.numel(), _sp()))
    return (pose_out, uv_out, order) if want_order else (pose_out, uv_out)


def group_pairs_dev(obs_ptr_host, obs_ptr, obs_pose, n_poses):
    """The grouped pair list of `group_pairs(build_pairs(...))`, built on the device (csrc/pair_group.hip): obs_ptr_host the
    CSR offsets as numpy (for the pair counts: Q must be known to size the outputs), obs_ptr / obs_pose their device tensors
    (observations sorted by pose inside every landmark).  Returns device tensors (pair_a, pair_b, group_ptr [G + 1])."""
    torch = _torch()
    dev = obs_pose.device
    k = np.diff(np.asarray(obs_ptr_host, dtype=np.int64))
    pair_off = np.concatenate([[0], np.cumsum(k * (k + 1) // 2)]).astype(np.int64)
    Q, N = int(pair_off[-1]), len(k)
    cap = int(min(Q, n_poses * (n_poses + 1) // 2)) + 1
    pa = torch.empty(max(Q, 1), dtype=torch.int64, device=dev)
    pb = torch.empty(max(Q, 1), dtype=torch.int64, device=dev)
    gp = torch.empty(cap, dtype=torch.int64, device=dev)
    ng = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(int(_lib.lib().mqs_sba_group_pairs_workspace_bytes(Q)), dtype=torch.uint8, device=dev)
    off_d = torch.from_numpy(pair_off).to(dev)
    _lib.check(_lib.lib().mqs_sba_group_pairs_dev(_p(obs_ptr), _p(obs_pose), N, _p(off_d), Q, int(n_poses), _p(pa), _p(pb), _p(gp), cap,
                                                  _p(ng), _p(ws), ws.numel(), _sp()))
    G = int(ng.item())                                    # the one synchronisation of the set-up
    return pa[:Q], pb[:Q], gp[:G + 1]


def _make_structs():
    import ctypes
    vp, i64, i32, f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double

    class ProblemDev(ctypes.Structure):                     # include/mqslam.h: mqs_sba_problem_dev
        _fields_ = ([(k, i64) for k in ("P", "N", "M", "Q", "G")] +
                    [(k, vp) for k in ("poses", "poses_new", "points", "points_new", "pose_cam", "calib", "sigma", "obs_ptr", "obs_pose",
                                       "obs_uv", "pair_a", "pair_b", "group_ptr", "prior_w", "prior_xyz", "pose_prior_idx",
                                       "pose_prior_poses", "pose_prior_sigmas", "odo_from", "odo_to", "odo_meas", "odo_sigmas")] +
                    [("n_odo", i64), ("half_bandwidth", i64), ("S", vp), ("g", vp), ("workspace", vp), ("workspace_bytes", i64),
                     ("n_pose_prior", i32), ("reserved", i32)])

    class LmParams(ctypes.Structure):                       # mqs_sba_lm_params
        _fields_ = [(k, f64) for k in ("lambda_initial", "lambda_factor", "lambda_upper", "abs_tol", "rel_tol")] + \
                   [("max_iterations", i32), ("damping", i32)]
    return ProblemDev, LmParams


_ProblemDev, _LmParams = _make_structs()


class SparseBundleAdjuster:
    def __init__(self, problem, device="cuda:0"):
        torch = _torch()
        pr = problem                           # observations in the order they came (recorder / file order inside a landmark)
        self.problem = pr
        dev = torch.device(device)
        self.dev = dev
        f64, i32, i64 = torch.float64, torch.int32, torch.int64
        # every host array of the problem in ONE upload (a problem behind a keyframe of the SLAM loop is ~20 arrays of a few KB:
        # one copy each was half of the set-up's time): laid out back to back, 16-byte aligned, then viewed per array
        np_dt = {f64: np.float64, i32: np.int32, i64: np.int64}
        staged, blobs, off = {}, [], 0
        def stage(name, a, dt):
            nonlocal off
            a = np.ascontiguousarray(np.asarray(a), dtype=np_dt[dt])
            staged[name] = (off, a.shape, dt, a.nbytes)
            blobs.append(a.reshape(-1).view(np.uint8))
            pad = (-a.nbytes) % 16
            if pad:
                blobs.append(np.zeros(pad, np.uint8))
            off += a.nbytes + pad
        def uploaded():
            flat = torch.from_numpy(np.concatenate(blobs) if blobs else np.zeros(16, np.uint8)).to(dev)
            out = {}
            for name, (o, shape, dt, nbytes) in staged.items():
                out[name] = flat[o:o + nbytes].view(dt).view(*shape) if nbytes else torch.empty(shape, dtype=dt, device=dev)
            return out
        self.P, self.N, self.M = len(pr.poses), len(pr.points), len(pr.obs_pose)
        has_prior = pr.prior_w is not None and np.any(pr.prior_w > 0)
        self.npp = len(pr.pose_prior_idx)
        self.n_odo = len(pr.odo_from)
        stage("poses", pr.poses, f64); stage("pose_cam", pr.pose_cam, i32); stage("calib", pr.calib, f64); stage("sigma", pr.sigma, f64)
        stage("points", pr.points, f64); stage("obs_ptr", pr.obs_ptr, i64); stage("obs_pose", pr.obs_pose, i32)
        stage("obs_uv", np.asarray(pr.obs_uv).reshape(-1, 2), f64)
        if has_prior:
            stage("prior_w", pr.prior_w, f64); stage("prior_xyz", pr.prior_xyz, f64)
        if self.npp:
            stage("pp_idx", pr.pose_prior_idx, i32); stage("pp_poses", pr.poses[pr.pose_prior_idx], f64)      # prior = initial pose (:273)
            stage("pp_sigmas", pr.pose_prior_sigmas, f64)
        if self.n_odo:
            stage("odo_from", pr.odo_from, i32); stage("odo_to", pr.odo_to, i32); stage("odo_meas", pr.odo_meas, f64)
            stage("odo_sigmas", pr.odo_sigmas, f64)
        up = uploaded()
        self.poses, self.pose_cam, self.calib, self.sigma = up["poses"], up["pose_cam"], up["calib"], up["sigma"]
        self.points, self.obs_ptr = up["points"], up["obs_ptr"]
        # the observations sorted by pose inside every landmark, and every pair of observations of a landmark grouped by pose
        # pair: both built on the device (numpy twins `sort_observations_by_pose`, `build_pairs` / `group_pairs` above: the same
        # lists; the host sort alone was 7 ms of a 9 ms set-up at the kt2 shape, round 3)
        with torch.cuda.device(dev):
            self.obs_pose, self.obs_uv = sort_observations_dev(self.obs_ptr, up["obs_pose"], up["obs_uv"], len(pr.points), len(pr.poses))
            self.pair_a, self.pair_b, self.group_ptr = group_pairs_dev(pr.obs_ptr, self.obs_ptr, self.obs_pose, len(pr.poses))
        self.Q, self.G = int(self.pair_a.numel()), int(self.group_ptr.numel()) - 1
        self.prior_w, self.prior_xyz = up.get("prior_w"), up.get("prior_xyz")
        self.pp_idx, self.pp_poses, self.pp_sigmas = up.get("pp_idx"), up.get("pp_poses"), up.get("pp_sigmas")
        # half bandwidth of the reduced camera system: poses coupled by a common landmark or an odometry link are at
        # most `dmax` apart in pose index (observations are sorted by pose inside a landmark)
        ptr = np.asarray(pr.obs_ptr, dtype=np.int64)
        has = ptr[1:] > ptr[:-1]
        if has.any():                                            # largest pose-index span of a landmark (observations in any order)
            op_h = np.asarray(pr.obs_pose, dtype=np.int64)
            starts = ptr[:-1][has]
            dmax = int((np.maximum.reduceat(op_h, starts) - np.minimum.reduceat(op_h, starts)).max())
        else:
            dmax = 0
        if len(pr.odo_from):
            dmax = max(dmax, int(np.abs(np.asarray(pr.odo_from, dtype=np.int64) - np.asarray(pr.odo_to, dtype=np.int64)).max()))
        self.half_bandwidth = 6 * (dmax + 1) - 1
        if self.n_odo:
            self.odo_from, self.odo_to = up["odo_from"], up["odo_to"]
            self.odo_meas, self.odo_sigmas = up["odo_meas"], up["odo_sigmas"]
            self.odo_cost = torch.zeros(1, dtype=f64, device=dev)
        n6 = 6 * self.P
        self.n6 = n6
        self.S = torch.empty(n6 * n6, dtype=f64, device=dev)
        self.g = torch.empty(n6, dtype=f64, device=dev)
        self.info = torch.zeros(4, dtype=f64, device=dev)
        self.cost_out = torch.zeros(2, dtype=f64, device=dev)
        self.bad = torch.zeros(1, dtype=torch.int32, device=dev)
        self.poses_new = torch.empty_like(self.poses)
        self.points_new = torch.empty_like(self.points)
        self.ws = torch.empty(int(_lib.lib().mqs_sba_workspace_bytes(self.P, self.N, self.M)), dtype=torch.uint8,
                              device=dev)

    def linearize(self, lam=0.0):
        _lib.check(_lib.lib().mqs_sba_linearize_grouped_dev(
            _p(self.poses), _p(self.pose_cam), self.P, _p(self.calib), _p(self.sigma), _p(self.points), self.N,
            _p(self.obs_ptr), _p(self.obs_pose), _p(self.obs_uv), self.M, _p(self.pair_a), _p(self.pair_b), self.Q,
            _p(self.group_ptr), self.G, _p(self.prior_w), _p(self.prior_xyz), _p(self.pp_idx), _p(self.pp_poses), _p(self.pp_sigmas), self.npp,
            float(lam), _p(self.S), _p(self.g), _p(self.info), _p(self.ws), self.ws.numel(), _sp()))
        if self.n_odo:
            self._between(self.poses, self.S, self.g)
        return self.S.view(self.n6, self.n6), self.g

    def _between(self, poses, S=None, g=None):
        """Odometry factors at `poses`: into (S, g) when given; returns nothing (cost in self.odo_cost)."""
        self.odo_cost.zero_()
        _lib.check(_lib.lib().mqs_sba_between_dev(_p(poses), self.P, _p(self.odo_from), _p(self.odo_to), _p(self.odo_meas),
                                                  _p(self.odo_sigmas), self.n_odo, _p(S), _p(g), _p(self.odo_cost), _sp()))

    def solve(self, lam=0.0):
        """Destroys S (replaced by its Cholesky factor) and g (replaced by dpose); retracts the poses."""
        _lib.check(_lib.lib().mqs_sba_solve_banded_dev(_p(self.S), _p(self.g), self.P, self.half_bandwidth, float(lam),
                                                       _p(self.poses), _p(self.poses_new), _p(self.bad), _sp()))
        return self.g

    def backsub(self, lam=0.0):
        _lib.check(_lib.lib().mqs_sba_backsub_dev(
            _p(self.poses), _p(self.pose_cam), self.P, _p(self.calib), _p(self.sigma), _p(self.points), self.N,
            _p(self.obs_ptr), _p(self.obs_pose), _p(self.obs_uv), self.M, _p(self.prior_w), _p(self.prior_xyz),
            float(lam), _p(self.g), _p(self.points_new), _p(self.ws), self.ws.numel(), _sp()))
        return self.points_new

    def cost(self, poses=None, points=None):
        poses = self.poses if poses is None else poses
        points = self.points if points is None else points
        _lib.check(_lib.lib().mqs_sba_cost_dev(
            _p(poses), _p(self.pose_cam), self.P, _p(self.calib), _p(self.sigma), _p(points), self.N, _p(self.obs_ptr),
            _p(self.obs_pose), _p(self.obs_uv), self.M, _p(self.prior_w), _p(self.prior_xyz), _p(self.cost_out),
            _p(self.ws), self.ws.numel(), _sp()))
        c = float(self.cost_out[0].item())
        if self.npp:
            c += self._pose_prior_cost(poses)
        if self.n_odo:
            self._between(poses)
            c += float(self.odo_cost.item())
        return c

    def worst_residuals(self, poses=None, points=None, with_min_depth=False):
        """Per landmark the largest pixel residual of its observations at (poses, points) (default: the current estimate);
        +inf when an observation lies behind its camera.  numpy [N]; with_min_depth: also the landmark's smallest depth along the
        optical axes of the cameras that see it."""
        poses = self.poses if poses is None else poses
        points = self.points if points is None else points
        out = _torch().empty((2, max(self.N, 1)), dtype=_torch().float64, device=self.dev)
        _lib.check(_lib.lib().mqs_sba_worst_residual_dev(
            _p(poses), _p(self.pose_cam), self.P, _p(self.calib), _p(self.sigma), _p(points), self.N, _p(self.obs_ptr),
            _p(self.obs_pose), _p(self.obs_uv), self.M, _p(out[0]), _p(out[1]) if with_min_depth else None, _p(self.ws), self.ws.numel(), _sp()))
        h = out[:, :self.N].cpu().numpy()
        return (h[0], h[1]) if with_min_depth else h[0]

    def _pose_prior_cost(self, poses):
        P = poses.cpu().numpy()
        pr = self.problem
        cost = 0.0
        for k, j in enumerate(pr.pose_prior_idx):
            R0 = pr.poses[j, :9].reshape(3, 3)
            R = P[j, :9].reshape(3, 3)
            Rr = R0.T @ R
            th = np.arccos(min(1.0, max(-1.0, 0.5 * (np.trace(Rr) - 1))))
            v = np.array([Rr[2, 1] - Rr[1, 2], Rr[0, 2] - Rr[2, 0], Rr[1, 0] - Rr[0, 1]])
            w = 0.5 * v if th < 1e-10 else v * th / (2 * np.sin(th))
            e = np.concatenate([w, R0.T @ (P[j, 9:] - pr.poses[j, 9:])]) / pr.pose_prior_sigmas[k]
            cost += 0.5 * float(e.dot(e))
        return cost

    def step(self, lam):
        self.linearize(lam)
        self.solve(lam)
        self.backsub(lam)

    def accept(self):
        self.poses, self.poses_new = self.poses_new, self.poses
        self.points, self.points_new = self.points_new, self.points

    def _problem_struct(self):
        """The problem as the C ABI's `mqs_sba_problem_dev` (include/mqslam.h): device pointers of the tensors this object holds."""
        z = lambda t: _p(t) if t is not None else None
        return _ProblemDev(
            P=self.P, N=self.N, M=self.M, Q=self.Q, G=self.G, poses=_p(self.poses), poses_new=_p(self.poses_new), points=_p(self.points),
            points_new=_p(self.points_new), pose_cam=_p(self.pose_cam), calib=_p(self.calib), sigma=_p(self.sigma), obs_ptr=_p(self.obs_ptr),
            obs_pose=_p(self.obs_pose), obs_uv=_p(self.obs_uv), pair_a=_p(self.pair_a), pair_b=_p(self.pair_b), group_ptr=_p(self.group_ptr),
            prior_w=z(self.prior_w), prior_xyz=z(self.prior_xyz), pose_prior_idx=z(self.pp_idx), pose_prior_poses=z(self.pp_poses),
            pose_prior_sigmas=z(self.pp_sigmas), odo_from=_p(self.odo_from) if self.n_odo else None, odo_to=_p(self.odo_to) if self.n_odo else None,
            odo_meas=_p(self.odo_meas) if self.n_odo else None, odo_sigmas=_p(self.odo_sigmas) if self.n_odo else None, n_odo=self.n_odo,
            half_bandwidth=self.half_bandwidth, S=_p(self.S), g=_p(self.g), workspace=_p(self.ws), workspace_bytes=self.ws.numel(),
            n_pose_prior=self.npp, reserved=0)

    def optimize(self, iters=LM_MAX_ITERATIONS, mode="lm", verbose=False, damping="gtsam"):
        """mode "lm": GTSAM 3.2.1's default Levenberg-Marquardt schedule (bundle_adjust.cpp:323-324);
        mode "gn": plain Gauss-Newton.  damping: "gtsam" (default) adds lambda * I to the landmark blocks and to the reduced
        system, as GTSAM 3.2.1's default LevenbergMarquardtParams do (diagonalDamping = false) -- what the optimiser of
        bundle_adjust.cpp:323-324 runs; "marquardt" scales the diagonals by (1 + lambda) instead (invariant to the units of
        the variables; this build's default until round 3): the same optimum by a different iterate path.
        Returns the cost history.
        The LM loop itself runs inside the library (`mqs_sba_optimize_lm_dev`, csrc/ba_sparse.hip: one host synchronisation per
        trial); `optimize_host_loop` is the same schedule driven from here over the individual entry points (verbose runs, and
        the comparison in tests/test_ba_files.py)."""
        if mode != "lm" or verbose:
            return self.optimize_host_loop(iters, mode, verbose, damping)
        import ctypes
        lm = _LmParams(lambda_initial=LM_LAMBDA_INITIAL, lambda_factor=LM_LAMBDA_FACTOR, lambda_upper=LM_LAMBDA_UPPER, abs_tol=LM_ABS_TOL,
                       rel_tol=LM_REL_TOL, max_iterations=int(min(iters, LM_MAX_ITERATIONS)), damping={"gtsam": 0, "marquardt": 1}[damping])
        pr = self._problem_struct()
        hist = np.zeros(lm.max_iterations + 1)
        n = ctypes.c_int32(0)
        with _torch().cuda.device(self.dev):
            _lib.check(_lib.lib().mqs_sba_optimize_lm_dev(ctypes.byref(pr), ctypes.byref(lm), hist.ctypes.data_as(_lib.c_f64p), len(hist),
                                                          ctypes.byref(n), _sp()))
        return [float(v) for v in hist[:n.value]]

    def optimize_host_loop(self, iters=LM_MAX_ITERATIONS, mode="lm", verbose=False, damping="gtsam"):
        """The optimiser's loop driven from Python over the library's entry points (see `optimize`)."""
        sgn = {"marquardt": 1.0, "gtsam": -1.0}[damping]
        hist = [self.cost()]
        if mode == "gn":
            for it in range(iters):
                self.step(0.0)
                if int(self.bad.item()) != 0:
                    # the undamped reduced system was not positive definite: the step is garbage, nothing is accepted
                    raise RuntimeError("Gauss-Newton iteration %d: the reduced camera system is not positive definite "
                                       "(gauge not fixed, or degenerate geometry); use mode='lm'" % it)
                self.accept()
                hist.append(self.cost())
            return hist
        lam, cur = LM_LAMBDA_INITIAL, hist[0]
        for _ in range(min(iters, LM_MAX_ITERATIONS)):
            improved = False
            while lam <= LM_LAMBDA_UPPER:
                self.step(sgn * lam)
                ok = int(self.bad.item()) == 0
                new = self.cost(self.poses_new, self.points_new) if ok else float("inf")
                if verbose:
                    print("  lm lambda %.1e cost %.6e -> %.6e" % (lam, cur, new))
                if new <= cur:
                    self.accept()
                    lam = max(lam / LM_LAMBDA_FACTOR, 1e-20)
                    improved = True
                    break
                lam *= LM_LAMBDA_FACTOR
            if not improved:
                break
            hist.append(new)
            done = abs(cur - new) < LM_ABS_TOL or abs(cur - new) / max(cur, 1e-300) < LM_REL_TOL
            cur = new
            if done:
                break
        return hist
