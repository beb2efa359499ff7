"""
Bundle adjustment driver over the gfx950 kernels (csrc/ba.hip) -- the in-process counterpart of
the reference's stand-alone tool `Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp`:

  graph structure   bundle_adjust.cpp:268-298  (pose priors on first-frame poses, point priors on
                    step-0 landmarks, one GenericProjectionFactor<Pose3, Point3, Cal3DS2> per
                    observation)
  optimiser         bundle_adjust.cpp:323-324  LevenbergMarquardtOptimizer(graph, values).optimize()
                    -> `optimize(mode="lm")`; `mode="gn"` is the plain Gauss-Newton loop that
                    BASELINE.json's metric counts.

All state lives in device tensors (torch is used for memory, the current stream and
torch.distributed only).  One Gauss-Newton iteration is ONE call into the library
(`mqs_ba_gn_iteration_dev`, csrc/ba_iter.hip) that enqueues, with no host round trip,
    linearise + on-chip Schur elimination  ->  [all-reduce of the 6C x 6C system across ranks]
    -> reduced camera solve + pose retraction  ->  landmark back-substitution.
Multi-GPU: landmarks are sharded across ranks (poses, calibrations replicated); the only
collective is ONE all-reduce (RCCL over xGMI) of (6C)^2 + 6C + 2 doubles per iteration; pose
priors are added identically on every rank after the reduce, so every rank solves the same
system and no broadcast is needed.  Transport: `process_group=sharding.CComm` (an RCCL communicator
inside the library's context: the all-reduce is issued from C between the launches) or a
torch.distributed group / True (the collective is issued from Python between the two halves
`mqs_ba_gn_begin_dev` / `mqs_ba_gn_finish_dev`; what the gloo tests use).
"""
import ctypes
import time

import numpy as np

from . import _lib
from . import sharding

# LevenbergMarquardtParams defaults of GTSAM 3.2.1 [SURVEY.md 8(a) B4]
LM_LAMBDA_INITIAL = 1e-5
LM_LAMBDA_FACTOR = 10.0
LM_LAMBDA_UPPER = 1e5
LM_MAX_ITERATIONS = 100
LM_REL_TOL = 1e-5
LM_ABS_TOL = 1e-5


def _torch():
    import torch
    return torch


def _sp():
    return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class BundleAdjuster:
    """
    Dense-visibility BA problem on one device (this rank's landmark shard).

        poses  (C,12) f64  camera-to-world [R row-major | t]      calib (C,9) f64   sigma (C,) f64
        points (N,3) f64   obs (C,N,2) f64 pixels                 mask (C,N) uint8 or None
        prior_w (N,) f64 / prior_xyz (N,3) f64 or None            (PriorFactor<Point3>)
        pose_prior = (prior_poses (C,12), prior_sigmas (C,6), prior_mask (C,) uint8) or None
    """

    def __init__(self, poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None,
                 pose_prior=None, process_group=None):
        torch = _torch()
        self.dev = points.device
        f64 = torch.float64

        def chk(t, shape, dtype=f64, name=""):
            if t is None:
                return None
            if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
                raise ValueError("%s must be a contiguous %s device tensor" % (name, dtype))
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s must have shape %r, got %r" % (name, shape, tuple(t.shape)))
            return t

        self.C = int(poses.shape[0])
        self.N = int(points.shape[0])
        C, N = self.C, self.N
        if not (1 <= C <= 8):
            raise ValueError("number of cameras must be in [1, 8]")
        poses0 = chk(poses, (C, 12), name="poses")
        self.calib = chk(calib, (C, 9), name="calib")
        self.sigma = chk(sigma, (C,), name="sigma")
        points0 = chk(points, (N, 3), name="points")
        self.obs = chk(obs, (C, N, 2), name="obs")
        self.mask = chk(mask, (C, N), torch.uint8, "mask")
        self.prior_w = chk(prior_w, (N,), name="prior_w")
        self.prior_xyz = chk(prior_xyz, (N, 3), name="prior_xyz")
        if (self.prior_w is None) != (self.prior_xyz is None):
            raise ValueError("prior_w and prior_xyz go together")
        if pose_prior is not None:
            pp, ps, pm = pose_prior
            self.prior_poses = chk(pp, (C, 12), name="prior_poses")
            self.prior_sigmas = chk(ps, (C, 6), name="prior_sigmas")
            self.prior_mask = chk(pm, (C,), torch.uint8, "prior_mask")
        else:
            self.prior_poses = self.prior_sigmas = self.prior_mask = None
        self.ccomm = process_group if isinstance(process_group, sharding.CComm) else None
        self.pg = None if self.ccomm is not None else process_group
        n6 = 6 * C
        self.n6 = n6
        self.lin = torch.zeros(n6 * n6 + n6 + 2, dtype=f64, device=self.dev)
        self.dpose = torch.zeros(n6, dtype=f64, device=self.dev)
        self.info = torch.zeros(2, dtype=f64, device=self.dev)
        self.cost_out = torch.zeros(2, dtype=f64, device=self.dev)
        # the estimate alternates between two buffer pairs; the library's problem handle knows which one is current
        self._poses = [poses0, torch.empty_like(poses0)]
        self._points = [points0, torch.empty_like(points0)]
        ws = int(_lib.lib().mqs_ba_workspace_bytes(C, N))
        self.ws = torch.empty(max(ws, 65536), dtype=torch.uint8, device=self.dev)
        self.lam = 0.0
        self.cost_history = []
        self._h = None
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().mqs_ba_problem_create(
            self.ccomm.ctx.handle if self.ccomm is not None else None, C, N, _p(self._poses[0]), _p(self._poses[1]),
            _p(self.calib), _p(self.sigma), _p(self._points[0]), _p(self._points[1]), _p(self.obs), _p(self.mask),
            _p(self.prior_w), _p(self.prior_xyz), _p(self.prior_poses), _p(self.prior_sigmas), _p(self.prior_mask),
            _p(self.lin), _p(self.dpose), _p(self.info), _p(self.ws), self.ws.numel(), ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mqs_ba_problem_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- current / next estimate (the handle owns the index) ---------------------------
    @property
    def _cur(self):
        return int(_lib.lib().mqs_ba_problem_current(self._h))

    @property
    def poses(self):
        return self._poses[self._cur]

    @property
    def points(self):
        return self._points[self._cur]

    @property
    def poses_new(self):
        return self._poses[1 - self._cur]

    @property
    def points_new(self):
        return self._points[1 - self._cur]

    def accept(self):
        """Makes the trial estimate (poses_new / points_new) the current one."""
        _lib.check(_lib.lib().mqs_ba_problem_set_current(self._h, 1 - self._cur))

    # ---- the four launches -------------------------------------------------------------
    def linearize(self, lam=0.0):
        _lib.check(_lib.lib().mqs_ba_linearize_dev(
            _p(self.poses), _p(self.calib), _p(self.sigma), self.C, _p(self.points), _p(self.obs), _p(self.mask),
            _p(self.prior_w), _p(self.prior_xyz), self.N, float(lam), _p(self.lin), _p(self.ws), self.ws.numel(),
            _sp()))
        return self.lin

    def all_reduce(self, async_op=False):
        """Sum of the reduced camera system over ranks.  async_op=True returns the collective's work handle (or
        None on one rank): the reduce then runs on RCCL's stream and `wait()` orders the current stream after it."""
        if self.ccomm is not None:
            self.ccomm.all_reduce_sum_(self.lin)                 # on the current stream, in order: nothing to wait for
            return None
        if self.pg is not None:
            return sharding.all_reduce_sum_(self.lin, None if self.pg is True else self.pg, async_op=async_op)
        return None

    def solve(self, lam=0.0, retract_into=None):
        out = self.poses_new if retract_into is None else retract_into
        _lib.check(_lib.lib().mqs_ba_solve_dev(
            _p(self.lin), self.C, _p(self.poses), _p(self.prior_poses), _p(self.prior_sigmas), _p(self.prior_mask),
            float(lam), _p(self.dpose), _p(out), _p(self.info), _sp()))
        return self.dpose

    def backsub(self, lam=0.0, into=None):
        out = self.points_new if into is None else into
        _lib.check(_lib.lib().mqs_ba_backsub_dev(
            _p(self.poses), _p(self.calib), _p(self.sigma), self.C, _p(self.points), _p(self.obs), _p(self.mask),
            _p(self.prior_w), _p(self.prior_xyz), self.N, float(lam), _p(self.dpose), _p(out), _sp()))
        return out

    def cost(self, poses=None, points=None):
        """0.5 * sum |r/sigma|^2 over this rank's factors and point priors (device tensor [cost, count])."""
        poses = self.poses if poses is None else poses
        points = self.points if points is None else points
        _lib.check(_lib.lib().mqs_ba_cost_dev(
            _p(poses), _p(self.calib), _p(self.sigma), self.C, _p(points), _p(self.obs), _p(self.mask),
            _p(self.prior_w), _p(self.prior_xyz), self.N, _p(self.cost_out), _p(self.ws), self.ws.numel(), _sp()))
        return self.cost_out

    # ---- iterations --------------------------------------------------------------------
    def check(self):
        """Raises RuntimeError when a bounded device-side wait of an iteration gave up (the finalizer pieces of the fused tail,
        or a peer's row of the reduced system: csrc/ba.hip, csrc/peer_dev.h) -- the estimate is then not valid.  Synchronises
        the current stream.  Called wherever this class hands results to the host (`total_cost`, `optimize`,
        `gauss_newton_iterations`); call it yourself after a run of bare `gauss_newton_iteration`s."""
        _lib.check(_lib.lib().mqs_ba_problem_status(self._h, _sp()))
        if self.ccomm is not None and self.ccomm.peer_state() and self.ccomm.peer_timed_out():
            raise RuntimeError("a rank's row of the reduced camera system did not arrive within the wait's bound (peer transport: 30 s for a problem's first reduction, 2 s afterwards)")

    def gauss_newton_iteration(self, lam=0.0, overlap=None):
        """One undamped (lam = 0) or fixed-damping iteration, fully asynchronous.  `overlap`: a callable that
        enqueues work independent of this problem on the current stream; it is issued between the start of the
        all-reduce and the wait for it, so that the latency-bound collective (4.8 KB over xGMI) hides under it."""
        L = _lib.lib()
        if self.pg is None:
            _lib.check(L.mqs_ba_gn_iteration_dev(self._h, float(lam), _sp()))          # one call: C issues everything
            if overlap is not None:
                overlap()                                                              # no host-side collective to hide it under
            return
        _lib.check(L.mqs_ba_gn_begin_dev(self._h, float(lam), _sp()))
        work = self.all_reduce(async_op=True)
        if overlap is not None:
            overlap()
        if work is not None:
            work.wait()
        _lib.check(L.mqs_ba_gn_finish_dev(self._h, float(lam), 1, _sp()))

    def gauss_newton_iterations(self, iters, lam=0.0, check=True):
        """`iters` iterations enqueued by one library call (single GPU or the C-level communicator).  check=True (default):
        waits for them and raises if one of their bounded waits gave up (`check`); check=False leaves the call asynchronous."""
        if self.pg is not None:
            for _ in range(iters):
                self.gauss_newton_iteration(lam)
        else:
            _lib.check(_lib.lib().mqs_ba_gn_iterations_dev(self._h, int(iters), float(lam), _sp()))
        if check:
            self.check()

    def total_cost(self, poses=None, points=None):
        """Host float: projection + point-prior cost summed over ranks, plus pose-prior cost."""
        torch = _torch()
        self.check()                                             # never a cost of an estimate a timed-out iteration left behind
        c = self.cost(poses, points).clone()
        if self.ccomm is not None:
            self.ccomm.all_reduce_sum_(c)
        elif self.pg is not None:
            import torch.distributed as dist
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.pg if self.pg is not True else None)
        total = float(c[0].item())
        if self.prior_mask is not None:
            total += self._pose_prior_cost(self.poses if poses is None else poses)
        return total

    def _pose_prior_cost(self, poses):
        torch = _torch()
        P = poses.cpu().numpy()
        P0 = self.prior_poses.cpu().numpy()
        S = self.prior_sigmas.cpu().numpy()
        M = self.prior_mask.cpu().numpy()
        cost = 0.0
        for c in range(self.C):
            if not M[c]:
                continue
            R0 = P0[c, :9].reshape(3, 3)
            R = P[c, :9].reshape(3, 3)
            Rr = R0.T @ R
            cth = min(1.0, max(-1.0, 0.5 * (np.trace(Rr) - 1)))
            th = np.arccos(cth)
            v = np.array([Rr[2, 1] - Rr[1, 2], Rr[0, 2] - Rr[2, 0], Rr[1, 0] - Rr[0, 1]])
            w = 0.5 * v if th < 1e-10 else v * th / (2 * np.sin(th))
            e = np.concatenate([w, R0.T @ (P[c, 9:] - P0[c, 9:])]) / S[c]
            cost += 0.5 * float(e.dot(e))
        return cost

    def optimize(self, iters=10, mode="gn", verbose=False, damping="gtsam"):
        """
        mode="gn": `iters` Gauss-Newton iterations (cost recorded before each and at the end).
        mode="lm": Levenberg-Marquardt with GTSAM 3.2.1's default schedule (lambda0 1e-5, factor
        10, relative/absolute error tolerance 1e-5, <= 100 iterations): bundle_adjust.cpp:323-324.
        damping: "gtsam" (default) adds lambda * I, GTSAM 3.2.1's default (diagonalDamping = false): the reference's iterate
        path; "marquardt" scales the diagonals by (1 + lambda) (invariant to the units of the variables; the default until
        round 3) -- the same optimum.  Returns the cost history (host floats).
        """
        sgn = {"marquardt": 1.0, "gtsam": -1.0}[damping]
        hist = []
        if mode == "gn":
            for _ in range(iters):
                hist.append(self.total_cost())
                self.gauss_newton_iteration(0.0)
            hist.append(self.total_cost())
        elif mode == "lm":
            lam = LM_LAMBDA_INITIAL
            cur = self.total_cost()
            hist.append(cur)
            for _ in range(min(iters, LM_MAX_ITERATIONS)):
                improved = False
                while lam <= LM_LAMBDA_UPPER:
                    self.linearize(sgn * lam)
                    self.all_reduce()
                    self.solve(sgn * lam)
                    self.backsub(sgn * lam)
                    new = self.total_cost(self.poses_new, self.points_new)
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
                if abs(cur - new) < LM_ABS_TOL or abs(cur - new) / max(cur, 1e-300) < LM_REL_TOL:
                    cur = new
                    break
                cur = new
        else:
            raise ValueError("mode must be 'gn' or 'lm'")
        self.cost_history = hist
        return hist

    # ---- benchmark helpers (bench.py) --------------------------------------------------
    def time_kernel(self, what, reps=20, lam=0.0):
        """Average launch duration (ms) of one kernel of the iteration, hipEvents on the current stream:
        what = "linearize" (the kernel alone), "finalize", "solve", "backsub", "solve_backsub" (the tail of an iteration as
        mqs_ba_gn_iteration_dev issues it: ONE launch for C <= 4).  The estimate is not advanced."""
        code = {"linearize": 0, "finalize": 1, "solve": 2, "backsub": 3, "solve_backsub": 4}[what]
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.lib().mqs_ba_time_dev(
            code, _p(self.poses), _p(self.calib), _p(self.sigma), self.C, _p(self.points), _p(self.obs), _p(self.mask),
            _p(self.prior_w), _p(self.prior_xyz), self.N, float(lam), _p(self.lin), _p(self.dpose), _p(self.poses_new),
            _p(self.points_new), _p(self.ws), self.ws.numel(), int(reps), _sp(), ctypes.byref(ms)))
        return float(ms.value)

    def benchmark_report(self, world, dist):
        torch = _torch()
        reps = 10

        def timed(fn):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        c0 = self.total_cost()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.gauss_newton_iteration(0.0)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        c1 = self.total_cost()
        C, N = self.C, self.N
        ms_lin = timed(lambda: self.linearize(0.0))
        ms_lin_k = self.time_kernel("linearize")
        ms_fin_k = self.time_kernel("finalize")
        self.linearize(0.0)
        ms_solve = self.time_kernel("solve")
        ms_back = self.time_kernel("backsub")
        ms_tail = self.time_kernel("solve_backsub")
        ms_ar = timed(self.all_reduce) if dist is not None else 0.0
        bytes_iter = N * (2 * (24 + 16 * C) + 24)
        return {
            "gn_iters_per_s": round(reps / dt, 1), "ms_per_iter": round(1e3 * dt / reps, 4),
            "landmarks_total": N * world, "cameras": C, "iterations_timed": reps,
            "kernels_ms": {"linearize_schur": round(ms_lin, 4), "linearize_kernel_only": round(ms_lin_k, 5),
                           "finalize_kernel_only": round(ms_fin_k, 5), "solve_retract": round(ms_solve, 4),
                           "backsub": round(ms_back, 4), "solve_retract_backsub_one_launch": round(ms_tail, 4),
                           "all_reduce": round(ms_ar, 4)},
            "launches_per_iteration": _launches_per_iteration(C),
            "algorithmic_GBps_per_gpu": round(bytes_iter / (1e-3 * (ms_lin + ms_back)) / 1e9, 1),
            "cost_before": c0, "cost_after_%d_more_iterations" % reps: c1,
            "timed_at": "the optimum: the problem has been iterated by the timed steps before this report (cost_before == cost_after); the "
                        "kernels' work does not depend on the values (no data-dependent branch on the hot path), and "
                        "`from_perturbed_start` below is the same count of iterations timed from SURVEY 8(d)'s perturbed start",
        }


def _launches_per_iteration(C):
    """Kernel launches of one mqs_ba_gn_iteration_dev on one GPU: lineariser + tail (finalize, solve, retraction and back-substitution in
    one launch) for C <= 4; the finalize as its own launch under MQS_BA_FINALIZE=kernel; four launches above four cameras."""
    import os
    if C > 4:
        return 4
    return 3 if os.environ.get("MQS_BA_FINALIZE", "") == "kernel" or os.environ.get("MQS_BA_TAIL", "") == "split" else 2


def time_iterations(ba, iters=200, warm=60):
    """ms per one-call Gauss-Newton iteration (hipEvents on the current stream around `iters` back-to-back iterations)."""
    torch = _torch()
    ba.gauss_newton_iterations(warm)                                # clock and caches
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ba.gauss_newton_iterations(iters, check=False)                  # the check synchronises: after the second event
    e1.record()
    e1.synchronize()
    ba.check()
    return e0.elapsed_time(e1) / iters


def shard_proxy_report(ba, iters=200):
    """Gauss-Newton iterations per second of a single-GPU problem sized like ONE rank's shard of a strong-scaling run
    (BASELINE configs[3]: 1e6 landmarks over 8 GPUs = 125 000 per rank), through the one-call iteration, with the kernels of
    an iteration timed one by one beside it: the serial floor of the sharded run before any collective costs anything."""
    ms = time_iterations(ba, iters)
    k = {"linearize_kernel_only": ba.time_kernel("linearize", reps=50), "finalize_kernel_only": ba.time_kernel("finalize", reps=50)}
    ba.linearize(0.0)
    k["solve_retract"] = ba.time_kernel("solve", reps=50)
    k["backsub"] = ba.time_kernel("backsub", reps=50)
    k["solve_retract_backsub_one_launch"] = ba.time_kernel("solve_backsub", reps=50)
    return {"landmarks": ba.N, "cameras": ba.C, "ms_per_iter": round(ms, 5), "gn_iters_per_s": round(1e3 / ms, 1),
            "launches_per_iteration": _launches_per_iteration(ba.C), "kernels_ms": {n: round(v, 5) for n, v in k.items()},
            "kernels_note": "stand-alone launches of the split entry points; inside the one-call iteration the finalize is the first "
                            "workgroups of the tail's launch (MQS_BA_FINALIZE=kernel: a launch of its own)",
            "sum_of_the_launches_ms": round(k["linearize_kernel_only"] + k["finalize_kernel_only"] + k["solve_retract_backsub_one_launch"], 5)}


def bundle_adjust(poses, calib, sigma, points, obs, mask=None, prior_w=None, prior_xyz=None, pose_prior=None,
                  iters=10, mode="gn", device=0):
    """
    numpy in / numpy out form of the optimiser (BA boundary (i)): uploads the problem, runs
    `iters` Gauss-Newton iterations (mode="gn") or Levenberg-Marquardt with GTSAM's default schedule (mode="lm",
    bundle_adjust.cpp:323-324) on the device and returns (poses (C,12), points (N,3), cost_history).
    Shapes as in BundleAdjuster; pose_prior = (prior_poses (C,12), prior_sigmas (C,6), prior_mask (C,)) or None.
    """
    torch = _torch()
    _lib.lib()                                                   # raises without the HIP library: no CPU path
    dev = torch.device("cuda", int(device))
    f64 = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    u8 = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(dev)
    pp = None
    if pose_prior is not None:
        pp = (f64(pose_prior[0]), f64(pose_prior[1]), u8(pose_prior[2]))
    with torch.cuda.device(dev):
        ba = BundleAdjuster(f64(poses), f64(calib), f64(np.asarray(sigma, dtype=np.float64).reshape(-1)), f64(points), f64(obs),
                            u8(mask), f64(prior_w), f64(prior_xyz), pp)
        hist = ba.optimize(iters=iters, mode=mode)
        return ba.poses.cpu().numpy(), ba.points.cpu().numpy(), hist


def pose_from_world_to_camera(P):
    """3x4 world->camera matrix [R_wc | t_wc] -> camera-to-world pose12 (IO.hpp:221-227 convention)."""
    Rwc, twc = P[:, :3], P[:, 3]
    return np.concatenate([Rwc.T.reshape(-1), -Rwc.T @ twc])


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K


def make_benchmark_problem(u, P, points_init, dev, seed=0, process_group=None, pixel_sigma=1.0, prior_first=4):
    """
    SURVEY.md 8(d) BA benchmark scene: observations = the triangulation benchmark's (noisy,
    pixel-discretised) measurements in pixels; initial poses = truth o Exp(N(0, diag(0.02 rad x3,
    0.1 x3))) (sigmas of GenerateData.hpp:108-109); Cal3DS2(480,480,0,320,240,0,0,0,0); sigma_pixel 1;
    gauge: pose prior on camera 0, point priors (sigma 0.2, GenerateData.hpp:123) on the first 4
    landmarks (`prior_first`; a rank that holds a later shard of ONE scene passes 0).  `points_init`: (N,3) initial
    landmarks (numpy or device tensor).
    """
    torch = _torch()
    from . import synthetic as syn
    C, N = u.shape[0], u.shape[1]
    rng = np.random.Generator(np.random.PCG64(seed))
    obs = u * syn.FOCAL + syn.CENTRE
    poses_true = np.stack([pose_from_world_to_camera(P[c]) for c in range(C)])
    poses = poses_true.copy()
    for c in range(C):
        xi = np.concatenate([0.02 * rng.standard_normal(3), 0.1 * rng.standard_normal(3)])
        R = poses_true[c, :9].reshape(3, 3)
        poses[c, :9] = (R @ _so3_exp(xi[:3])).reshape(-1)
        poses[c, 9:] = poses_true[c, 9:] + R @ xi[3:]
    calib = np.tile(np.array([syn.FOCAL, syn.FOCAL, 0.0, syn.CENTRE[0], syn.CENTRE[1], 0, 0, 0, 0]), (C, 1))
    if isinstance(points_init, np.ndarray):
        pts = torch.from_numpy(np.ascontiguousarray(points_init)).to(dev)
    else:
        pts = points_init.clone()
    prior_w = np.zeros(N)
    prior_w[:prior_first] = 1.0 / 0.2 ** 2
    prior_xyz = pts.clone()
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    pose_prior = (t(poses_true), t(np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1))),
                  t(np.array([1] + [0] * (C - 1), dtype=np.uint8), torch.uint8))
    return BundleAdjuster(t(poses), t(calib), t(np.full(C, pixel_sigma)), pts, t(obs), None, t(prior_w), prior_xyz,
                          pose_prior, process_group)
