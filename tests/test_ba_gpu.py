"""
GPU parity of the bundle-adjustment kernels (through the C ABI) against the BA oracle
(oracle/ba_np.py): reduced camera system, gradient, cost, back-substitution, solve + retraction,
the Gauss-Newton / LM loops, and full-size properties at 1e6 landmarks x 4 cameras.
Tolerance: 1e-5 relative (north_star: "BA residual parity <= 1e-5"); observed ~1e-12.
"""
import ctypes
import numpy as np
import pytest

from oracle import ba_np
from ba_util import make_scene

pytestmark = pytest.mark.gpu
TOL = 1e-5


def to_dev(sc):
    import torch
    t = lambda a, dt=None: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return dict(poses=t(sc["poses"]), calib=t(sc["calib"]), sigma=t(sc["sigma"]), points=t(sc["points"]),
                obs=t(sc["obs"]), mask=t(sc["mask"]), prior_w=t(sc["prior_w"]), prior_xyz=t(sc["prior_xyz"]))


def adjuster(gpu, sc, pose_prior=None):
    import torch
    d = to_dev(sc)
    pp = None
    if pose_prior is not None:
        pp = tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in pose_prior)
    return gpu.bundle_adjustment.BundleAdjuster(pose_prior=pp, **d)


def split_lin(lin, C):
    n6 = 6 * C
    lin = lin.cpu().numpy()
    return lin[:n6 * n6].reshape(n6, n6), lin[n6 * n6:n6 * n6 + n6], lin[-2], lin[-1]


CASES = [dict(N=300, C=2), dict(N=513, C=3, distortion=True), dict(N=300, C=4, distortion=True, masked_frac=0.3),
         dict(N=257, C=4, behind=9), dict(N=64, C=4), dict(N=1, C=2), dict(N=700, C=5), dict(N=300, C=6, distortion=True)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_linearize_and_backsub_parity(case, lam, gpu):
    import torch
    kw = dict(case)
    N, C = kw.pop("N"), kw.pop("C")
    sc = make_scene(N, C, seed=N + C, **kw)
    ba = adjuster(gpu, sc)
    S, g, cost, nv = split_lin(ba.linearize(lam), C)
    So, go, co, nvo, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                              sc["mask"], sc["prior_w"], sc["prior_xyz"], lam)
    assert np.abs(S - So).max() <= TOL * np.abs(So).max()
    assert np.abs(S - So).max() <= 1e-10 * np.abs(So).max()          # what fp64 delivers
    assert np.abs(g - go).max() <= 1e-10 * np.abs(go).max()
    assert cost == pytest.approx(co, rel=1e-12) and nv == nvo
    np.testing.assert_array_equal(S, S.T)                              # symmetrised exactly
    # back-substitution at the same linearisation point
    dpose = np.linalg.solve(So + 1e-3 * np.diag(np.diag(So)), go)
    ba.dpose.copy_(torch.from_numpy(dpose).cuda())
    pts_new = ba.backsub(lam).cpu().numpy()
    dp = ba_np.backsub(pieces, dpose)
    assert np.abs(pts_new - (sc["points"] + dp)).max() <= 1e-9 * max(1.0, np.abs(dp).max())
    # cost kernel agrees with the linearise kernel's cost
    c2 = ba.cost().cpu().numpy()
    assert c2[0] == pytest.approx(co, rel=1e-12) and c2[1] == nvo


def test_solve_and_retract(gpu):
    import torch
    sc = make_scene(400, 4, seed=2)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (4, 1)), np.array([1, 0, 1, 0], dtype=np.uint8))
    ba = adjuster(gpu, sc, pp)
    ba.linearize(0.0)
    ba.solve(0.0)
    S, g, _, _ = split_lin(ba.lin, 4)
    Hp, gp, cp = ba_np.pose_prior_terms(sc["poses"], *pp)
    d_ref = np.linalg.solve(S + Hp, g + gp)
    d = ba.dpose.cpu().numpy()
    assert np.abs(d - d_ref).max() <= 1e-9 * np.abs(d_ref).max()
    info = ba.info.cpu().numpy()
    assert info[0] == pytest.approx(cp, rel=1e-10) and info[1] == 0.0
    new = ba.poses_new.cpu().numpy()
    for c in range(4):
        np.testing.assert_allclose(new[c], ba_np.retract_pose(sc["poses"][c], d_ref[6 * c:6 * c + 6]), atol=1e-10)
        R = new[c, :9].reshape(3, 3)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
    # damping
    ba.solve(0.5)
    A = S + Hp
    d_lm = np.linalg.solve(A + 0.5 * np.diag(np.diag(A)), g + gp)
    assert np.abs(ba.dpose.cpu().numpy() - d_lm).max() <= 1e-9 * np.abs(d_lm).max()


@pytest.mark.parametrize("C", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("cond", [1e2, 1e8])
def test_reduced_solve_on_random_systems(C, cond, gpu):
    """mqs_ba_solve_dev on random symmetric positive definite systems of prescribed condition number, with and without
    damping and pose priors.  C <= 4 runs the one-wavefront solve that carries L^-1 and L^-1 b through the factorisation loop
    (ba_solve_small_kernel: no substitution chains), C > 4 the row-per-lane kernel with its LDS substitutions."""
    import torch
    n = 6 * C
    rng = np.random.default_rng(100 * C + int(np.log10(cond)))
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    S = (Q * np.geomspace(1.0, cond, n)) @ Q.T
    S = 0.5 * (S + S.T)
    g = rng.standard_normal(n) * np.sqrt(cond)
    L = gpu._lib.lib()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    poses = np.tile(np.concatenate([np.eye(3).reshape(-1), [0.0, 0.0, 0.0]]), (C, 1))
    poses += 0.0
    pd = torch.from_numpy(poses).cuda()
    prior_poses = torch.from_numpy(np.stack([ba_np.retract_pose(poses[c], 0.01 * rng.standard_normal(6)) for c in range(C)])).cuda()
    prior_sig = torch.from_numpy(np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1))).cuda()
    prior_mask = torch.from_numpy((np.arange(C) % 2 == 0).astype(np.uint8)).cuda()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for lam, with_prior in ((0.0, False), (1e-3, False), (-0.5, False), (0.0, True)):
        lin = torch.from_numpy(np.concatenate([S.reshape(-1), g, [0.0, 0.0]])).cuda()
        dpose = torch.zeros(n, dtype=torch.float64, device="cuda")
        out = torch.zeros((C, 12), dtype=torch.float64, device="cuda")
        info = torch.zeros(2, dtype=torch.float64, device="cuda")
        gpu._lib.check(L.mqs_ba_solve_dev(p(lin), C, p(pd), p(prior_poses) if with_prior else None, p(prior_sig) if with_prior else None,
                                          p(prior_mask) if with_prior else None, lam, p(dpose), p(out), p(info), sp))
        A, b = S.copy(), g.copy()
        if with_prior:
            Hp, gp, cp = ba_np.pose_prior_terms(poses, prior_poses.cpu().numpy(), prior_sig.cpu().numpy(), prior_mask.cpu().numpy())
            A, b = A + Hp, b + gp
            assert info.cpu().numpy()[0] == pytest.approx(cp, rel=1e-10)
        A = A + (lam * np.diag(np.diag(A)) if lam >= 0 else -lam * np.eye(n))
        ref = np.linalg.solve(A, b)
        d = dpose.cpu().numpy()
        # forward error of a Cholesky solve ~ cond * eps; the residual is the sharper check
        assert np.abs(d - ref).max() <= 50 * np.linalg.cond(A) * 2.2e-16 * np.abs(ref).max() + 1e-300
        assert np.abs(A @ d - b).max() <= 1e-9 * (np.abs(A).max() * np.abs(d).max() + np.abs(b).max())
        assert info.cpu().numpy()[1] == 0.0
        new = out.cpu().numpy()
        for c in range(C):
            np.testing.assert_allclose(new[c], ba_np.retract_pose(poses[c], d[6 * c:6 * c + 6]), atol=1e-9 * max(1.0, np.abs(d).max()))
    # not positive definite: flagged, no NaN escapes into the poses' rotation part being non-finite is acceptable, the flag is not optional
    lin = torch.from_numpy(np.concatenate([(-S).reshape(-1), g, [0.0, 0.0]])).cuda()
    gpu._lib.check(L.mqs_ba_solve_dev(p(lin), C, p(pd), None, None, None, 0.0, p(dpose), p(out), p(info), sp))
    assert info.cpu().numpy()[1] == 1.0


def test_gauss_newton_matches_oracle_every_iteration(gpu):
    """BA residual (cost) parity <= 1e-5 after each of the 10 Gauss-Newton iterations."""
    sc = make_scene(500, 4, seed=11)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (4, 1)), np.array([1, 0, 0, 0], dtype=np.uint8))
    ba = adjuster(gpu, sc, pp)
    hist = ba.optimize(iters=10, mode="gn")
    poses_o, points_o, hist_o = ba_np.gauss_newton(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                                   None, sc["prior_w"], sc["prior_xyz"], pp, iters=10)
    assert len(hist) == 11
    for a, b in zip(hist, hist_o):
        assert a == pytest.approx(b, rel=TOL)
    assert hist[-1] < 0.05 * hist[0]
    assert np.abs(ba.points.cpu().numpy() - points_o).max() < 1e-6
    assert np.abs(ba.poses.cpu().numpy() - poses_o).max() < 1e-7


def test_numpy_bundle_adjust_entry_point(gpu):
    """bundle_adjust(): numpy in / numpy out, same result as the oracle's Gauss-Newton loop; masked observations."""
    sc = make_scene(300, 3, seed=21, masked_frac=0.15, distortion=True)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (3, 1)), np.array([1, 0, 0], dtype=np.uint8))
    poses, points, hist = gpu.bundle_adjustment.bundle_adjust(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                                              sc["mask"], sc["prior_w"], sc["prior_xyz"], pp, iters=6, mode="gn")
    poses_o, points_o, hist_o = ba_np.gauss_newton(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                                   sc["mask"], sc["prior_w"], sc["prior_xyz"], pp, iters=6)
    assert poses.shape == (3, 12) and points.shape == (300, 3) and len(hist) == 7
    for a, b in zip(hist, hist_o):
        assert a == pytest.approx(b, rel=TOL)
    assert np.abs(points - points_o).max() < 1e-6 and np.abs(poses - poses_o).max() < 1e-7
    with pytest.raises(ValueError):
        gpu.bundle_adjustment.bundle_adjust(sc["poses"], sc["calib"], sc["sigma"], sc["points"][:10], sc["obs"])


def test_levenberg_marquardt_decreases_cost(gpu):
    sc = make_scene(400, 3, seed=5, distortion=True, pose_noise=(0.05, 0.4), point_noise=0.3)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (3, 1)), np.array([1, 0, 0], dtype=np.uint8))
    ba = adjuster(gpu, sc, pp)
    hist = ba.optimize(iters=30, mode="lm")
    assert all(b <= a for a, b in zip(hist, hist[1:]))
    assert hist[-1] < 0.01 * hist[0]
    assert hist[-1] / (400 * 3) < 1.5               # chi^2 per factor ~ noise level


def test_host_pointer_abi(gpu):
    sc = make_scene(300, 3, seed=8, masked_frac=0.2)
    lib, ctx = gpu._lib.lib(), gpu._lib.default_context()
    f64, u8 = gpu._lib.c_f64p, gpu._lib.c_u8p
    P = lambda a, t=f64: None if a is None else np.ascontiguousarray(a).ctypes.data_as(t)
    n6 = 18
    out = np.zeros(n6 * n6 + n6 + 2)
    args = [np.ascontiguousarray(sc[k]) for k in ("poses", "calib", "sigma", "points", "obs", "mask", "prior_w", "prior_xyz")]
    gpu._lib.check(lib.mqs_ba_linearize(ctx.handle, P(args[0]), P(args[1]), P(args[2]), 3, P(args[3]), P(args[4]),
                                        P(args[5], u8), P(args[6]), P(args[7]), 300, 0.0, P(out)))
    So, go, co, nvo, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                              sc["mask"], sc["prior_w"], sc["prior_xyz"])
    assert np.abs(out[:n6 * n6].reshape(n6, n6) - So).max() <= 1e-10 * np.abs(So).max()
    dpose = np.linalg.solve(So + 1e-3 * np.eye(n6), go)
    pout = np.zeros((300, 3))
    gpu._lib.check(lib.mqs_ba_backsub(ctx.handle, P(args[0]), P(args[1]), P(args[2]), 3, P(args[3]), P(args[4]),
                                      P(args[5], u8), P(args[6]), P(args[7]), 300, 0.0, P(dpose), P(pout)))
    assert np.abs(pout - (sc["points"] + ba_np.backsub(pieces, dpose))).max() < 1e-9


def test_bad_arguments_fail_loudly(gpu):
    lib = gpu._lib.lib()
    rc = lib.mqs_ba_linearize_dev(None, None, None, 4, None, None, None, None, None, 10, 0.0, None, None, 0, None)
    assert rc < 0 and b"null" in lib.mqs_last_error()
    rc = lib.mqs_ba_solve_dev(None, 9, None, None, None, None, 0.0, None, None, None, None)
    assert rc < 0


def test_full_size_properties_1e6x4(gpu, c_oracle):
    """BASELINE configs[3] shape on one GPU: shard additivity (the multi-GPU contract: the sum of the
    shards' systems equals the whole, <= 1e-10), bitwise determinism, oracle agreement on a sample,
    and monotone cost over 10 GN iterations."""
    import torch
    N, C = 1_000_000, 4
    syn = gpu.synthetic
    u, P, pts = syn.triangulation_problem(N, C)
    rng = np.random.default_rng(0)
    init = pts + 0.05 * rng.standard_normal(pts.shape)
    ba = gpu.bundle_adjustment.make_benchmark_problem(u, P, init, torch.device("cuda", 0), seed=1)
    lin = ba.linearize(0.0).clone()
    lin2 = ba.linearize(0.0).clone()
    assert torch.equal(lin, lin2)                                   # reproducible: no atomics
    h = N // 2 + 129
    BA = gpu.bundle_adjustment.BundleAdjuster
    parts = []
    for sl in (slice(0, h), slice(h, N)):
        sub = BA(ba.poses, ba.calib, ba.sigma, ba.points[sl].clone(), ba.obs[:, sl].clone(), None,
                 ba.prior_w[sl].clone(), ba.prior_xyz[sl].clone())
        parts.append(sub.linearize(0.0).clone())
    tot = (parts[0] + parts[1]).cpu().numpy()
    ref = lin.cpu().numpy()
    assert np.abs(tot - ref).max() <= 1e-10 * np.abs(ref).max()
    # oracle on the first 3000 landmarks
    n = 3000
    sub = BA(ba.poses, ba.calib, ba.sigma, ba.points[:n].clone(), ba.obs[:, :n].clone(), None,
             ba.prior_w[:n].clone(), ba.prior_xyz[:n].clone())
    S, g, cost, nv = split_lin(sub.linearize(0.0), C)
    So, go, co, nvo, _ = ba_np.linearize(ba.poses.cpu().numpy(), ba.calib.cpu().numpy(), ba.sigma.cpu().numpy(),
                                         ba.points[:n].cpu().numpy(), ba.obs[:, :n].cpu().numpy(), None,
                                         ba.prior_w[:n].cpu().numpy(), ba.prior_xyz[:n].cpu().numpy())
    assert np.abs(S - So).max() <= 1e-10 * np.abs(So).max() and cost == pytest.approx(co, rel=1e-12)
    # the WHOLE 1e6 x 4 reduced system, gradient, cost, count and a back-substitution against the oracle's C port (OpenMP)
    hh = lambda t: t.cpu().numpy()
    Sw, gw, cw, nw = split_lin(lin, C)
    Sc, gc, cc, nc = c_oracle.ba_linearize(hh(ba.poses), hh(ba.calib), hh(ba.sigma), hh(ba.points), hh(ba.obs), None,
                                           hh(ba.prior_w), hh(ba.prior_xyz), 0.0, use_omp=True)
    assert np.abs(Sw - Sc).max() <= 1e-10 * np.abs(Sc).max() and np.abs(gw - gc).max() <= 1e-10 * np.abs(gc).max()
    assert cw == pytest.approx(cc, rel=1e-11) and nw == nc
    dpose = np.linalg.solve(Sc + 1e-3 * np.diag(np.diag(Sc)), gc)
    ba.dpose.copy_(torch.from_numpy(dpose).cuda())
    pts_new = ba.backsub(0.0).cpu().numpy()
    pts_o = c_oracle.ba_backsub(hh(ba.poses), hh(ba.calib), hh(ba.sigma), hh(ba.points), hh(ba.obs), dpose, None,
                                hh(ba.prior_w), hh(ba.prior_xyz), 0.0, use_omp=True)
    assert np.abs(pts_new - pts_o).max() <= 1e-9 * max(1.0, np.abs(pts_o - hh(ba.points)).max())
    hist = ba.optimize(iters=10, mode="gn")
    assert hist[-1] < hist[0] and all(b <= a * (1 + 1e-9) for a, b in zip(hist[1:], hist[2:]))
    assert hist[-1] / (N * C) < 1.0                                  # chi^2 per factor at pixel-noise level


def test_one_call_iteration_equals_the_four_launches(gpu):
    """mqs_ba_gn_iteration_dev (one library call per Gauss-Newton iteration) == linearise, solve, back-substitute issued one
    by one, bit for bit; mqs_ba_gn_iterations_dev == the same call repeated; the buffer pair alternates."""
    import torch
    sc = make_scene(777, 4, seed=3, distortion=True)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (4, 1)), np.array([1, 0, 0, 0], dtype=np.uint8))
    a, b, c = adjuster(gpu, sc, pp), adjuster(gpu, sc, pp), adjuster(gpu, sc, pp)
    for it in range(3):
        a.gauss_newton_iteration(0.0)
        assert a._cur == (it + 1) % 2
        b.linearize(0.0)
        b.solve(0.0)
        b.backsub(0.0)
        b.accept()
    c.gauss_newton_iterations(3)
    torch.cuda.synchronize()
    assert torch.equal(a.poses, b.poses) and torch.equal(a.points, b.points)
    assert torch.equal(a.poses, c.poses) and torch.equal(a.points, c.points)
    assert torch.equal(a.lin, b.lin)
    # the split form with a trial step: nothing becomes current until accepted (Levenberg-Marquardt)
    L = gpu._lib.lib()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    before = a.poses.clone()
    gpu._lib.check(L.mqs_ba_gn_begin_dev(a._h, 1e-3, sp))
    gpu._lib.check(L.mqs_ba_gn_finish_dev(a._h, 1e-3, 0, sp))
    torch.cuda.synchronize()
    assert torch.equal(a.poses, before) and not torch.equal(a.poses_new, before)


@pytest.mark.parametrize("case", [dict(N=70_000, C=4), dict(N=150_000, C=4, distortion=True, masked_frac=0.2), dict(N=131_000, C=3, distortion=True),
                                  dict(N=262_207, C=2), dict(N=10_000, C=4)])
def test_a_run_of_iterations_in_one_launch_each_equals_the_two_launch_iterations(case, gpu):
    """mqs_ba_gn_iterations_dev on a shard-sized problem: lineariser, K - 1 launches that each finish one iteration and linearise
    the next (ba_iterate_kernel), tail -- against the same K iterations issued one by one (two launches each): bit-identical
    poses, landmarks, reduced system and dpose, with priors, masks and distortion, for 2, 3 and 4 cameras, chunks of 1 to 4 rows per
    wave, and for a problem too small for the one-launch form (10 000 landmarks: fewer workgroups than finalizer pieces), where
    the call falls back to the two-launch iteration."""
    import torch
    kw = dict(case)
    N, C = kw.pop("N"), kw.pop("C")
    sc = make_scene(N, C, seed=N % 97 + C, **kw)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1)), np.array([1] + [0] * (C - 1), dtype=np.uint8))
    a, b = adjuster(gpu, sc, pp), adjuster(gpu, sc, pp)
    K = 5
    for _ in range(K):
        a.gauss_newton_iteration(0.0)
    b.gauss_newton_iterations(K)
    torch.cuda.synchronize()
    assert a._cur == b._cur == K % 2
    assert torch.equal(a.poses, b.poses) and torch.equal(a.points, b.points)
    assert torch.equal(a.lin, b.lin) and torch.equal(a.dpose, b.dpose) and torch.equal(a.info, b.info)
    assert b.total_cost() < 1.001 * a.total_cost() and np.isfinite(b.total_cost())
    # an even count and a damped run continue from there
    a.gauss_newton_iteration(-1e-3); a.gauss_newton_iteration(-1e-3)
    b.gauss_newton_iterations(2, lam=-1e-3)
    torch.cuda.synchronize()
    assert torch.equal(a.poses, b.poses) and torch.equal(a.points, b.points)


def test_c_abi_communicator_single_rank(gpu):
    """The library's RCCL communicator (mqs_comm_*) with one rank: the all-reduce is the identity, and a BundleAdjuster bound
    to it runs the one-call iteration with the collective issued from C (the N-rank case needs N GPUs: bench.py --gpus N)."""
    import torch
    cc = gpu.sharding.init_c_comm(0, 1, 0)
    try:
        assert gpu._lib.lib().mqs_comm_world_size(cc.ctx.handle) == 1
        t = torch.arange(602, dtype=torch.float64, device="cuda")
        cc.all_reduce_sum_(t)
        torch.cuda.synchronize()
        assert torch.equal(t, torch.arange(602, dtype=torch.float64, device="cuda"))
        with pytest.raises(ValueError):
            cc.all_reduce_sum_(t.float())
        sc = make_scene(300, 3, seed=8)
        d = to_dev(sc)
        ba = gpu.bundle_adjustment.BundleAdjuster(process_group=cc, **d)
        ref = adjuster(gpu, sc)
        ba.gauss_newton_iterations(2)
        ref.gauss_newton_iterations(2)
        torch.cuda.synchronize()
        assert torch.equal(ba.poses, ref.poses) and torch.equal(ba.points, ref.points)
        assert ba.total_cost() == ref.total_cost()
    finally:
        cc.close()


def test_gtsam_style_damping(gpu):
    """lambda < 0 in the C ABI = Levenberg damping |lambda| * I on the landmark blocks and on the reduced system -- GTSAM
    3.2.1's default (diagonalDamping = false).  The damped system equals the oracle's un-eliminated one with lambda * I
    added to every variable, and LM with either damping lands on the same optimum."""
    sc = make_scene(120, 3, seed=17, distortion=True)
    lam = 0.37
    ba = adjuster(gpu, sc)
    S, g, cost, nv = split_lin(ba.linearize(-lam), 3)
    So, go, co, nvo, _ = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"], None,
                                         sc["prior_w"], sc["prior_xyz"], -lam)       # the oracle's Hll + lam * I, eliminated
    assert np.abs(S - So).max() <= 1e-10 * np.abs(So).max()
    assert np.abs(g - go).max() <= 1e-10 * np.abs(go).max() and cost == pytest.approx(co, rel=1e-12)
    Sm, _, _, _, _ = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"], None,
                                     sc["prior_w"], sc["prior_xyz"], lam)
    assert np.abs(Sm - So).max() > 1e-3 * np.abs(So).max()                            # and it is not the Marquardt scaling
    # the solve adds lambda to the diagonal of S (not a scaling)
    ba.solve(-lam)
    dp = ba.dpose.cpu().numpy()
    np.testing.assert_allclose(dp, np.linalg.solve(S + lam * np.eye(18), g), rtol=1e-8, atol=1e-12)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (3, 1)), np.array([1, 0, 0], dtype=np.uint8))
    a, b = adjuster(gpu, sc, pp), adjuster(gpu, sc, pp)
    ha = a.optimize(iters=40, mode="lm", damping="marquardt")
    hb = b.optimize(iters=40, mode="lm", damping="gtsam")
    # GTSAM's stop rule (relative decrease < 1e-5) leaves each run within ~1e-3 of the optimum's cost
    assert hb[-1] == pytest.approx(ha[-1], rel=3e-3), (ha[-1], hb[-1])
    assert all(y <= x for x, y in zip(hb, hb[1:])) and hb[-1] < 0.05 * hb[0]
    assert np.abs(a.poses.cpu().numpy() - b.poses.cpu().numpy()).max() < 2e-2


def distort_observations(obs_px, calib):
    """Pixels of an ideal pinhole camera -> pixels of the same camera with the Cal3DS2 lens model in `calib`
    (IO.hpp:230-236 loads fx fy s u0 v0 k1 k2 p1 p2), so that a distorted calibration sees consistent measurements."""
    out = np.empty_like(obs_px)
    for c in range(obs_px.shape[0]):
        fx, fy, s, u0, v0, k1, k2, p1, p2 = calib[c]
        y = (obs_px[c, :, 1] - v0) / fy
        x = (obs_px[c, :, 0] - u0 - s * y) / fx
        r2 = x * x + y * y
        g = 1 + k1 * r2 + k2 * r2 * r2
        xd = g * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = g * y + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
        out[c, :, 0] = fx * xd + s * yd + u0
        out[c, :, 1] = fy * yd + v0
    return out


# k1 = 0.3: the lens of triangulation_comparison.py:127-147; the second set exercises every coefficient and the skew
DIST_K1 = [0.0, 0.3, 0.0, 0.0, 0.0]
DIST_ALL = [0.7, 0.08, -0.02, 0.001, -0.0015]
WAVE_CASES = [dict(N=180_001, C=4), dict(N=150_000, C=3, masked=True), dict(N=70_000, C=2),
              dict(N=262_144 + 63, C=4, masked=True, lam=1e-3)]
# the same walks with lens distortion: wl_chunk<C, 1..4, NODIST = false> (csrc/ba.hip), masked and unmasked, C = 2, 3, 4
WAVE_CASES += [dict(N=n, C=c, masked=m, dist=d) for n, c, m, d in [
    (70_000, 4, False, DIST_K1), (70_000, 2, True, DIST_ALL), (150_000, 3, False, DIST_ALL), (150_000, 4, True, DIST_K1),
    (180_001, 4, False, DIST_ALL), (180_001, 2, False, DIST_K1), (180_001, 3, True, DIST_K1),
    (262_144 + 63, 4, True, DIST_ALL), (262_144 + 63, 3, False, DIST_K1), (262_144 + 63, 2, True, DIST_ALL)]]
# from 400 000 landmarks on the launcher takes ba_linearize_wave_kernel<C, SCALAR = true> (camera blocks through scalar loads): a wave
# owns 6-7 rows at 400 k (chunks of 4 + 2 and 4 + 3) and 9-10 at 600 k (4 + 4 + 1, 4 + 4 + 2): every chunk size of that form too,
# C = 2, 3, 4, with and without lens distortion and mask (test_full_size_properties_1e6x4 is its C = 4 no-distortion run)
WAVE_CASES += [dict(N=n, C=c, masked=m, dist=d) for n, c, m, d in [
    (400_001, 2, True, DIST_ALL), (410_000, 3, False, DIST_K1), (420_003, 4, True, DIST_ALL), (600_001, 4, False, None),
    (600_001, 3, True, None), (610_000, 2, False, DIST_K1)]]


@pytest.mark.parametrize("case", WAVE_CASES, ids=lambda c: "N%d-C%d%s%s" % (c["N"], c["C"], "-masked" if c.get("masked") else "",
                                                                             "-dist" if c.get("dist") else ""))
def test_wave_lineariser_every_chunk_size_against_the_c_oracle(case, gpu, c_oracle):
    """Sizes at which a wave of the wave-level lineariser owns 1 to 5 rows of 64 landmarks, i.e. walks chunks of 1, 2, 3 and 4
    landmarks per lane (the 1e6 test only sees chunks of 4 and 3; the small cases only chunks of 1): reduced system, gradient,
    cost and count against the oracle's C port on the whole problem, priors on scattered landmarks, a mask with unseen
    factors and NaN measurements behind it, and the round-1 kernel (MQS_BA_LINEARIZER=lane is a process-wide switch, so it
    is compared through shard additivity instead: the sum of two shards' systems equals the whole)."""
    import torch
    N, C = case["N"], case["C"]
    syn = gpu.synthetic
    u, P, pts = syn.triangulation_problem(N, C)
    rng = np.random.default_rng(N)
    ba = gpu.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.03 * rng.standard_normal(pts.shape), torch.device("cuda", 0), seed=2)
    if case.get("dist"):
        cal = ba.calib.cpu().numpy()
        cal[:, 2] = case["dist"][0]
        cal[:, 5:] = case["dist"][1:]
        cal[C - 1, 5:] *= 0.5                                   # cameras need not share a lens
        ba.obs.copy_(torch.from_numpy(distort_observations(ba.obs.cpu().numpy(), cal)))
        ba.calib.copy_(torch.from_numpy(cal))
    pw = np.zeros(N)
    sel = rng.choice(N, 5000, replace=False)
    pw[sel] = rng.uniform(1.0, 30.0, 5000)
    ba.prior_w.copy_(torch.from_numpy(pw))
    mask = None
    if case.get("masked"):
        mask = (rng.random((C, N)) > 0.2).astype(np.uint8)
        mask[:2] = 1
        obs = ba.obs.cpu().numpy()
        obs[mask == 0] = np.nan                                # a masked slot may hold anything
        ba.obs.copy_(torch.from_numpy(obs))
        BA = gpu.bundle_adjustment.BundleAdjuster
        ba = BA(ba.poses, ba.calib, ba.sigma, ba.points, ba.obs, torch.from_numpy(mask).cuda(), ba.prior_w, ba.prior_xyz)
    lam = case.get("lam", 0.0)
    S, g, cost, nv = split_lin(ba.linearize(lam), C)
    h = lambda t: None if t is None else t.cpu().numpy()
    obs_h = np.nan_to_num(h(ba.obs))                           # the oracle multiplies masked slots by 0
    So, go, co, nvo = c_oracle.ba_linearize(h(ba.poses), h(ba.calib), h(ba.sigma), h(ba.points), obs_h, mask, h(ba.prior_w),
                                            h(ba.prior_xyz), lam, use_omp=True)
    assert np.abs(S - So).max() <= 1e-10 * np.abs(So).max()
    assert np.abs(g - go).max() <= 1e-10 * np.abs(go).max()
    assert cost == pytest.approx(co, rel=1e-11) and nv == nvo
    np.testing.assert_array_equal(S, S.T)
    lin2 = ba.linearize(lam).clone()
    assert torch.equal(lin2, ba.linearize(lam))                  # bitwise reproducible
    # shard additivity at an uneven cut (the shards walk different chunk patterns than the whole)
    cut = N // 3 + 17
    BA = gpu.bundle_adjustment.BundleAdjuster
    parts = []
    for sl in (slice(0, cut), slice(cut, N)):
        m = None if ba.mask is None else ba.mask[:, sl].clone()
        sub = BA(ba.poses, ba.calib, ba.sigma, ba.points[sl].clone(), ba.obs[:, sl].clone(), m, ba.prior_w[sl].clone(),
                 ba.prior_xyz[sl].clone())
        parts.append(sub.linearize(lam).clone())
    tot = (parts[0] + parts[1]).cpu().numpy()
    assert np.abs(tot - lin2.cpu().numpy()).max() <= 1e-10 * np.abs(lin2.cpu().numpy()).max()
    # back-substitution of the same problem (its distortion branch is per kernel too) against the oracle's C port
    dpose = np.linalg.solve(So + 1e-3 * np.diag(np.diag(So)), go)
    ba.dpose.copy_(torch.from_numpy(dpose).cuda())
    pts_new = ba.backsub(lam).cpu().numpy()
    pts_o = c_oracle.ba_backsub(h(ba.poses), h(ba.calib), h(ba.sigma), h(ba.points), obs_h, dpose, mask, h(ba.prior_w),
                                h(ba.prior_xyz), lam, use_omp=True)
    assert np.abs(pts_new - pts_o).max() <= 1e-9 * max(1.0, np.abs(pts_o - h(ba.points)).max())


@pytest.mark.gpu
def test_fused_finalize_hand_over_under_changing_linearisation_points(gpu):
    """The flag-after-data hand-over inside the fused tail (the finalizer workgroups' piece sums, published with relaxed stores,
    an explicit store-acknowledge wait and an epoch flag): 300 one-call iterations, each from a different linearisation point so
    that a piece read before it landed -- the previous iteration's -- changes the result, fingerprinted and compared with the
    same run with the finalize as a launch of its own (MQS_BA_FINALIZE=kernel, read once per process: two processes)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "probes", "fused_finalize_stress.py")
    outs = []
    for mode in (None, "kernel"):
        env = {k: v for k, v in os.environ.items() if k != "MQS_BA_FINALIZE"}
        if mode:
            env["MQS_BA_FINALIZE"] = mode
        r = subprocess.run([sys.executable, script, "300", "90000"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["mode"] == "fused" and outs[1]["mode"] == "kernel"
    assert outs[0]["sha256_16"] == outs[1]["sha256_16"]


@pytest.mark.gpu
def test_second_form_of_the_tail_gives_the_same_bits(gpu):
    """MQS_BA_TAIL_FORM=2 (one twelve-wave workgroup per CU, the system solved once per CU, the next batch's loads in flight under
    the current one's arithmetic: built in round 4, measured slower, kept as an A/B form) against the default form: the same
    fingerprint over 60 one-call iterations from 60 linearisation points (the form is read once per process: two processes)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "probes", "fused_finalize_stress.py")
    outs = []
    for form in (None, "2"):
        env = {k: v for k, v in os.environ.items() if k not in ("MQS_BA_TAIL_FORM", "MQS_BA_FINALIZE")}
        if form:
            env["MQS_BA_TAIL_FORM"] = form
        r = subprocess.run([sys.executable, script, "60", "270000"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["sha256_16"] == outs[1]["sha256_16"]


@pytest.mark.gpu
def test_scalar_camera_blocks_follow_the_poses(gpu):
    """The wave lineariser's scalar-load form (N >= 400 000) reads the camera blocks from a copy every workgroup publishes to the
    workspace and pulls through the scalar cache: the same adjuster (same workspace, same addresses) linearised at 40 different
    pose sets in a row must see each of them -- a stale line in the scalar cache or the L2 would be the PREVIOUS poses.  Checked
    against the sum of two shards, which are small enough to take the LDS form and have workspaces of their own."""
    import torch
    N, C = 430_000, 4
    u, P, pts = gpu.synthetic.triangulation_problem(N, C)
    ba = gpu.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.02, torch.device("cuda", 0), seed=3)
    BA = gpu.bundle_adjustment.BundleAdjuster
    cut = N // 2 + 33
    subs = [BA(ba.poses, ba.calib, ba.sigma, ba.points[sl].clone(), ba.obs[:, sl].clone(), None, ba.prior_w[sl].clone(),
               ba.prior_xyz[sl].clone()) for sl in (slice(0, cut), slice(cut, N))]
    poses0 = ba.poses.clone()
    g = torch.Generator(device="cuda").manual_seed(11)
    worst = 0.0
    for k in range(40):
        noise = torch.zeros_like(poses0)
        noise[:, 9:] = 0.05 * torch.randn((C, 3), generator=g, device="cuda", dtype=torch.float64)
        for b in [ba] + subs:
            b.poses.copy_(poses0 + noise)
        whole = ba.linearize(0.0).clone()
        parts = subs[0].linearize(0.0).clone() + subs[1].linearize(0.0)
        worst = max(worst, float((whole - parts).abs().max() / whole.abs().max()))
    assert worst <= 1e-10, worst



@pytest.mark.gpu
def test_a_finalizer_flag_that_never_comes_is_an_error_not_an_answer(gpu):
    """The fused tail waits (bounded, 2 s) for the finalizer workgroups of its own launch.  A wait that gives up must not fold
    whatever sits in the piece sums: with one flag withheld (the library's test hook) the iteration publishes nothing -- the
    poses and landmarks of the next estimate keep the poison they were filled with, info[1] = 2 -- and the host gets a
    RuntimeError from the calls that hand results over (`gauss_newton_iterations`, `total_cost`, `check`) and from the next
    iteration's entry; a healthy problem next to it is not affected."""
    import torch
    L = gpu._lib.lib()
    sc = make_scene(70_000, 4, seed=5)
    ba = adjuster(gpu, sc)
    ok = adjuster(gpu, sc)
    ba.gauss_newton_iterations(2)                                     # healthy: no error, estimate advances
    poison = 12345.678
    ba.poses_new.fill_(poison)
    ba.points_new.fill_(poison)
    assert L.mqs_debug_ba_withhold_flag(17) == 0
    try:
        with pytest.raises(RuntimeError, match="gave up waiting"):
            ba.gauss_newton_iterations(1)
    finally:
        assert L.mqs_debug_ba_withhold_flag(-1) == 0
    # nothing was published by the iteration that timed out (its outputs are the `current` buffers now: the handle flipped)
    assert bool((ba.poses == poison).all()) and bool((ba.points == poison).all())
    assert float(ba.info[1].item()) == 2.0
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ba.total_cost()
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ba.gauss_newton_iteration()                                   # sticky: the entry point refuses to build on it
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ba.check()
    ok.gauss_newton_iterations(2)                                     # the hook is off again and other problems never saw it
    assert np.isfinite(ok.total_cost())
