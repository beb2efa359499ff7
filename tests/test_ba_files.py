"""
B5 + 8(f) rank 1: the reference's on-disk BA format and the general sparse solver.
CPU: loaders / validators / writers on the reference's committed data files (tests/golden/ba_example,
tests/golden/ba_svo are copies of those DATA files).  GPU: sparse kernels against the sparse oracle, the
dense-vs-sparse equivalence, and the committed converged outputs as loose anchors.
"""
import os
import numpy as np
import pytest

from oracle import ba_np

HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(HERE, "golden", "ba_example")
SVO = os.path.join(HERE, "golden", "ba_svo")


def load(mqs, d, name, nc, fps):
    io = mqs.ba_io
    fn = io.create_filenames(d, name, nc)
    data = io.load_data(fn, fps)
    io.validate_data_integrity(data, nc)
    return fn, data


def test_example_files_load_and_validate(mqs):
    io = mqs.ba_io
    fn, data = load(mqs, EX, "synthetic", 2, 1)
    assert len(data.points3D) == 8 and len(data.point3DAddedIdxs) == 20 and len(data.calibrations) == 2
    assert np.abs(np.abs(data.points3D) - 10.0).max() < 0.5                 # noisy corners of the 20-unit cube
    np.testing.assert_array_equal(np.abs(data.points3D[:4]), 10.0)         # step-0 points are exact
    ok, log = io.validate_sufficiently_constrained(data, True)
    assert ok and log[-1][1] == 3 * 8 + 6 * 40
    pr = io.build_sparse_problem(data, use_odometry=True)
    assert pr.poses.shape == (40, 12) and len(pr.obs_pose) == 320 and len(pr.odo_from) == 58
    assert (pr.prior_w > 0).sum() == 4 and pr.prior_w.max() == pytest.approx(2 / 0.2 ** 2)   # seen by both cameras
    for p in pr.poses:
        R = p[:9].reshape(3, 3)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
    np.testing.assert_allclose(pr.calib[0], [500, 500, 0, 320, 240, 0, 0, 0, 0])
    assert fn.map_out.endswith("map_out-synthetic-BA.pcd")


def test_svo_files_load(mqs):
    io = mqs.ba_io
    fn, data = load(mqs, SVO, "slam2", 1, 50)
    pr = io.build_sparse_problem(data)
    assert pr.poses.shape == (186, 12) and pr.points.shape == (1046, 3) and len(pr.obs_pose) == 7494
    ok, _ = io.validate_sufficiently_constrained(data, False)
    assert ok


def test_validators_reject_bad_data(mqs):
    io = mqs.ba_io
    fn, data = load(mqs, EX, "synthetic", 2, 1)
    data.point2D3DAssocs[0][3].append((5, 0, 0))            # looks into the future (frame 5 at step 3)
    with pytest.raises(ValueError):
        io.validate_data_integrity(data, 2)
    fn, data = load(mqs, EX, "synthetic", 2, 1)
    data.point3DAddedIdxs[2].append(0)                      # point 0 added twice
    with pytest.raises(ValueError):
        io.validate_sufficiently_constrained(data, False)


def test_write_read_round_trip(mqs, tmp_path):
    io = mqs.ba_io
    fn, data = load(mqs, EX, "synthetic", 2, 1)
    out = io.create_filenames(str(tmp_path), "synthetic", 2)
    io.save_result(out._replace(map_out=out.map_in, trajectories_out=out.trajectories_in), data)
    pts = io.load_map(out.map_in)
    np.testing.assert_allclose(pts, data.points3D, rtol=1e-15)
    tr = io.load_trajectory(out.trajectories_in[1])
    assert len(tr) == 20
    for (t0, p0), (t1, p1) in zip(tr, data.poses[1]):
        assert t0 == t1
        np.testing.assert_allclose(p0, p1, atol=1e-12)
    q = io.R_to_quat(io.quat_to_R(0.1, -0.2, 0.3, 0.9))
    np.testing.assert_allclose(q, np.array([0.1, -0.2, 0.3, 0.9]) / np.linalg.norm([0.1, -0.2, 0.3, 0.9]), atol=1e-15)


def test_pair_grouping_host_logic(mqs):
    """build_pairs / group_pairs: every (a <= b) pair of a landmark's observations once, sorted by pose pair, groups tile the
    list, the order inside a group is the landmark order (what makes the device sum reproducible)."""
    sb = mqs.sparse_ba
    obs_ptr = np.array([0, 3, 3, 5, 6], dtype=np.int64)                 # landmarks with 3, 0, 2, 1 observations
    obs_pose = np.array([0, 1, 4, 1, 4, 1], dtype=np.int32)            # sorted by pose inside each landmark
    pa, pb = sb.build_pairs(obs_ptr)
    assert len(pa) == 6 + 0 + 3 + 1 and (pa <= pb).all()
    ga, gb, gp = sb.group_pairs(pa, pb, obs_pose, 5)
    keys = obs_pose[ga].astype(np.int64) * 5 + obs_pose[gb]
    assert (np.diff(keys) >= 0).all() and gp[0] == 0 and gp[-1] == len(pa)
    assert [int(keys[a]) for a in gp[:-1]] == sorted(set(keys.tolist()))
    assert sorted(zip(ga.tolist(), gb.tolist())) == sorted(zip(pa.tolist(), pb.tolist()))
    k14 = np.nonzero(keys == 1 * 5 + 4)[0]                               # pose pair (1, 4): landmark 0 first, then landmark 2
    assert list(zip(ga[k14], gb[k14])) == [(1, 2), (3, 4)]
    e = sb.group_pairs(np.zeros(0, np.int64), np.zeros(0, np.int64), obs_pose, 5)
    assert len(e[0]) == 0 and e[2].tolist() == [0]
    # observations in ANY order inside a landmark: the pairs are oriented so that pose(a) <= pose(b) before grouping, so that
    # a pose pair has exactly one group (one writer of its block)
    unsorted_pose = np.array([4, 0, 1, 4, 1, 1], dtype=np.int32)
    ga, gb, gp = sb.group_pairs(pa, pb, unsorted_pose, 5)
    assert (unsorted_pose[ga] <= unsorted_pose[gb]).all()
    keys = unsorted_pose[ga].astype(np.int64) * 5 + unsorted_pose[gb]
    assert (np.diff(keys) >= 0).all() and [int(keys[a]) for a in gp[:-1]] == sorted(set(keys.tolist()))
    assert sorted(zip(np.minimum(ga, gb).tolist(), np.maximum(ga, gb).tolist())) == sorted(zip(pa.tolist(), pb.tolist()))


def _lin_oracle(pr, lam=0.0):
    S, g, cost, nv, pieces = ba_np.sparse_linearize(pr.poses, pr.pose_cam, pr.calib, pr.sigma, pr.points, pr.obs_ptr,
                                                    pr.obs_pose, pr.obs_uv, pr.prior_w, pr.prior_xyz, lam)
    Hp, gp, cp = ba_np.sparse_pose_prior_terms(pr.poses, pr.pose_prior_idx, pr.poses[pr.pose_prior_idx], pr.pose_prior_sigmas)
    return S + Hp, g + gp, cost, nv, pieces, cp


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["example", "svo"])
@pytest.mark.parametrize("lam", [0.0, 1e-3])
def test_sparse_linearize_solve_backsub_parity(which, lam, gpu):
    d, name, nc, fps = (EX, "synthetic", 2, 1) if which == "example" else (SVO, "slam2", 1, 50)
    fn, data = load(gpu, d, name, nc, fps)
    pr0 = gpu.ba_io.build_sparse_problem(data)
    rng = np.random.default_rng(0)                                    # move off the files' 6-digit optimum
    pr0 = pr0._replace(points=pr0.points + 0.01 * rng.standard_normal(pr0.points.shape))
    ba = gpu.sparse_ba.SparseBundleAdjuster(pr0)
    pr = ba.problem
    S, g = ba.linearize(lam)
    S, g = S.cpu().numpy().copy(), g.cpu().numpy().copy()
    info = ba.info.cpu().numpy()
    So, go, co, nvo, pieces, cp = _lin_oracle(pr, lam)
    assert np.abs(S - So).max() <= 1e-9 * np.abs(So).max()
    assert np.abs(g - go).max() <= 1e-9 * np.abs(go).max()
    assert info[0] == pytest.approx(co, rel=1e-11) and info[1] == nvo and info[2] == pytest.approx(cp, rel=1e-9, abs=1e-12)
    np.testing.assert_array_equal(S, S.T)
    # blocked Cholesky solve (with damping), retraction, back-substitution
    d_ref = np.linalg.solve(So + lam * np.diag(np.diag(So)) + (0 if lam else 1e-30) * np.eye(len(go)), go)
    ba.solve(lam)
    dpose = ba.g.cpu().numpy()
    assert int(ba.bad.item()) == 0
    assert np.abs(dpose - d_ref).max() <= 1e-6 * np.abs(d_ref).max()
    new = ba.poses_new.cpu().numpy()
    for j in (0, len(new) // 2, len(new) - 1):
        np.testing.assert_allclose(new[j], ba_np.retract_pose(pr.poses[j], dpose[6 * j:6 * j + 6]), atol=1e-10)
    pts = ba.backsub(lam).cpu().numpy()
    dp = ba_np.sparse_backsub(pieces, dpose)
    assert np.abs(pts - (pr.points + dp)).max() <= 1e-8 * max(1.0, np.abs(dp).max())
    assert ba.cost() == pytest.approx(co + cp, rel=1e-10)


@pytest.mark.gpu
def test_grouped_pair_blocks_are_reproducible_and_equal_the_atomic_path(gpu):
    """The atomic-free pair stage (pairs grouped by pose pair, one writer per 6 x 6 block): bitwise identical from run to
    run, equal to the atomic path to rounding, and the grouping tiles the pair list."""
    import ctypes
    fn, data = load(gpu, SVO, "slam2", 1, 50)
    ba = gpu.sparse_ba.SparseBundleAdjuster(gpu.ba_io.build_sparse_problem(data))
    gp = ba.group_ptr.cpu().numpy()
    pa, pb, op = ba.pair_a.cpu().numpy(), ba.pair_b.cpu().numpy(), ba.obs_pose.cpu().numpy()
    assert gp[0] == 0 and gp[-1] == ba.Q and (np.diff(gp) > 0).all()
    keys = op[pa].astype(np.int64) * ba.P + op[pb]
    assert (np.diff(keys) >= 0).all() and len(np.unique(keys)) == ba.G
    assert all(len(np.unique(keys[gp[k]:gp[k + 1]])) == 1 for k in range(0, ba.G, max(1, ba.G // 50)))
    S1, g1 = ba.linearize(1e-3)
    S1, g1 = S1.clone(), g1.clone()
    S2, g2 = ba.linearize(1e-3)
    import torch
    assert torch.equal(S1, S2) and torch.equal(g1, g2)                           # no atomics: bitwise reproducible
    Sm = S2.reshape(ba.n6, ba.n6)
    assert torch.equal(Sm, Sm.T) and float(Sm.abs().max()) > 0                  # both triangles written by the pair stage itself
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    Sa, ga = torch.empty_like(ba.S), torch.empty_like(ba.g)
    gpu._lib.check(gpu._lib.lib().mqs_sba_linearize_dev(
        P(ba.poses), P(ba.pose_cam), ba.P, P(ba.calib), P(ba.sigma), P(ba.points), ba.N, P(ba.obs_ptr), P(ba.obs_pose),
        P(ba.obs_uv), ba.M, P(ba.pair_a), P(ba.pair_b), ba.Q, P(ba.prior_w), P(ba.prior_xyz), P(ba.pp_idx), P(ba.pp_poses),
        P(ba.pp_sigmas), ba.npp, 1e-3, P(Sa), P(ga), P(ba.info), P(ba.ws), ba.ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    scale = float(S1.abs().max())
    assert float((Sa - S2.reshape(-1)).abs().max()) <= 1e-12 * scale
    assert float((ga - g2).abs().max()) <= 1e-12 * float(g2.abs().max())
    # the device-side check of the grouping (info[3]): 0 for the helper's canonical grouping, > 0 when a caller hands over
    # pairs whose orientation is reversed (two groups would write one block)
    ba.linearize(0.0)
    torch.cuda.synchronize()
    assert float(ba.info[3].item()) == 0.0
    rev_a, rev_b = ba.pair_b.clone(), ba.pair_a.clone()
    gpu._lib.check(gpu._lib.lib().mqs_sba_linearize_grouped_dev(
        P(ba.poses), P(ba.pose_cam), ba.P, P(ba.calib), P(ba.sigma), P(ba.points), ba.N, P(ba.obs_ptr), P(ba.obs_pose),
        P(ba.obs_uv), ba.M, P(rev_a), P(rev_b), ba.Q, P(ba.group_ptr), ba.G, P(ba.prior_w), P(ba.prior_xyz), P(ba.pp_idx),
        P(ba.pp_poses), P(ba.pp_sigmas), ba.npp, 0.0, P(Sa), P(ga), P(ba.info), P(ba.ws), ba.ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert float(ba.info[3].item()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(50, 300), (200, 101), (250, 1500), (100, 17), (1, 6), (300, 1800), (881, 101), (97, 102), (90, 35),
                                  (333, 65), (120, 103)])
def test_reduced_system_solve_against_numpy(case, gpu):
    """mqs_sba_solve_banded_dev on random symmetric positive definite systems: a narrow band (blocked factorisation +
    product-form substitutions), dense systems small enough for one panel chunk in LDS, dense systems that need the chunked
    substitution, and one large enough for the dense substitution launches (n = 1800; round 1 sent it to rocSOLVER); with and
    without damping.  Long narrow bands are cut into independent
    chunks (csrc/chol_nd.hip: n = 5286 / hb = 101, the ICL kt2 shape, into 8; 1998 / 65 into 4; 1200 / 101, 720 / 103 and
    540 / 35 into 2; 600 / 17 into 4); sizes that are not multiples of the 32-wide block and bands narrower than a block are
    among them."""
    import ctypes
    import torch
    P_, hb = case
    n = 6 * P_
    rng = np.random.default_rng(n + hb)
    B = rng.standard_normal((n, n))
    i, j = np.indices((n, n))
    B[np.abs(i - j) > hb] = 0.0
    S = B @ B.T                                            # half bandwidth <= 2 hb ...
    S[np.abs(i - j) > hb] = 0.0                            # ... cut back to hb, then made diagonally dominant
    S = 0.5 * (S + S.T) + np.diag(np.abs(S).sum(1) + 1.0)
    g = rng.standard_normal(n)
    for lam in (0.0, 0.1):
        Sd = torch.from_numpy(S.copy()).cuda().reshape(-1)
        x = torch.from_numpy(g.copy()).cuda()
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        poses = torch.zeros((P_, 12), dtype=torch.float64, device="cuda")
        gpu._lib.check(gpu._lib.lib().mqs_sba_solve_banded_dev(
            ctypes.c_void_p(Sd.data_ptr()), ctypes.c_void_p(x.data_ptr()), P_, min(hb, n), lam, ctypes.c_void_p(poses.data_ptr()),
            None, ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ref = np.linalg.solve(S + lam * np.diag(np.diag(S)), g)
        assert int(bad.item()) == 0
        assert np.abs(x.cpu().numpy() - ref).max() <= 1e-10 * np.abs(ref).max()
        # the input contract is the LOWER triangle (include/mqslam.h): the same system with its strict upper triangle poisoned
        # gives the same bits
        Sp = S.copy()
        Sp[np.triu_indices(n, 1)] = np.nan
        Sd2 = torch.from_numpy(Sp).cuda().reshape(-1)
        x2 = torch.from_numpy(g.copy()).cuda()
        gpu._lib.check(gpu._lib.lib().mqs_sba_solve_banded_dev(
            ctypes.c_void_p(Sd2.data_ptr()), ctypes.c_void_p(x2.data_ptr()), P_, min(hb, n), lam, ctypes.c_void_p(poses.data_ptr()),
            None, ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        assert int(bad.item()) == 0 and torch.equal(x2, x)


@pytest.mark.gpu
def test_chunked_solve_plan_cache_is_bounded(gpu):
    """A session whose pose count grows with every keyframe asks for a new chunked-solve plan per shape: the cache is an LRU
    (csrc/chol_nd.hip) -- it never holds more than 8 plans, and a shape that was evicted is rebuilt and still solves."""
    import ctypes
    import torch
    L = gpu._lib.lib()
    rng = np.random.default_rng(5)
    first = None
    for k, P_ in enumerate(list(range(100, 112)) + [100]):
        n, hb = 6 * P_, 17
        S = np.zeros((n, n))
        for d in range(1, hb + 1):
            S[np.arange(n - d), np.arange(d, n)] = rng.standard_normal(n - d)
        S = S + S.T
        S[np.arange(n), np.arange(n)] = np.abs(S).sum(axis=1) + 1.0
        g = rng.standard_normal(n)
        Sd, x = torch.from_numpy(S.copy()).cuda().reshape(-1), torch.from_numpy(g.copy()).cuda()
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        poses = torch.zeros((P_, 12), dtype=torch.float64, device="cuda")
        gpu._lib.check(L.mqs_sba_solve_banded_dev(ctypes.c_void_p(Sd.data_ptr()), ctypes.c_void_p(x.data_ptr()), P_, hb, 0.0,
                                                  ctypes.c_void_p(poses.data_ptr()), None, ctypes.c_void_p(bad.data_ptr()),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ref = np.linalg.solve(S, g)
        assert int(bad.item()) == 0 and np.abs(x.cpu().numpy() - ref).max() <= 1e-10 * np.abs(ref).max()
        assert 1 <= L.mqs_sba_solve_plan_cache_size() <= 8
    assert L.mqs_sba_solve_plan_cache_size() == 8


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(881, 101), (333, 65), (150, 17), (500, 7), (107, 31), (219, 33), (214, 64), (186, 35)])
def test_chunked_cholesky_equals_the_natural_order(case, gpu, monkeypatch):
    """The same banded system solved with the band cut into 1 (natural order), 2, 4, 8 and 16 chunks (MQS_SBA_PARTS): every
    cut reproduces numpy's solve, agrees with the natural order to rounding, and is bit-reproducible run to run (no atomics,
    one writer per tile in every launch: tests/test_chol_plan.py checks that property of the plan on the CPU)."""
    import ctypes
    import torch
    P_, hb = case
    n = 6 * P_
    rng = np.random.default_rng(7 * n + hb)
    S = np.zeros((n, n))
    for d in range(1, hb + 1):
        S[np.arange(n - d), np.arange(d, n)] = rng.standard_normal(n - d)
    S = S + S.T
    S[np.arange(n), np.arange(n)] = np.abs(S).sum(axis=1) + 1.0 + rng.random(n)
    g = rng.standard_normal(n)
    ref = np.linalg.solve(S, g)
    lib = gpu._lib.lib()

    def solve():
        Sd = torch.from_numpy(S.copy()).cuda().reshape(-1)
        x = torch.from_numpy(g.copy()).cuda()
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        poses = torch.zeros((P_, 12), dtype=torch.float64, device="cuda")
        gpu._lib.check(lib.mqs_sba_solve_banded_dev(
            ctypes.c_void_p(Sd.data_ptr()), ctypes.c_void_p(x.data_ptr()), P_, hb, 0.0, ctypes.c_void_p(poses.data_ptr()), None,
            ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert int(bad.item()) == 0
        return x.cpu().numpy()

    results = {}
    for parts in (1, 2, 4, 8, 16):
        monkeypatch.setenv("MQS_SBA_PARTS", str(parts))
        a, b = solve(), solve()
        np.testing.assert_array_equal(a, b)
        assert np.abs(a - ref).max() <= 1e-11 * np.abs(ref).max(), parts
        results[parts] = a
    for parts in (2, 4, 8, 16):
        assert np.abs(results[parts] - results[1]).max() <= 1e-12 * np.abs(ref).max()
    need = {parts: lib.mqs_sba_solve_plan_dump(n, hb, parts, None, 0) for parts in (1, 2, 8)}
    assert need[1] == 0 and need[2] > 0                                     # the cut does apply to these shapes
    # a matrix that is not positive definite is reported, cut or not
    monkeypatch.setenv("MQS_SBA_PARTS", "0")
    Sbad = S.copy()
    Sbad[n // 2, n // 2] = -1.0
    Sd = torch.from_numpy(Sbad).cuda().reshape(-1)
    x = torch.from_numpy(g.copy()).cuda()
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    poses = torch.zeros((P_, 12), dtype=torch.float64, device="cuda")
    gpu._lib.check(lib.mqs_sba_solve_banded_dev(
        ctypes.c_void_p(Sd.data_ptr()), ctypes.c_void_p(x.data_ptr()), P_, hb, 0.0, ctypes.c_void_p(poses.data_ptr()), None,
        ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert int(bad.item()) == 1


@pytest.mark.gpu
def test_chunked_cholesky_on_two_streams_at_once(gpu):
    """Two banded systems of the same shape solved concurrently on two HIP streams, five times over: the plans (and the scratch
    memory of their deferred sums) are per stream, so the solves do not disturb each other -- every result equals numpy's and
    repeats bit for bit."""
    import ctypes
    import torch
    P_, hb = 333, 65
    n = 6 * P_
    lib = gpu._lib.lib()
    systems = []
    for seed in (1, 2):
        rng = np.random.default_rng(seed)
        S = np.zeros((n, n))
        for d in range(1, hb + 1):
            S[np.arange(n - d), np.arange(d, n)] = rng.standard_normal(n - d)
        S = S + S.T
        S[np.arange(n), np.arange(n)] = np.abs(S).sum(axis=1) + 1.0
        g = rng.standard_normal(n)
        systems.append((S, g, np.linalg.solve(S, g)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    poses = torch.zeros((P_, 12), dtype=torch.float64, device="cuda")
    first = [None, None]
    for rep in range(5):
        outs = []
        for k, (S, g, ref) in enumerate(systems):
            with torch.cuda.stream(streams[k]):
                Sd = torch.from_numpy(S.copy()).cuda().reshape(-1)
                x = torch.from_numpy(g.copy()).cuda()
                bad = torch.zeros(1, dtype=torch.int32, device="cuda")
                gpu._lib.check(lib.mqs_sba_solve_banded_dev(
                    ctypes.c_void_p(Sd.data_ptr()), ctypes.c_void_p(x.data_ptr()), P_, hb, 0.0, ctypes.c_void_p(poses.data_ptr()), None,
                    ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(streams[k].cuda_stream)))
                outs.append((Sd, x, bad))
        torch.cuda.synchronize()
        for k, (Sd, x, bad) in enumerate(outs):
            xs = x.cpu().numpy()
            assert int(bad.item()) == 0
            assert np.abs(xs - systems[k][2]).max() <= 1e-11 * np.abs(systems[k][2]).max()
            if first[k] is None:
                first[k] = xs
            np.testing.assert_array_equal(xs, first[k])


@pytest.mark.gpu
def test_odometry_between_factors(gpu):
    """B3 (bundle_adjust.cpp:301-309, useOdometry = 1) on the reference's example files: the odometry factors'
    contribution to the reduced camera system and to the cost equals the oracle's, and LM with them converges."""
    fn, data = load(gpu, EX, "synthetic", 2, 1)
    pr0 = gpu.ba_io.build_sparse_problem(data, use_odometry=True)
    assert len(pr0.odo_from) == sum(len(a) for a in data.odometryAssocs) > 0
    rng = np.random.default_rng(1)
    pr0 = pr0._replace(points=pr0.points + 0.01 * rng.standard_normal(pr0.points.shape))
    ba = gpu.sparse_ba.SparseBundleAdjuster(pr0)
    pr = ba.problem
    S, g = ba.linearize(0.0)
    S, g = S.cpu().numpy().copy(), g.cpu().numpy().copy()
    So, go, co, nvo, pieces, cp = _lin_oracle(pr, 0.0)
    Hb, gb, cb = ba_np.sparse_between_terms(pr.poses, pr.odo_from, pr.odo_to, pr.odo_meas, pr.odo_sigmas)
    assert cb > 0 and np.abs(Hb).max() > 0
    assert np.abs(S - (So + Hb)).max() <= 1e-9 * np.abs(So + Hb).max()
    assert np.abs(g - (go + gb)).max() <= 1e-9 * np.abs(go + gb).max()
    np.testing.assert_allclose(S, S.T, rtol=0, atol=1e-9 * np.abs(S).max())
    assert ba.cost() == pytest.approx(co + cp + cb, rel=1e-10)
    hist = ba.optimize(mode="lm")
    assert hist[-1] < hist[0] and all(b <= a for a, b in zip(hist, hist[1:]))
    # the committed -BA trajectory was produced WITH odometry: using it brings this build's output closer to it
    io = gpu.ba_io
    ref_tr = io.load_trajectory(os.path.join(EX, "traj_out.cam0-synthetic-BA.txt"))
    k = [i for i, key in enumerate(ba.problem.pose_key) if key[0] == 0]
    with_odo = np.median([np.linalg.norm(ba.poses.cpu().numpy()[i][9:] - r[1][9:]) for i, r in zip(k, ref_tr)])
    ba0 = gpu.sparse_ba.SparseBundleAdjuster(io.build_sparse_problem(data, use_odometry=False))
    ba0.optimize(mode="lm")
    without = np.median([np.linalg.norm(ba0.poses.cpu().numpy()[i][9:] - r[1][9:]) for i, r in zip(k, ref_tr)])
    print("median distance to the reference's -BA trajectory: with odometry %.4f, without %.4f" % (with_odo, without))
    assert with_odo < 1e-3 < 0.05 < without                       # measured 2.7e-4 vs 0.108 (files carry 6 digits)
    # ... and the optimised map IS the reference's GTSAM output (input map: 0.49 away)
    ref_pts = io.load_map(os.path.join(EX, "map_out-synthetic-BA.pcd"))
    assert np.abs(ba.points.cpu().numpy() - ref_pts).max() < 1e-3            # measured 2.2e-4
    poses = ba.poses.cpu().numpy()
    for c in (0, 1):
        ref_c = io.load_trajectory(os.path.join(EX, "traj_out.cam%d-synthetic-BA.txt" % c))
        kc = [i for i, key in enumerate(ba.problem.pose_key) if key[0] == c]
        assert max(np.abs(poses[i][:9] - r[1][:9]).max() for i, r in zip(kc, ref_c)) < 3e-4      # rotations, measured 7e-5


@pytest.mark.gpu
def test_sparse_equals_dense_kernels(gpu):
    """The same 4-pose full-visibility scene through both code paths."""
    import torch
    from ba_util import make_scene
    sc = make_scene(600, 4, seed=21, distortion=True, masked_frac=0.25)
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dense = gpu.bundle_adjustment.BundleAdjuster(d(sc["poses"]), d(sc["calib"]), d(sc["sigma"]), d(sc["points"]), d(sc["obs"]),
                                                 d(sc["mask"]), d(sc["prior_w"]), d(sc["prior_xyz"]))
    lin = dense.linearize(0.0).cpu().numpy()
    ptr, op, uv = [0], [], []
    for i in range(600):
        for c in range(4):
            if sc["mask"][c, i]:
                op.append(c); uv.append(sc["obs"][c, i])
        ptr.append(len(op))
    SP = gpu.ba_io.SparseProblem
    e = np.zeros(0)
    pr = SP(poses=sc["poses"], pose_cam=np.arange(4, dtype=np.int32), pose_key=[(c, 0) for c in range(4)], calib=sc["calib"],
            sigma=sc["sigma"], points=sc["points"], obs_ptr=np.array(ptr, dtype=np.int64), obs_pose=np.array(op, dtype=np.int32),
            obs_uv=np.array(uv), prior_w=sc["prior_w"], prior_xyz=sc["prior_xyz"], pose_prior_idx=np.zeros(0, np.int32),
            pose_prior_sigmas=np.zeros((0, 6)), odo_from=e.astype(np.int32), odo_to=e.astype(np.int32),
            odo_meas=np.zeros((0, 12)), odo_sigmas=np.zeros((0, 6)))
    sp = gpu.sparse_ba.SparseBundleAdjuster(pr)
    S, g = sp.linearize(0.0)
    S, g = S.cpu().numpy(), g.cpu().numpy()
    assert np.abs(S.reshape(-1) - lin[:576]).max() <= 1e-10 * np.abs(lin[:576]).max()
    assert np.abs(g - lin[576:600]).max() <= 1e-10 * np.abs(lin[576:600]).max()
    assert sp.info[0].item() == pytest.approx(lin[600], rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("example", True, "gtsam"), ("example", False, "marquardt"), ("svo", False, "gtsam")])
def test_the_optimiser_inside_the_library_takes_the_host_loop_s_steps(case, gpu):
    """`mqs_sba_optimize_lm_dev` (the LM loop of bundle_adjust.cpp:323-324 inside the library, one synchronisation per trial)
    against the same schedule driven from Python over the individual entry points: the same accepted costs, the same estimate.
    (Not bit for bit: the atomics of the odometry and pose-prior kernels order their sums differently from run to run, and the
    host loop evaluates the pose prior's cost in numpy.)"""
    which, odo, damping = case
    if which == "example":
        fn, data = load(gpu, EX, "synthetic", 2, 1)
    else:
        fn, data = load(gpu, SVO, "slam2", 1, 50)
    pr = gpu.ba_io.build_sparse_problem(data, use_odometry=odo)
    rng = np.random.default_rng(5)
    pr = pr._replace(points=pr.points + 0.005 * rng.standard_normal(pr.points.shape))
    a, b = gpu.sparse_ba.SparseBundleAdjuster(pr), gpu.sparse_ba.SparseBundleAdjuster(pr)
    ha = a.optimize(mode="lm", damping=damping)
    hb = b.optimize_host_loop(mode="lm", damping=damping)
    assert len(ha) == len(hb) >= 3
    np.testing.assert_allclose(ha, hb, rtol=1e-9)
    assert ha[-1] < ha[0]
    pa, pb = a.poses.cpu().numpy(), b.poses.cpu().numpy()
    assert np.abs(pa - pb).max() <= 1e-8 * max(1.0, np.abs(pb).max())
    assert np.abs(a.points.cpu().numpy() - b.points.cpu().numpy()).max() <= 1e-7 * max(1.0, np.abs(b.points.cpu().numpy()).max())
    # a second call continues from the adjusted estimate: its first cost is the last one of the first call
    assert a.optimize(mode="lm", damping=damping)[0] == pytest.approx(ha[-1], rel=1e-12)
    assert a.cost() == pytest.approx(ha[-1], rel=1e-9) or a.cost() <= ha[-1]


@pytest.mark.gpu
def test_the_optimiser_inside_the_library_at_its_edges(gpu):
    """mqs_sba_optimize_lm_dev with nothing to do and with parts of the problem absent: zero iterations return the start cost and
    move nothing; a converged estimate is left where it is; no odometry, no pose prior (the gauge then rests on the point priors),
    a landmark without observations -- each equals the host loop; a non-positive lambda factor is refused."""
    import ctypes
    fn, data = load(gpu, EX, "synthetic", 2, 1)
    pr = gpu.ba_io.build_sparse_problem(data, use_odometry=True)
    SB = gpu.sparse_ba.SparseBundleAdjuster
    a = SB(pr)
    p0, x0 = a.poses.clone(), a.points.clone()
    c0 = a.cost()
    assert a.optimize(iters=0, mode="lm") == [pytest.approx(c0, rel=1e-12)]
    assert bool((a.poses == p0).all()) and bool((a.points == x0).all())
    h = a.optimize(mode="lm")
    again = a.optimize(mode="lm")                                                # converged: at most one more small step
    assert len(again) <= 2 and again[-1] <= h[-1] * (1 + 1e-9)
    # parts absent
    ptr = np.asarray(pr.obs_ptr).copy()
    variants = {
        "no odometry": pr._replace(odo_from=pr.odo_from[:0], odo_to=pr.odo_to[:0], odo_meas=pr.odo_meas[:0], odo_sigmas=pr.odo_sigmas[:0]),
        "no pose prior": pr._replace(pose_prior_idx=pr.pose_prior_idx[:0], pose_prior_sigmas=pr.pose_prior_sigmas[:0]),
        "a landmark nobody sees": pr._replace(points=np.vstack([pr.points, [[1.0, 2.0, 3.0]]]), obs_ptr=np.append(ptr, ptr[-1]),
                                              prior_w=None if pr.prior_w is None else np.append(pr.prior_w, 0.0),
                                              prior_xyz=None if pr.prior_xyz is None else np.vstack([pr.prior_xyz, [[0.0, 0.0, 0.0]]])),
    }
    rng = np.random.default_rng(9)
    for name, q in variants.items():
        q = q._replace(points=q.points + 0.004 * rng.standard_normal(q.points.shape))
        u, v = SB(q), SB(q)
        hu, hv = u.optimize(mode="lm"), v.optimize_host_loop(mode="lm")
        assert len(hu) == len(hv) and len(hu) >= 2, name
        np.testing.assert_allclose(hu, hv, rtol=1e-9, err_msg=name)
        assert np.abs(u.poses.cpu().numpy() - v.poses.cpu().numpy()).max() < 1e-8, name
    assert np.abs(u.points.cpu().numpy()[-1] - q.points[-1]).max() == 0.0        # the unseen landmark stays where it was
    # refused arguments
    lm = gpu.sparse_ba._LmParams(lambda_initial=1e-5, lambda_factor=1.0, lambda_upper=1e5, abs_tol=1e-5, rel_tol=1e-5, max_iterations=10, damping=0)
    prd = a._problem_struct()
    hist = np.zeros(11)
    n = ctypes.c_int32(0)
    rc = gpu._lib.lib().mqs_sba_optimize_lm_dev(ctypes.byref(prd), ctypes.byref(lm), hist.ctypes.data_as(gpu._lib.c_f64p), 11, ctypes.byref(n), None)
    assert rc == -1                                                               # MQS_E_ARG


@pytest.mark.gpu
def test_the_optimiser_on_a_problem_without_landmarks(gpu):
    """N = 0: a pose graph -- a prior on the first pose and odometry factors between neighbours (bundle_adjust.cpp:301-309) -- goes
    through the same entry points (no landmark kernels run): the chain, started from perturbed poses, settles where the factors
    cost three orders of magnitude less; library driver and host loop agree, and the oracle prices the result the same."""
    fn, data = load(gpu, EX, "synthetic", 2, 1)
    full = gpu.ba_io.build_sparse_problem(data, use_odometry=True)
    keep = [i for i, k in enumerate(full.pose_key) if k[0] == 0]                # camera 0's poses: a chain of odometry factors
    idx = {i: j for j, i in enumerate(keep)}
    odo = [k for k in range(len(full.odo_from)) if int(full.odo_from[k]) in idx and int(full.odo_to[k]) in idx]
    assert len(odo) >= len(keep) - 1
    truth = full.poses[keep]
    rng = np.random.default_rng(4)
    start = truth.copy()
    start[1:, 9:] += 0.05 * rng.standard_normal((len(keep) - 1, 3))
    pr = full._replace(poses=start, pose_cam=full.pose_cam[keep], pose_key=[full.pose_key[i] for i in keep], points=np.zeros((0, 3)),
                       obs_ptr=np.zeros(1, np.int64), obs_pose=np.zeros(0, np.int32), obs_uv=np.zeros((0, 2)), prior_w=None, prior_xyz=None,
                       pose_prior_idx=np.array([0], np.int32), pose_prior_sigmas=np.array([[1e-3] * 6]),
                       odo_from=np.array([idx[int(full.odo_from[k])] for k in odo], np.int32),
                       odo_to=np.array([idx[int(full.odo_to[k])] for k in odo], np.int32), odo_meas=full.odo_meas[odo], odo_sigmas=full.odo_sigmas[odo])
    SB = gpu.sparse_ba.SparseBundleAdjuster
    a, b = SB(pr), SB(pr)
    ha, hb = a.optimize(mode="lm"), b.optimize_host_loop(mode="lm")
    assert len(ha) == len(hb) >= 2 and ha[-1] < 1e-3 * ha[0]
    np.testing.assert_allclose(ha, hb, rtol=1e-8, atol=1e-12)
    # what is left is the oracle's cost of the odometry factors at the result (the prior on the first pose costs next to nothing)
    Hb, gb, cb = ba_np.sparse_between_terms(a.poses.cpu().numpy(), pr.odo_from, pr.odo_to, pr.odo_meas, pr.odo_sigmas)
    assert cb <= ha[-1] * (1 + 1e-9) + 1e-12 and cb >= 0.9 * ha[-1] - 1e-9
    assert a.cost() == pytest.approx(ha[-1], rel=1e-9, abs=1e-12)


@pytest.mark.gpu
def test_worst_residual_per_landmark_equals_the_host_projection(gpu):
    """`mqs_sba_worst_residual_dev` (the screen either side of an adjustment in the SLAM loop) against the numpy projection that
    tests/test_slam_loop.py pins to the oracle: per landmark the largest pixel residual; +inf behind a camera; 0 without observations."""
    fn, data = load(gpu, SVO, "slam2", 1, 50)
    pr = gpu.ba_io.build_sparse_problem(data)
    rng = np.random.default_rng(2)
    pts = pr.points + 0.02 * rng.standard_normal(pr.points.shape)
    ptr = np.asarray(pr.obs_ptr)
    i_behind = int(np.argmax(np.diff(ptr) > 0))                               # a landmark with observations: put it behind its first camera
    j = int(pr.obs_pose[ptr[i_behind]])
    R, c = pr.poses[j, :9].reshape(3, 3), pr.poses[j, 9:]
    pts[i_behind] = c - 2.0 * R[:, 2]
    pr = pr._replace(points=pts, sigma=np.full_like(pr.sigma, 1.7))
    ba = gpu.sparse_ba.SparseBundleAdjuster(pr)
    got, zmin = ba.worst_residuals(with_min_depth=True)
    np.testing.assert_array_equal(got, ba.worst_residuals())
    lm = np.repeat(np.arange(len(pts)), np.diff(ptr))
    res = gpu.slam_device._reprojection_residuals(pr.poses, pts, pr.calib[0], lm, np.asarray(pr.obs_pose), np.asarray(pr.obs_uv).reshape(-1, 2))
    want = np.zeros(len(pts))
    np.maximum.at(want, lm, res)
    assert np.isinf(got[i_behind])                                            # (the numpy twin clamps the depth instead)
    ok = np.arange(len(pts)) != i_behind
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9)
    assert np.all(got[np.diff(ptr) == 0] == 0.0)
    # the smallest depth of a landmark in the cameras that see it
    Rz = pr.poses[:, :9].reshape(-1, 3, 3)[:, :, 2]
    depth = np.einsum("nk,nk->n", Rz[np.asarray(pr.obs_pose)], pts[lm] - pr.poses[np.asarray(pr.obs_pose), 9:])
    want_z = np.full(len(pts), np.inf)
    np.minimum.at(want_z, lm, depth)
    np.testing.assert_allclose(zmin[np.isfinite(want_z)], want_z[np.isfinite(want_z)], rtol=1e-12, atol=1e-12)
    assert np.all(np.isinf(zmin[np.diff(ptr) == 0])) and zmin[i_behind] < 0


@pytest.mark.gpu
def test_example_converges_near_committed_reference_output(gpu):
    """Loose anchor: the committed *-BA outputs were produced with odometry + iSAM2 and
    6-digit inputs, so only 'same basin' is asserted: the optimised map is the 20-unit cube and the
    optimised trajectory stays within the reference's own correction magnitude of its -BA output."""
    io = gpu.ba_io
    fn, data = load(gpu, EX, "synthetic", 2, 1)
    pr = io.build_sparse_problem(data)
    ba = gpu.sparse_ba.SparseBundleAdjuster(pr)
    hist = ba.optimize(mode="lm")
    assert hist[-1] < hist[0] and all(b <= a for a, b in zip(hist, hist[1:]))
    assert hist[-1] / len(pr.obs_pose) < 2.0                     # chi^2 per factor ~ 1 (pixel sigma 1)
    pts = ba.points.cpu().numpy()
    ref_pts = io.load_map(os.path.join(EX, "map_out-synthetic-BA.pcd"))
    assert np.abs(pts - ref_pts).max() < 0.5                     # cube of edge 20
    ref_tr = io.load_trajectory(os.path.join(EX, "traj_out.cam0-synthetic-BA.txt"))
    in_tr = io.load_trajectory(os.path.join(EX, "traj_out.cam0-synthetic.txt"))
    ours = ba.poses.cpu().numpy()
    k = [i for i, key in enumerate(ba.problem.pose_key) if key[0] == 0]
    d_ours = np.array([np.linalg.norm(ours[i][9:] - r[1][9:]) for i, r in zip(k, ref_tr)])
    d_in = np.array([np.linalg.norm(a[1][9:] - r[1][9:]) for a, r in zip(in_tr, ref_tr)])
    assert np.median(d_ours) < np.median(d_in)                   # closer to the reference's BA output than the input was


@pytest.mark.gpu
def test_svo_dataset_full_optimisation(gpu, tmp_path):
    """The reference's real data set (186 poses, 1046 points, 7494 observations), file in -> file out through
    the CLI-compatible tool; anchored on the committed slam2-BA output."""
    import shutil, subprocess, sys
    work = tmp_path / "svo"
    shutil.copytree(SVO, work)
    for f in ("traj_out.cam0-slam2-BA.txt", "map_out-slam2-BA.pcd"):
        os.remove(work / f)
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bundle_adjust.py"), str(work), "slam2", "1", "50", "0", "1", "0", "1", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    io = gpu.ba_io
    ours = io.load_map(str(work / "map_out-slam2-BA.pcd"))
    ref = io.load_map(os.path.join(SVO, "map_out-slam2-BA.pcd"))
    inp = io.load_map(os.path.join(SVO, "map_out-slam2.pcd"))
    assert ours.shape == ref.shape == (1046, 3)
    d_ours = np.linalg.norm(ours - ref, axis=1)
    d_in = np.linalg.norm(inp - ref, axis=1)
    # measured (tools/svo_measure.py, round 2): median 5.6e-4, 90 % 1.4e-3 against an input 0.65 away -- the reference's own
    # optimum to the precision GTSAM's stop rule (relative decrease 1e-5) leaves; pinned ~2x above the measurement
    assert np.median(d_ours) < 1.2e-3 and np.quantile(d_ours, 0.9) < 3e-3 and np.median(d_in) > 0.5
    tr = io.load_trajectory(str(work / "traj_out.cam0-slam2-BA.txt"))
    rt = io.load_trajectory(os.path.join(SVO, "traj_out.cam0-slam2-BA.txt"))
    it = io.load_trajectory(os.path.join(SVO, "traj_out.cam0-slam2.txt"))
    assert len(tr) == len(rt) == 186
    e_ours = np.median([np.linalg.norm(a[1][9:] - b[1][9:]) for a, b in zip(tr, rt)])
    e_in = np.median([np.linalg.norm(a[1][9:] - b[1][9:]) for a, b in zip(it, rt)])
    assert e_ours < 1.3e-3 and e_in > 0.5                         # measured 6.5e-4 (input 0.76)
    assert max(np.linalg.norm(a[1][9:] - b[1][9:]) for a, b in zip(tr, rt)) < 0.04           # measured 0.020
    assert np.median([np.abs(a[1][:9] - b[1][:9]).max() for a, b in zip(tr, rt)]) < 1.5e-4      # rotations: measured 7.1e-5
    # the reference's published accuracy numbers: ATE rmse 0.395356 m before BA,
    # 0.021598 m after the reference's (GTSAM) BA -- reproduced by the restated evaluate_ate on the committed
    # files, and matched by this build's BA output
    from util import ate_rmse
    gt = [(t, p[9:]) for t, p in io.load_trajectory(os.path.join(SVO, "traj_groundtruth.txt"))]
    xyz = lambda tr_: [(t, p[9:]) for t, p in tr_]
    assert ate_rmse(xyz(it), gt)[0] == pytest.approx(0.395356, abs=1e-6)
    assert ate_rmse(xyz(rt), gt)[0] == pytest.approx(0.021598, abs=1e-6)
    ours_ate, npairs = ate_rmse(xyz(tr), gt)
    assert npairs == 186 and ours_ate == pytest.approx(0.021598, rel=0.015)                   # measured 0.021446


def test_recorder_writes_what_the_loader_reads(mqs, tmp_path):
    """BundleAdjustmentInfoContainer (the SLAM-side recorder, slam2.py:743-865) -> BA_info.* files -> load_data: features keep
    their per-frame numbering when a frame receives features in several steps, steps without events are empty blocks."""
    io = mqs.ba_io
    info = io.BundleAdjustmentInfoContainer(str(tmp_path), "rec", 1)
    info.set_calibration(np.array([[480.0, 0.5, 320.0], [0, 470.0, 240.0], [0, 0, 1.0]]), [0.01, -0.002, 1e-4, 2e-4])
    info.set_point3DAddedIdxs([0, 1, 2])
    info.add_points2D_3Dassoc([[10, 11], [20, 21], [30, 31]], [0, 1, 2], 0)
    info.next_step()                                                     # frame 1: nothing recorded
    info.next_step()                                                     # frame 2: tracks + a new landmark seen in frames 0 and 2
    info.add_points2D_3Dassoc([[12, 13], [22, 23]], [0, 1], 2)
    info.set_point3DAddedIdxs([3])
    info.add_points2D_3Dassoc([[40, 41]], [3], 0)                        # appended to frame 0's list: index 3 there
    info.add_points2D_3Dassoc([[42, 43]], [3], 2)
    P = np.eye(4)
    P[:3, 3] = [0.1, -0.2, 0.3]
    info.add_odometry(P, 0, 2)
    info.write_all()
    info.write_noise(point2D=1.0)
    text = lambda name: open(os.path.join(str(tmp_path), name)).read().split("\n")
    assert text("BA_info.measurements.points2D.cam0-rec.txt")[2:] == [
        "%.16e %.16e" % (10, 11), "%.16e %.16e" % (20, 21), "%.16e %.16e" % (30, 31), "%.16e %.16e" % (40, 41), "", "",
        "%.16e %.16e" % (12, 13), "%.16e %.16e" % (22, 23), "%.16e %.16e" % (42, 43), ""]
    assert text("BA_info.measurements.point2D3DAssocs.cam0-rec.txt")[2:] == ["0 0 0", "0 1 1", "0 2 2", "", "", "2 0 0", "2 1 1", "0 3 3", "2 2 3", ""]
    assert text("BA_info.measurements.point3DAddedIdxs-rec.txt")[2:] == ["0", "1", "2", "", "", "3", ""]
    assert text("BA_info.measurements.odometryAssocs-rec.txt")[2:] == ["", "", "0 0 0 2", ""]
    od = text("BA_info.measurements.odometry-rec.txt")[2:]
    assert od[:2] == ["", ""] and [float(v) for v in od[2].split()] == pytest.approx([-0.1, 0.2, -0.3, 0, 0, 0, 1])
    assert [float(v) for v in text("BA_info.calibrations.cam0.txt")[1].split()] == [480.0, 470.0, 0.5, 320.0, 240.0, 0.01, -0.002, 1e-4, 2e-4]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 300, 9), (200, 5000, 17), (881, 13293, 17), (7, 50, 7), (70000 // 100, 3, 2)])
def test_pair_grouping_on_the_device_equals_numpy(shape, gpu):
    """mqs_sba_group_pairs_dev (csrc/pair_group.hip: generation, stable radix sort by pose pair, ordered compaction of the
    group boundaries) returns the lists of sparse_ba.build_pairs + group_pairs (numpy stable argsort) element for element:
    a small problem, a mid-size one, the ICL-kt2 shape (2.0 M pairs), landmarks seen by every pose, a tiny one."""
    import torch
    P_, N_, span = shape
    rng = np.random.default_rng(P_ * 7 + N_)
    S = gpu.sparse_ba
    ptr, poses = [0], []
    for i in range(N_):
        first = int(rng.integers(0, max(1, P_ - span + 1)))
        seen = np.sort(rng.choice(np.arange(first, min(P_, first + span)), size=int(rng.integers(0, min(span, P_) + 1)), replace=False))
        poses.extend(seen.tolist())
        ptr.append(len(poses))
    obs_ptr, obs_pose = np.array(ptr, dtype=np.int64), np.array(poses, dtype=np.int32)
    pa, pb = S.build_pairs(obs_ptr)
    pa, pb, gp = S.group_pairs(pa, pb, obs_pose, P_)
    da, db, dg = S.group_pairs_dev(obs_ptr, torch.from_numpy(obs_ptr).cuda(), torch.from_numpy(obs_pose).cuda(), P_)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(da.cpu().numpy(), pa)
    np.testing.assert_array_equal(db.cpu().numpy(), pb)
    np.testing.assert_array_equal(dg.cpu().numpy(), gp)


def test_observation_sort_is_the_stable_per_landmark_sort(mqs):
    """sparse_ba.sort_observations_by_pose (one vectorised lexsort) against the per-landmark stable argsort it replaces."""
    import collections
    rng = np.random.default_rng(3)
    Pr = collections.namedtuple("Pr", "obs_ptr obs_pose obs_uv")
    k = rng.integers(0, 9, 400)
    ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    op = rng.integers(0, 12, int(ptr[-1])).astype(np.int32)                    # repeated poses: stability matters
    uv = rng.standard_normal((int(ptr[-1]), 2))
    out = mqs.sparse_ba.sort_observations_by_pose(Pr(ptr, op, uv))
    for i in range(len(k)):
        a, b = int(ptr[i]), int(ptr[i + 1])
        o = np.argsort(op[a:b], kind="stable")
        np.testing.assert_array_equal(out.obs_pose[a:b], op[a:b][o])
        np.testing.assert_array_equal(out.obs_uv[a:b], uv[a:b][o])


@pytest.mark.gpu
def test_observation_sort_on_the_device_equals_numpy(gpu):
    """mqs_sba_sort_observations_dev (key = landmark * P + pose, stable radix sort, one gather) against the numpy statement of the
    same sort: random CSR problems with empty landmarks and repeated poses inside a landmark (stability decides the order of the
    measurements), from a handful to 300 k observations, and the reversed SVO problem."""
    import collections
    import torch
    Pr = collections.namedtuple("Pr", "obs_ptr obs_pose obs_uv")
    rng = np.random.default_rng(5)
    cases = [(1, 3, 5), (7, 5, 3), (400, 12, 9), (5000, 300, 40), (20000, 881, 31)]
    for N, P, kmax in cases:
        k = rng.integers(0, kmax, N)
        k[rng.integers(0, N, max(1, N // 10))] = 0                                  # empty landmarks
        ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
        M = int(ptr[-1])
        op = rng.integers(0, P, M).astype(np.int32)
        uv = rng.standard_normal((M, 2))
        want = gpu.sparse_ba.sort_observations_by_pose(Pr(ptr, op, uv))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        got_pose, got_uv, order = gpu.sparse_ba.sort_observations_dev(t(ptr), t(op), t(uv), N, P, want_order=True)
        np.testing.assert_array_equal(got_pose.cpu().numpy(), want.obs_pose)
        np.testing.assert_array_equal(got_uv.cpu().numpy(), want.obs_uv)
        o = order.cpu().numpy()
        np.testing.assert_array_equal(op[o], want.obs_pose)
        assert sorted(o.tolist()) == list(range(M))
    # a recorded problem handed over in REVERSE pose order inside every landmark: the adjuster sorts it on the device and
    # linearises to the same reduced system as from the file order
    fn, data = load(gpu, SVO, "slam2", 1, 50)
    pr = gpu.ba_io.build_sparse_problem(data)
    ptr = np.asarray(pr.obs_ptr)
    rev = np.concatenate([np.arange(int(ptr[i + 1]) - 1, int(ptr[i]) - 1, -1) for i in range(len(ptr) - 1)]).astype(np.int64)
    pr_rev = pr._replace(obs_pose=np.asarray(pr.obs_pose)[rev].copy(), obs_uv=np.asarray(pr.obs_uv)[rev].copy())
    a, b = gpu.sparse_ba.SparseBundleAdjuster(pr), gpu.sparse_ba.SparseBundleAdjuster(pr_rev)
    assert torch.equal(a.obs_pose, b.obs_pose) and torch.equal(a.obs_uv, b.obs_uv)
    Sa, ga = a.linearize(0.0)
    Sb, gb = b.linearize(0.0)
    assert torch.equal(Sa, Sb) and torch.equal(ga, gb)
