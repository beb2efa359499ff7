"""
Pose from 3D-2D correspondences (the solvePnP / solvePnPRansac step of
Work/SLAM/application/own/slam2.py:453-490, 576-577).

Pin: on the reference's RECORDED inlier tracks (SVO data set) the minimiser of the pixel reprojection error
reproduces the reference's own recorded poses -- outputs of the real cv2.solvePnP -- so the oracle, the
device arithmetic compiled for the host, and the GPU path are all checked against real reference output.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import pnp_np

HERE = os.path.dirname(os.path.abspath(__file__))
SVO = os.path.join(HERE, "golden", "ba_svo")
F64 = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(F64)


def svo_frames(mqs):
    """Per recorded frame: (frame, world points of the tracked already-triangulated landmarks, their pixels)."""
    io = mqs.ba_io
    d = io.load_data(io.create_filenames(SVO, "slam2", 1), 50)
    cal = d.calibrations[0]
    intr = np.array([cal[0], cal[1], cal[3], cal[4], cal[5], cal[6], cal[7], cal[8], 0.0])
    added = {}
    for s, ids in enumerate(d.point3DAddedIdxs):
        for p in ids:
            added[p] = s
    frames = []
    for f in range(1, len(d.point2D3DAssocs[0])):
        old = [(i2, p3) for (fr, i2, p3) in d.point2D3DAssocs[0][f] if fr == f and added[p3] < f]
        X = np.array([d.points3D[p3] for _, p3 in old])
        uv = np.array([d.points2D[0][f][i2] for i2, _ in old])
        frames.append((f, X, uv))
    return d, intr, frames


def world_to_camera(pose12):
    R = pose12[:9].reshape(3, 3)
    return np.concatenate([R.T, (-R.T @ pose12[9:]).reshape(3, 1)], axis=1)


def synthetic_problem(n, seed, outliers=0, noise=0.5, dist=True):
    rng = np.random.default_rng(seed)
    intr = np.array([480.0, 470.0, 320.0, 240.0, 0.08, -0.02, 0.001, -0.0015, 0.0]) if dist else \
        np.array([480.0, 480.0, 320.0, 240.0, 0, 0, 0, 0, 0.0])
    rvec = rng.normal(0, 0.3, 3)
    tvec = np.array([0.3, -0.2, 8.0]) + rng.normal(0, 0.3, 3)
    X = rng.uniform(-2.5, 2.5, (n, 3))
    uv = pnp_np.project(rvec, tvec, X, intr) + noise * rng.standard_normal((n, 2))
    if outliers:
        uv[:outliers] += rng.uniform(20, 80, (outliers, 2)) * rng.choice([-1, 1], (outliers, 2))
    return X, uv, intr, rvec, tvec


# ------------------------------------------------------------------------------------------------
# CPU: oracle against the reference's recorded poses; device arithmetic (host build) against the oracle
# ------------------------------------------------------------------------------------------------
def test_oracle_reproduces_recorded_poses_on_plain_frames(mqs):
    d, intr, frames = svo_frames(mqs)
    # frames 1..6 add no landmarks: their recorded pose is cv2.solvePnP on exactly these inliers (slam2.py:489)
    for f, X, uv in frames[:6]:
        assert not d.point3DAddedIdxs[f]
        P0 = world_to_camera(d.poses[0][f - 1][1])
        rv, tv, cost, _ = pnp_np.solve_pnp(X, uv, intr, pnp_np.rodrigues_inv(P0[:, :3]), P0[:, 3])
        Pr = world_to_camera(d.poses[0][f][1])
        assert np.abs(pnp_np.rodrigues(rv) - Pr[:, :3]).max() < 5e-7
        assert np.abs(tv - Pr[:, 3]).max() < 5e-7
        assert np.sqrt(cost / len(X)) < 1.0                     # px RMS: max_solvePnP_reproj_error = 2 (slam2.py:1091)


@pytest.mark.parametrize("n,dist", [(6, False), (12, True), (80, True)])
def test_device_math_vs_oracle_with_and_without_start(n, dist, host_math):
    X, uv, intr, rvec, tvec = synthetic_problem(n, seed=n, dist=dist)
    Xc, uvc = np.ascontiguousarray(X), np.ascontiguousarray(uv)
    ro, to, cost_o, _ = pnp_np.solve_pnp(X, uv, intr)
    for use_guess in (0, 1):
        P = np.ascontiguousarray(np.c_[pnp_np.rodrigues(rvec + 0.05), tvec + 0.3]) if use_guess else np.zeros((3, 4))
        info = np.zeros(3)
        rc = host_math.host_pnp_refine(_p(Xc), _p(uvc), ctypes.c_int64(n), _p(intr), _p(P), use_guess, 100,
                                       ctypes.c_double(1e-13), _p(info))
        assert rc == 0 and info[2] == 1.0
        np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(ro), atol=2e-8)
        np.testing.assert_allclose(P[:, 3], to, atol=2e-7)
        np.testing.assert_allclose(info[0], cost_o, rtol=1e-9, atol=1e-18)
        np.testing.assert_allclose(P[:, :3] @ P[:, :3].T, np.eye(3), atol=1e-12)


def planar_problem(n, seed, noise=0.3):
    rng = np.random.default_rng(seed)
    intr = np.array([480.0, 480.0, 320.0, 240.0, -0.06, 0.01, 0.0005, -0.0003, 0.0])
    rvec = np.array([0.2, -0.3, 0.1]) + rng.normal(0, 0.1, 3)
    tvec = np.array([0.5, -0.3, 9.0])
    a, b = rng.uniform(-3, 3, n), rng.uniform(-2, 2, n)
    e1, e2 = np.array([0.8, 0.1, 0.59]), np.array([-0.2, 0.97, 0.1])
    e1 /= np.linalg.norm(e1); e2 -= e1 * (e1 @ e2); e2 /= np.linalg.norm(e2)
    X = np.array([1.0, -2.0, 0.5]) + a[:, None] * e1 + b[:, None] * e2          # a tilted plane, not through the origin
    uv = pnp_np.project(rvec, tvec, X, intr) + noise * rng.standard_normal((n, 2))
    return X, uv, intr, rvec, tvec


@pytest.mark.parametrize("n", [6, 30])
def test_device_math_planar_start(n, host_math):
    """Coplanar points (chessboard / plane initialisation, slam2.py:1136-1157): the 3-D DLT is rank deficient; the start
    comes from the plane homography, as in OpenCV."""
    X, uv, intr, rvec, tvec = planar_problem(n, seed=n)
    ro, to, cost_o, _ = pnp_np.solve_pnp(X, uv, intr)
    P = np.zeros((3, 4)); info = np.zeros(3)
    rc = host_math.host_pnp_refine(_p(np.ascontiguousarray(X)), _p(np.ascontiguousarray(uv)), ctypes.c_int64(n), _p(intr), _p(P), 0,
                                   100, ctypes.c_double(1e-13), _p(info))
    assert rc == 0
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(ro), atol=2e-8)
    np.testing.assert_allclose(P[:, 3], to, atol=2e-7)
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(rvec), atol=0.05)      # and it is the true pose, not a mirror
    Pd = np.zeros((3, 4))
    assert host_math.host_pnp_dlt(_p(np.ascontiguousarray(X)), _p(np.ascontiguousarray(uv)), ctypes.c_int64(n), _p(intr), _p(Pd)) == 0
    np.testing.assert_allclose(Pd[:, :3] @ Pd[:, :3].T, np.eye(3), atol=1e-10)
    np.testing.assert_allclose(Pd[:, :3], pnp_np.rodrigues(rvec), atol=0.1)


def test_device_dlt_vs_oracle_dlt(host_math):
    X, uv, intr, rvec, tvec = synthetic_problem(40, seed=3, noise=0.0)
    R, t = pnp_np.dlt_pose(X, uv, intr)
    P = np.zeros((3, 4))
    assert host_math.host_pnp_dlt(_p(np.ascontiguousarray(X)), _p(np.ascontiguousarray(uv)), ctypes.c_int64(40), _p(intr), _p(P)) == 0
    # exact data: both recover the true pose up to the 5-iteration undistortion residual
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(rvec), atol=1e-5)
    np.testing.assert_allclose(P[:, :3], R, atol=1e-5)
    np.testing.assert_allclose(P[:, 3], t, atol=1e-4)


def test_device_math_reproduces_recorded_poses(mqs, host_math):
    d, intr, frames = svo_frames(mqs)
    for f, X, uv in frames[:6]:
        P = np.ascontiguousarray(world_to_camera(d.poses[0][f - 1][1]))
        info = np.zeros(3)
        assert host_math.host_pnp_refine(_p(np.ascontiguousarray(X)), _p(np.ascontiguousarray(uv)), ctypes.c_int64(len(X)),
                                         _p(intr), _p(P), 1, 100, ctypes.c_double(1e-13), _p(info)) == 0
        Pr = world_to_camera(d.poses[0][f][1])
        assert np.abs(P - Pr).max() < 5e-7


def test_rodrigues_round_trip(mqs):
    rng = np.random.default_rng(0)
    for _ in range(20):
        r = rng.normal(0, 1.2, 3)
        r *= min(1.0, 3.0 / np.linalg.norm(r))                      # keep the angle below pi: unique vector
        R = mqs.pnp.Rodrigues(r)
        np.testing.assert_allclose(R, pnp_np.rodrigues(r), atol=1e-14)
        np.testing.assert_allclose(mqs.pnp.Rodrigues(R).ravel(), r, atol=1e-10)
    Rpi = mqs.pnp.Rodrigues(np.array([np.pi, 0, 0]))
    np.testing.assert_allclose(np.abs(mqs.pnp.Rodrigues(Rpi).ravel()), [np.pi, 0, 0], atol=1e-6)


def test_facade_argument_errors(mqs):
    with pytest.raises(ValueError):
        mqs.pnp.solve_pnp_pose(np.zeros((5, 2)), np.zeros((5, 2)), np.ones(9))
    with pytest.raises(ValueError):
        mqs.pnp.solve_pnp_pose(np.zeros((5, 3)), np.zeros((4, 2)), np.ones(9))
    with pytest.raises(ValueError):
        mqs.pnp.solvePnP(np.zeros((8, 3)), np.zeros((8, 2)), np.eye(3), None, useExtrinsicGuess=True)
    s = mqs.pnp.draw_samples(30, 16, 6, seed=1)
    assert s.shape == (16, 6) and s.dtype == np.int32
    assert all(len(set(row)) == 6 for row in s) and s.max() < 30
    np.testing.assert_array_equal(s, mqs.pnp.draw_samples(30, 16, 6, seed=1))


# ------------------------------------------------------------------------------------------------
# GPU: parity against the oracle through the C ABI, and against the reference's recorded poses
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,dist", [(6, False), (7, True), (64, True), (65, True), (300, True), (5000, False)])
def test_gpu_solve_pnp_vs_oracle(n, dist, gpu):
    X, uv, intr, rvec, tvec = synthetic_problem(n, seed=100 + n, dist=dist)
    ro, to, cost_o, _ = pnp_np.solve_pnp(X, uv, intr)
    K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    for guess in (False, True):
        kw = dict(rvec=rvec + 0.05, tvec=tvec + 0.3, useExtrinsicGuess=True) if guess else {}
        ret, r, t = gpu.pnp.solvePnP(X, uv, K, intr[4:8], **kw)
        assert ret and r.shape == (3, 1) and t.shape == (3, 1)
        np.testing.assert_allclose(pnp_np.rodrigues(r.ravel()), pnp_np.rodrigues(ro), atol=2e-8)
        np.testing.assert_allclose(t.ravel(), to, atol=2e-7)
    P, info = gpu.pnp.solve_pnp_pose(X.astype(np.float32), uv.astype(np.float32), intr)      # slam2.py:19 passes float32
    assert info[2] == n and int(info[3]) & 1
    np.testing.assert_allclose(info[0], pnp_np.solve_pnp(X.astype(np.float32), uv.astype(np.float32), intr)[2], rtol=1e-7)


@pytest.mark.gpu
def test_gpu_planar_points_and_planar_ransac(gpu):
    X, uv, intr, rvec, tvec = planar_problem(80, seed=5)
    ro, to, _, _ = pnp_np.solve_pnp(X, uv, intr)
    P, info = gpu.pnp.solve_pnp_pose(X, uv, intr)
    assert not int(info[3]) & 2                                                   # the start did not fail
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(ro), atol=2e-8)
    np.testing.assert_allclose(P[:, 3], to, atol=2e-7)
    uv2 = uv.copy()
    uv2[:20] += 40.0
    samples = gpu.pnp.draw_samples(80, 64, 6, seed=2)
    P2, mask, best, _ = gpu.pnp.solve_pnp_ransac_pose(X, uv2, intr, 2.0, samples=samples)
    ro2, to2, mask_o, best_o = pnp_np.solve_pnp_ransac(X, uv2, intr, samples, 2.0)
    assert best == best_o and not mask[:20].any() and mask[20:].mean() > 0.9
    np.testing.assert_array_equal(mask, mask_o)
    np.testing.assert_allclose(P2[:, 3], to2, atol=2e-7)


@pytest.mark.gpu
def test_gpu_reproduces_recorded_poses(gpu):
    d, intr, frames = svo_frames(gpu)
    for f, X, uv in frames[:6]:
        P, info = gpu.pnp.solve_pnp_pose(X, uv, intr, world_to_camera(d.poses[0][f - 1][1]))
        assert np.abs(P - world_to_camera(d.poses[0][f][1])).max() < 5e-7


@pytest.mark.gpu
def test_gpu_batched_refine_matches_single(gpu):
    import torch
    X, uv, intr, rvec, tvec = synthetic_problem(200, seed=9)
    dev = torch.device("cuda", 0)
    idx = np.concatenate([np.arange(0, 50), np.arange(40, 200), np.arange(100, 107)]).astype(np.int32)
    ptr = np.array([0, 50, 210, 217], dtype=np.int32)
    P0 = np.tile(np.c_[pnp_np.rodrigues(rvec + 0.02), tvec + 0.1], (3, 1, 1))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dX, duv, dintr, didx, dptr, dP0 = t(X), t(uv), t(intr), t(idx), t(ptr), t(P0)
    out = torch.empty(3, 12, dtype=torch.float64, device=dev)
    info = torch.empty(3, 4, dtype=torch.float64, device=dev)
    L = gpu._lib
    L.check(L.lib().mqs_pnp_refine_dev(dX.data_ptr(), duv.data_ptr(), 200, didx.data_ptr(), dptr.data_ptr(), 3,
                                       dintr.data_ptr(), dP0.data_ptr(), 1, 100, ctypes.c_double(1e-12), out.data_ptr(),
                                       info.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    out = out.cpu().numpy().reshape(3, 3, 4)
    for b in range(3):
        sel = idx[ptr[b]:ptr[b + 1]]
        Pb, _ = gpu.pnp.solve_pnp_pose(X[sel], uv[sel], intr, P0[b], eps=1e-12)
        np.testing.assert_array_equal(out[b], Pb)                 # same kernel, same order of sums: bitwise
    assert (info.cpu().numpy()[:, 2] == [50, 160, 7]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,outliers", [(60, 15), (300, 90), (24, 0)])
def test_gpu_ransac_vs_oracle(n, outliers, gpu):
    X, uv, intr, rvec, tvec = synthetic_problem(n, seed=n + outliers, outliers=outliers)
    samples = gpu.pnp.draw_samples(n, 64, 6, seed=5)
    P, mask, best, info = gpu.pnp.solve_pnp_ransac_pose(X, uv, intr, 2.0, samples=samples)
    ro, to, mask_o, best_o = pnp_np.solve_pnp_ransac(X, uv, intr, samples, 2.0)
    assert best == best_o
    np.testing.assert_array_equal(mask, mask_o)                    # the inlier set: exact
    assert not mask[:outliers].any() and mask[outliers:].mean() > 0.9
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(ro), atol=2e-8)
    np.testing.assert_allclose(P[:, 3], to, atol=2e-7)
    np.testing.assert_allclose(P[:, :3], pnp_np.rodrigues(rvec), atol=5e-3)     # and it is the true pose
    # OpenCV-shaped facade
    K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1.0]])
    r, t, inl = gpu.pnp.solvePnPRansac(X, uv, K, intr[4:8], minInliersCount=10, reprojectionError=2.0, iterationsCount=64, seed=5)
    assert inl.shape == (int(mask.sum()), 1) and inl.dtype == np.int32
    np.testing.assert_array_equal(inl.ravel(), np.nonzero(mask)[0])


@pytest.mark.gpu
def test_gpu_ransac_degenerate_inputs(gpu):
    X, uv, intr, _, _ = synthetic_problem(20, seed=2)
    Xp = X.copy(); Xp[:] = Xp[0]                                    # all points coincide: no valid hypothesis
    P, mask, best, info = gpu.pnp.solve_pnp_ransac_pose(Xp, uv, intr, 2.0, hypotheses=8)
    assert best == -1 and not mask.any()
    with pytest.raises(ValueError):
        gpu.pnp.solve_pnp_ransac_pose(X[:5], uv[:5], intr, 2.0)
