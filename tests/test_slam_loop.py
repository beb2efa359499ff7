"""
End-to-end monocular loop (slam_loop.MonoSlam = slam2.py's handle_new_frame state machine) on a rendered sequence with
known camera motion and known scene geometry (a textured plane) -- BASELINE configs[4].  No reference output exists for
images (parity unpinned); what is asserted is the accuracy against ground truth and the reference's own gates.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_keypoint_mask_and_homography_helpers(mqs):
    L = mqs.slam_loop
    mk = L.keypoint_mask((480, 640), np.array([[5.0, 5.0], [320.2, 240.7], [639, 479]]))
    assert mk[240, 320] == 0 and mk[240, 333] == 1 and mk[0, 0] == 0 and mk[252, 320] == 0 and mk[253, 320] == 1    # centre (320, 240): truncated
    assert L.keypoint_mask((10, 10), np.zeros((0, 2))).all()
    rng = np.random.default_rng(0)
    p1 = rng.uniform(-1, 1, (50, 2))
    Ht = np.array([[1.02, 0.01, 0.03], [-0.02, 0.98, 0.01], [0.01, 0.02, 1.0]])
    q = np.c_[p1, np.ones(50)] @ Ht.T
    assert np.abs(L.homography_dlt(p1, q[:, :2] / q[:, 2:]) - Ht).max() < 1e-12


def test_homography_refinement_is_the_transfer_error_minimiser(mqs):
    """cv2.findHomography(method=0) = normalised DLT + Levenberg-Marquardt on the transfer error (fundam.cpp: estimator.refine).
    On exact correspondences the DLT is already the minimiser; with parallax (correspondences no homography explains) the refined
    H has a smaller transfer error than the DLT's, is a stationary point of it, and scipy's least_squares started from the DLT lands
    on the same H."""
    from scipy.optimize import least_squares
    L = mqs.slam_loop
    rng = np.random.default_rng(3)
    p1 = rng.uniform(-0.6, 0.6, (90, 2))
    Ht = np.array([[1.02, 0.01, 0.03], [-0.02, 0.98, 0.01], [0.01, 0.02, 1.0]])
    q = np.c_[p1, np.ones(len(p1))] @ Ht.T
    exact = q[:, :2] / q[:, 2:]
    assert np.abs(L.find_homography(p1, exact) - Ht).max() < 1e-11
    # two depth layers: a pure translation moves them by different amounts
    p2 = exact + np.where(rng.random(len(p1))[:, None] < 0.4, [0.03, -0.01], [0.0, 0.0]) + rng.normal(0, 2e-4, p1.shape)
    H0 = L.homography_dlt(p1, p2)
    H1 = L.homography_refine(H0, p1, p2)

    def res(h):
        qq = np.c_[p1, np.ones(len(p1))] @ np.append(h, 1.0).reshape(3, 3).T
        return (qq[:, :2] / qq[:, 2:] - p2).ravel()
    e0, e1 = np.sum(res(H0.ravel()[:8]) ** 2), np.sum(res(H1.ravel()[:8]) ** 2)
    assert e1 < e0 * (1 - 1e-4) and H1[2, 2] == 1.0
    ref = least_squares(res, H0.ravel()[:8], method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    assert np.abs(H1.ravel()[:8] - ref).max() < 1e-8
    w0, w1 = np.linalg.svd(H0, compute_uv=False), np.linalg.svd(H1, compute_uv=False)
    assert abs(w0[0] / w0[2] - w1[0] / w1[2]) > 1e-4                              # the digit keyframe_test's 1.04 looks at
    assert np.abs(L.homography_refine(H0, p1[:4], p2[:4]) - H0).max() == 0.0        # four pairs: nothing to refine (count > 4)


def test_keyframe_test_random_sample_is_the_reference_draw(mqs, monkeypatch):
    """keyframe_test's random sample of the tracks (slam2.py:48): `np.random.permutation(n)[:max_points]` from a seeded legacy
    generator -- the same indices as the reference's call draws after np.random.seed; a sample as large as the input is the
    all-points test; the decision is the singular-value ratio of the sampled homography."""
    L = mqs.slam_loop
    rng = np.random.default_rng(1)
    p1 = rng.uniform(100, 500, (120, 2))
    Ht = np.array([[1.0, 0.02, 3.0], [-0.01, 1.0, -2.0], [1e-5, 2e-5, 1.0]])
    q = np.c_[p1, np.ones(len(p1))] @ Ht.T
    p2 = q[:, :2] / q[:, 2:] + rng.normal(0, 0.2, (120, 2))
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    dist = np.zeros(4)
    seen = {}

    def record(p, K_, d_):                                       # stands in for the undistortion: what the test is given
        seen[len(seen)] = np.array(p)
        return np.array(p)
    monkeypatch.setattr(L.camera, "undistort_points", record)
    L.keyframe_test(p1, p2, K, dist, 30, np.random.RandomState(5))
    np.random.seed(5)
    idxs = np.random.permutation(120)[:30]                       # the reference's line, verbatim in effect
    np.testing.assert_array_equal(seen[0], p1[idxs])
    np.testing.assert_array_equal(seen[1], p2[idxs])
    seen.clear()
    a = L.keyframe_test(p1, p2, K, dist)
    b = L.keyframe_test(p1, p2, K, dist, 0)
    assert a == b and len(seen[0]) == 120
    m = L.MonoSlam(K, dist, (480, 640), max_homography_points="reference")
    assert m.max_homography_points == max(4, m.target_keypoints // 4) and L.MonoSlam(K, dist, (480, 640)).max_homography_points == 0


def test_adjuster_screen_residuals_equal_the_oracle_projection(mqs):
    """slam_device._reprojection_residuals (the in-loop adjuster's outlier screen, vectorised numpy) against the BA oracle's
    per-factor projection (oracle/ba_np.project, the Cal3DS2 model of bundle_adjust.cpp:289-298) on random poses, landmarks and a
    full distortion set."""
    from oracle import ba_np
    rng = np.random.default_rng(4)
    P, N, M = 7, 40, 300
    poses = np.zeros((P, 12))
    for k in range(P):
        w = 0.2 * rng.standard_normal(3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        poses[k, :9] = R.reshape(-1)
        poses[k, 9:] = rng.standard_normal(3) * 0.5
    pts = rng.standard_normal((N, 3)) + [0, 0, 8.0]
    calib = np.array([470.0, 481.0, 0.7, 320.0, 240.0, 0.08, -0.02, 0.001, -0.0015])
    lm, ps = rng.integers(0, N, M), rng.integers(0, P, M)
    uv = rng.uniform(0, 640, (M, 2))
    got = mqs.slam_device._reprojection_residuals(poses, pts, calib, lm, ps, uv)
    for k in range(M):
        uvh, _, _, valid = ba_np.project(poses[ps[k]], calib, pts[lm[k]])
        assert valid
        assert got[k] == pytest.approx(np.linalg.norm(uvh - uv[k]), rel=1e-12)


def test_rendered_sequence_is_consistent(mqs):
    seq = mqs.synthetic.PlaneSequence(frames=5)
    img = seq.render(0)
    assert img.shape == (480, 640) and img.dtype == np.uint8 and img.std() > 20
    # a plane point projects where the renderer drew it: the texture value at the projection matches the texture
    P = np.array([[-2.0, 0.5, 0.0], [0.0, -1.0, 0.0]])
    uv = seq.project(0, P)
    n = seq.tex.shape[0]
    for (u, v), X in zip(uv, P):
        t = seq.tex[int(round((X[1] + seq.extent) * (n - 1) / (2 * seq.extent))), int(round((X[0] + seq.extent) * (n - 1) / (2 * seq.extent)))]
        assert abs(float(img[int(round(v)), int(round(u))]) - t) < 12


@pytest.mark.gpu
def test_loop_to_bundle_adjustment_files_end_to_end(gpu, tmp_path):
    """BASELINE configs[4] in full: detect -> track -> pose -> triangulate per keyframe on the GPU, the recorded BA problem
    written as the reference's file set, the CLI-compatible bundle adjuster run on it, `-BA` files written."""
    import subprocess
    import run_slam_loop
    io = gpu.ba_io
    info = io.BundleAdjustmentInfoContainer(str(tmp_path), "loop", 1)
    out = run_slam_loop.run(40, ba_info=info, out_files=(str(tmp_path), "loop", 30))
    assert out["accepted"] == 40
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bundle_adjust.py"), str(tmp_path), "loop", "1", "30", "1", "1", "0", "1", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    fn = io.create_filenames(str(tmp_path), "loop", 1)
    seq = gpu.synthetic.PlaneSequence(frames=40)
    gt = seq.centres()
    before = np.array([p[9:] for _, p in io.load_trajectory(fn.trajectories_in[0])])
    after = np.array([p[9:] for _, p in io.load_trajectory(fn.trajectories_out[0])])
    assert before.shape == after.shape == (40, 3)
    e0 = np.sqrt(np.mean(np.sum((before - gt) ** 2, axis=1)))
    e1 = np.sqrt(np.mean(np.sum((after - gt) ** 2, axis=1)))
    path = np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()
    assert e1 < 0.01 * path and e1 < 1.5 * e0 + 1e-3                              # BA keeps (or improves) the accuracy
    pts = io.load_map(fn.map_out)
    n_init = len(pts) - out["landmarks_triangulated"]
    assert np.median(np.abs(pts[n_init:, 2])) < 0.15                             # the adjusted map stays on the plane
    line = [l for l in r.stdout.splitlines() if l.startswith("cost")][0]
    c0, c1 = float(line.split()[1]), float(line.split()[3])
    assert c1 < c0


@pytest.mark.gpu
def test_end_to_end_loop_on_rendered_sequence(gpu):
    import run_slam_loop
    out = run_slam_loop.run(40)
    assert out["accepted"] == 40 and out["keyframes"] >= 8
    assert out["landmarks_triangulated"] >= 200
    assert out["trajectory_rmse"] < 0.01 * out["path_length"]                 # < 1 % of the path (measured 0.3 %)
    assert out["map_plane_median_abs_z"] < 0.15                                # landmarks lie on the plane z = 0 (depth ~ 9)


def test_loop_host_logic_with_oracle_backends(mqs, c_oracle, monkeypatch, tmp_path):
    """The state machine (gates, bookkeeping, keyframe logic) on the CPU: every GPU call replaced by the oracle's
    restatement of the same step.  Small sequence so that the numpy tracker stays fast."""
    from types import SimpleNamespace
    from oracle import features_np as Fn, pnp_np, harness_np as H
    L = mqs.slam_loop

    def lk(prev, nxt, pts):
        n, s, e = Fn.calc_optical_flow_pyr_lk(prev, nxt, pts)
        return n, s.reshape(-1, 1), e.reshape(-1, 1)

    def intr_of(K, dist):
        return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], dist[0], dist[1], dist[2], dist[3], 0.0])

    def solve_pnp(objp, imgp, K, dist, rvec=None, tvec=None, useExtrinsicGuess=False):
        o, m = np.asarray(objp, np.float64), np.asarray(imgp, np.float64)
        if useExtrinsicGuess:
            rv, tv, _, _ = pnp_np.solve_pnp(o, m, intr_of(K, dist), np.asarray(rvec).ravel(), np.asarray(tvec).ravel())
        else:
            rv, tv, _, _ = pnp_np.solve_pnp(o, m, intr_of(K, dist))
        return True, rv.reshape(3, 1), tv.reshape(3, 1)

    def solve_pnp_ransac(objp, imgp, K, dist, minInliersCount=0, reprojectionError=2.0, seed=0):
        o, m = np.asarray(objp, np.float64), np.asarray(imgp, np.float64)
        samples = mqs.pnp.draw_samples(len(o), 16, 6, seed)
        rv, tv, mask, best = pnp_np.solve_pnp_ransac(o, m, intr_of(K, dist), samples, reprojectionError)
        return rv.reshape(3, 1), tv.reshape(3, 1), np.nonzero(mask)[0].astype(np.int32).reshape(-1, 1)

    def undistort(p, K, dist):
        x, y = H.undistort_normalized((p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1], *dist)
        return np.stack([x, y], 1)

    def reproj(objp, imgp, K, dist, rvec, tvec):
        uv = pnp_np.project(np.asarray(rvec).ravel(), np.asarray(tvec).ravel(), objp, intr_of(K, dist))
        return float(np.sqrt(((uv - imgp) ** 2).sum() / len(imgp))), uv

    def tri(u0, P0, u1, P1):
        return c_oracle.iterative_LS_triangulation(np.stack([u0, u1]), np.stack([P0[:3], P1[:3]]))

    monkeypatch.setattr(L, "features", SimpleNamespace(calcOpticalFlowPyrLK=lk, goodFeaturesToTrack=(
        lambda img, n, q, md, c=None, mask=None: Fn.good_features_to_track(img, n, q, md, mask) if n else np.zeros((0, 2), np.float32))))
    monkeypatch.setattr(L, "pnp", SimpleNamespace(solvePnP=solve_pnp, solvePnPRansac=solve_pnp_ransac,
                                                 Rodrigues=lambda r: pnp_np.rodrigues(np.asarray(r).ravel())))
    monkeypatch.setattr(L, "camera", SimpleNamespace(undistort_points=undistort, reprojection_error=reproj))
    monkeypatch.setattr(L, "triangulation", SimpleNamespace(iterative_LS_triangulation=tri))

    frames = 14
    seq = mqs.synthetic.PlaneSequence(image_size=(320, 240), f=240.0, frames=4 * frames)     # quarter-speed motion
    gx, gy = np.meshgrid(np.linspace(-5.0, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 14) & (imgp[:, 0] < seq.W - 14) & (imgp[:, 1] > 14) & (imgp[:, 1] < seq.H - 14)
    io = mqs.ba_io
    info = io.BundleAdjustmentInfoContainer(str(tmp_path), "loop", 1)
    slam = L.MonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=3, ba_info=info)
    slam.start(seq.render(0), objp[vis], imgp[vis])
    rets = [slam.handle_new_frame(seq.render(k)) for k in range(1, frames)]
    assert all(r in (1, 2) for r in rets) and rets.count(2) >= 1                 # every frame accepted, some keyframes
    traj, gt = slam.trajectory(), seq.centres()[:frames]
    assert np.linalg.norm(traj - gt, axis=1).max() < 0.08                        # path of ~0.35 units at depth ~9
    new = slam.objp[int(vis.sum()):]
    assert len(new) >= 10 and np.median(np.abs(new[:, 2])) < 0.6                 # short baselines: coarse, but on the plane
    assert (slam.lm >= -1).all() and slam.lm.max() < len(slam.objp)
    # what the loop recorded for the bundle adjuster (slam2.py:743-865): written in the reference's file set, read back by
    # the reader of the CLI tool, accepted by both of its validators, and turned into a well-posed sparse problem
    info.write_all()
    info.write_noise(point2D=1.0)
    fn = io.create_filenames(str(tmp_path), "loop", 1)
    io.save_slam_output(fn, 30, slam.projection_matrices(), slam.objp)
    data = io.load_data(fn, 30)
    io.validate_data_integrity(data, 1)
    ok, _ = io.validate_sufficiently_constrained(data, True)
    assert ok
    assert len(data.point3DAddedIdxs) == frames and sum(len(a) for a in data.point3DAddedIdxs) == len(slam.objp)
    assert sorted(i for a in data.point3DAddedIdxs for i in a) == list(range(len(slam.objp)))
    assert sum(len(o) for o in data.odometry) == rets.count(2)                   # one odometry edge per keyframe
    pr = io.build_sparse_problem(data, use_odometry=True)
    assert len(pr.poses) == frames and len(pr.points) == len(slam.objp)
    # every recorded observation reprojects near its measurement through the recorded pose and landmark
    from oracle import ba_np
    res = []
    for j in range(len(pr.points)):
        for k in range(pr.obs_ptr[j], pr.obs_ptr[j + 1]):
            uv, _, _, front = ba_np.project(pr.poses[pr.obs_pose[k]], pr.calib[0], pr.points[j])
            assert front
            res.append(np.linalg.norm(uv - pr.obs_uv[k]))
    # (all but a handful: a corner followed through the border-extended pyramid until its window has left the image -- as OpenCV's
    # tracker follows it -- is measured on mirrored content in its last frames, and nothing in slam2.py's flow takes it back out)
    assert len(res) > 200 and np.median(res) < 1.0 and np.percentile(res, 98) < 12.0
    # every landmark added at a keyframe is observed in every frame since the previous keyframe (>= 2 views)
    assert min(pr.obs_ptr[j + 1] - pr.obs_ptr[j] for j in range(int(vis.sum()), len(pr.points))) >= 2


@pytest.mark.gpu
def test_device_resident_loop_on_rendered_sequence(gpu):
    """slam_device.DeviceMonoSlam (csrc/slam_frame.hip): the same state machine with its state on the GPU and one library
    call per frame -- the accuracy bars of the host-driven loop, and the two loops agree on what they build (they draw
    different RANSAC samples, so not bit for bit: same number of accepted frames, keyframes within two, trajectories within
    0.2 % of the path of each other)."""
    import run_slam_loop
    dev = run_slam_loop.run_device(60)
    host = run_slam_loop.run(60)
    assert dev["accepted"] == 60 and dev["keyframes"] >= 8
    assert dev["landmarks_triangulated"] >= 200
    assert dev["trajectory_rmse"] < 0.01 * dev["path_length"]
    assert dev["map_plane_median_abs_z"] < 0.15
    assert abs(dev["keyframes"] - host["keyframes"]) <= 2
    assert abs(dev["trajectory_rmse"] - host["trajectory_rmse"]) < 0.002 * dev["path_length"]
    assert dev["tracks_at_the_end"] >= 100
    # the same path in 40 frames (half again the image motion per frame, a keyframe every second frame): the plain loop is at the
    # edge of what it tracks there -- which keyframes it takes decides between 0.03 and 0.08 of error (profiles/r04/15) -- and the
    # two loops still end within a percent of the path of each other
    dev, host = run_slam_loop.run_device(40), run_slam_loop.run(40)
    assert dev["accepted"] == 40 and host["accepted"] == 40 and abs(dev["keyframes"] - host["keyframes"]) <= 2
    assert dev["trajectory_rmse"] < 0.02 * dev["path_length"]
    assert abs(dev["trajectory_rmse"] - host["trajectory_rmse"]) < 0.01 * dev["path_length"]


@pytest.mark.gpu
def test_device_loop_with_bundle_adjustment_per_keyframe(gpu):
    """BASELINE configs[4] as written: detect -> track -> triangulate -> bundle-adjust per keyframe INSIDE the device-resident loop,
    with the matcher re-associating the top-up's corners with landmarks that lost their track (slam.py:81-127's match_OF_based).
    The adjusted trajectory beats the plain loop's AND the post-hoc adjustment of the recorded run (tools/run_slam_loop.py --ba:
    0.0056-0.0057 on this sequence); the poses the frames were first given (online) already profit from the adjusted map."""
    import run_slam_loop
    plain = run_slam_loop.run_device(60)
    ba = run_slam_loop.run_device(60, bundle_adjust="keyframe")
    both = run_slam_loop.run_device(60, bundle_adjust="keyframe", reassociate=True)
    for out in (ba, both):
        rep = out["bundle_adjust_per_keyframe"]
        assert out["accepted"] == 60 and out["keyframes"] >= 12
        assert rep["adjustments"] >= out["keyframes"] - 1
        assert rep["trajectory_rmse_adjusted"] <= 0.0057                               # the post-hoc adjustment's figure
        assert rep["trajectory_rmse_adjusted"] < 0.3 * plain["trajectory_rmse"]
        assert rep["trajectory_rmse_online"] < 0.6 * plain["trajectory_rmse"]
        assert out["trajectory_max_err"] < 0.5 * plain["trajectory_max_err"]            # no frame left behind by an outlier track
        assert out["map_plane_median_abs_z"] < plain["map_plane_median_abs_z"]          # the map is flatter too
        assert rep["last"]["cost_after"] <= rep["last"]["cost_before"]
    assert both["corners_reassociated_with_lost_landmarks"] > 20                        # the matcher has work in this loop
    # re-association keeps landmarks alive instead of duplicating them: no more landmarks per keyframe than without it
    assert both["landmarks_triangulated"] / both["keyframes"] <= 1.05 * ba["landmarks_triangulated"] / ba["keyframes"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 3])
def test_adjustment_in_the_loop_survives_a_grossly_mistracked_corner(gpu, seed):
    """The same rendering in 40 frames (half again the image motion per frame).  At frame 16 a corner that slid along an edge
    enters the adjustment with a residual of thousands of pixels (cost 1.9e7 against ~60): with these two RANSAC seeds the
    round's first version let Levenberg-Marquardt spread it over the poses before the screen saw it, wrote the result back, and
    never recovered (trajectory RMSE 0.037 / 0.043, worse than the plain loop's 0.032; profiles/r04/15).  The screen in front of
    the adjustment (ba_gross_pixels) keeps such a landmark out; the adjusted trajectory is then the same for every seed."""
    import run_slam_loop
    plain = run_slam_loop.run_device(40, seed=seed)
    ba = run_slam_loop.run_device(40, seed=seed, bundle_adjust="keyframe")
    rep = ba["bundle_adjust_per_keyframe"]
    assert ba["accepted"] == 40
    assert rep["trajectory_rmse_adjusted"] < 0.012 and rep["trajectory_rmse_adjusted"] < 0.4 * plain["trajectory_rmse"]
    assert rep["trajectory_rmse_online"] < plain["trajectory_rmse"]
    assert rep["last"]["cost_after"] < 400.0                                           # (the stuck runs ended at 420-460)


def _rendered(gpu, frames):
    import torch
    seq = gpu.synthetic.PlaneSequence(frames=max(60, frames))      # (up to 60 frames: the first ones of the 60-frame sweep; beyond: the sweep in that many frames)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
    return seq, objp[vis], imgp[vis], imgs


@pytest.mark.gpu
def test_device_loop_launch_forms_give_the_same_run(gpu, monkeypatch):
    """The loop's fused launches against the forms they replaced, on the rendered sequence incl. one broken frame (a rejection): the
    track filter inside the RANSAC hypotheses' launch (every hypothesis' wavefront compacts the tracks for itself; workgroup 0 writes
    the frame's state) against the one-workgroup filter kernel + the hypotheses (MQS_SLAM_FUSED_FILTER=0); the tracker's pyramid in
    one launch against one per level (MQS_LK_PYRAMID_PER_LEVEL=1); the decision kernel's two workgroups -- the keyframe test beside the
    pose refinement -- against one doing both in turn (MQS_SLAM_DECIDE_SPLIT=0).  Same decisions, same poses bit for bit, same tracks."""
    import torch
    seq, objp, imgp, imgs = _rendered(gpu, 30)
    imgs = list(imgs)
    imgs[17] = torch.zeros_like(imgs[17])                        # a frame the tracker loses: rejected, the loop goes on

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=3)
        slam.start(imgs[0], objp, imgp)
        rets = [slam.handle_new_frame(imgs[k]) for k in range(1, 30)]
        slam.finish()
        poses = [None if P is None else np.array(P) for P in slam.poses]
        tracks = [np.array(a) for a in slam.tracks()]
        slam.close()
        for k in env:
            monkeypatch.delenv(k)
        return rets, poses, tracks
    base = run()
    assert 0 in base[0] and 2 in base[0]
    for env in ({"MQS_SLAM_FUSED_FILTER": "0"}, {"MQS_LK_PYRAMID_PER_LEVEL": "1"}, {"MQS_SLAM_DECIDE_SPLIT": "0"}):
        other = run(**env)
        assert other[0] == base[0], env
        for a, b in zip(base[1], other[1]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), env
        for a, b in zip(base[2], other[2]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("window, history, legs", [(None, None, 0), (3, None, 3), (2, 2, 2)], ids=["every frame", "default selection", "bounded history"])
def test_resident_adjuster_equals_its_host_built_twin_at_every_keyframe(gpu, window, history, legs):
    """`mqs_slam_bundle_adjust(_window)` (csrc/slam_ba.hip: the whole adjustment in one persistent launch on the resident log, map and
    trajectory) against round 4's host-built path on the SAME state, behind every keyframe of the rendered sequence: the twin builds
    the CSR problem from the log read back, runs `sparse_ba.SparseBundleAdjuster` (the reference-pinned kernels of ba_sparse.hip) with
    the same screens and writes nothing back; then the device call runs.  Same poses to 1e-9 (different summation orders: measured
    5e-15), same landmarks as float32, same landmarks retired, same pass / iteration counts, same final cost to 1e-9 relative.
    Three forms: every accepted frame (the reference's whole-graph adjustment); the default SELECTION (round 6: the keyframes so far
    + every frame since the third keyframe from the end, the frames left out carried along with the keyframe in front of them);
    a selection with a bounded history -- frame 0 leaves the problem, the gauge becomes two pose priors at current values and the
    landmarks seen from frames outside keep a prior.  The residual screen behind a whole adjustment only (round 5's form), behind
    every three iterations (the default) and behind every two."""
    seq, objp, imgp, imgs = _rendered(gpu, 45)
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", ba_check=True, reassociate=True,
                                          ba_window_keyframes=window, ba_history_keyframes=history, ba_screen_iterations=legs)
    assert slam.ba_engine == "device"
    slam.start(imgs[0], objp, imgp)
    for k in range(1, 45):
        assert slam.handle_new_frame(imgs[k]) in (1, 2)
    slam.finish()
    assert len(slam.ba_checks) >= 8 and not slam.ba_fallbacks
    for chk in slam.ba_checks:
        h, d = chk["host_report"], chk["device_report"]
        assert chk["host_poses"].shape == chk["device_poses"].shape                # every accepted frame, selected or carried
        assert np.abs(chk["host_poses"] - chk["device_poses"]).max() < 1e-9
        n = len(chk["device_points"])
        assert np.array_equal(chk["host_points"][:n].astype(np.float32), chk["device_points"].astype(np.float32))
        m = min(len(chk["host_retired"]), len(chk["device_retired"]))
        assert np.array_equal(chk["host_retired"][:m], chk["device_retired"][:m]) and not chk["host_retired"][m:].any()
        for key in ("poses", "first_pose_of_the_window", "accepted_frames", "landmarks", "landmarks_adjusted", "observations", "passes",
                    "landmarks_screened_out", "lm_iterations"):
            assert h[key] == d[key], (key, h, d)
        assert abs(h["cost_after"] - d["cost_after"]) <= 1e-9 * max(1.0, h["cost_after"])
        assert abs(h["cost_before"] - d["cost_before"]) <= 1e-9 * max(1.0, h["cost_before"])
        assert d["repeated_observations_left_out"] == 0
    last = slam.ba_checks[-1]["device_report"]
    if window is None:
        assert all(c["device_report"]["poses"] == c["device_report"]["accepted_frames"] for c in slam.ba_checks)
    else:
        assert last["poses"] < last["accepted_frames"]                             # a selection was made ...
    if history is not None:
        assert last["first_pose_of_the_window"] > 0                                # ... and here frame 0 has left it
    slam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["device", "host"])
def test_loop_goes_on_after_finish(gpu, engine):
    """finish() adjusts once more when frames followed the last keyframe; the live state's base-keyframe pose must then be the last
    KEYFRAME's adjusted pose, not the last frame's (the tracks' base points belong to the keyframe): a loop that is continued after
    finish() triangulates its next keyframe against it.  (Round 4 wrote the last frame's pose into both.)"""
    seq, objp, imgp, imgs = _rendered(gpu, 60)
    gt = seq.centres()

    def run(stop):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", ba_engine=engine)
        slam.start(imgs[0], objp, imgp)
        for k in range(1, 60):
            assert slam.handle_new_frame(imgs[k]) in (1, 2)
            if k == stop:
                assert slam.keyframes[-1] != k                      # (frames behind the last keyframe: finish() has work)
                slam.finish()
        slam.finish()
        c = np.array([-P[:, :3].T @ P[:, 3] for P in slam.poses])
        slam.close()
        return float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1))))
    straight, resumed = run(None), run(30)
    assert straight < 0.0057 and resumed < 0.0057 and abs(resumed - straight) < 0.002


@pytest.mark.gpu
def test_runs_longer_than_the_adjusters_reach_stay_on_the_device_engine(gpu):
    """The resident adjuster stages every camera of its PROBLEM in LDS: 256 poses at most.  Until round 5 a run with more accepted
    frames fell back to the host-built engine (numpy + ~40 launches per trial); now the problem's poses are a selection of the
    accepted frames -- every keyframe and the latest frames that fit -- and the run stays on the device.  Here with the reach set to
    25 poses on the 60-frame rendering and the whole-graph form asked for (no window): every adjustment on the device, none with more
    than 25 poses, same accuracy bar as the unbounded run."""
    seq, objp, imgp, imgs = _rendered(gpu, 60)
    gt = seq.centres()
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", ba_window_keyframes=None)
    slam.BA_DEVICE_MAX_POSES = 25
    slam.start(imgs[0], objp, imgp)
    for k in range(1, 60):
        assert slam.handle_new_frame(imgs[k]) in (1, 2)
    slam.finish()
    assert all(r["engine"] == "device" for r in slam.ba_reports) and slam.ba_engine == "device" and not slam.ba_fallbacks
    assert max(r["poses"] for r in slam.ba_reports) <= 25 and slam.ba_reports[-1]["accepted_frames"] == 60
    assert slam.ba_reports[0]["poses"] == slam.ba_reports[0]["accepted_frames"]     # (while everything fits: every frame)
    c = np.array([-P[:, :3].T @ P[:, 3] for P in slam.poses])
    assert float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))) < 0.0057
    slam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("code", [-6, -7], ids=["timeout", "capacity"])
def test_an_adjustment_the_resident_adjuster_gives_up_on_goes_to_the_host_engine(gpu, code):
    """A launch whose grid-wide wait gave up (MQS_E_TIMEOUT) or whose resident lists do not hold the problem (MQS_E_CAPACITY: more
    co-observations than the hit lists hold -- a hovering camera) has written nothing back.  Until round 5 that raised out of
    handle_new_frame and the run died; now the host-built engine takes the adjustment over where it stands -- retired landmarks and
    odometry edges come from the device once (an edge the failed call brought is NOT kept there: the host measures it itself) --
    and the run goes on.  The failure is injected by the library's test hook."""
    from mqslam_amd import _lib
    seq, objp, imgp, imgs = _rendered(gpu, 60)
    gt = seq.centres()
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    slam.start(imgs[0], objp, imgp)
    for k in range(1, 60):
        if len(slam.ba_reports) == 5 and slam.ba_engine == "device":
            _lib.check(_lib.lib().mqs_debug_slam_ba_fail_next(slam._h, code))
        assert slam.handle_new_frame(imgs[k]) in (1, 2)
    slam.finish()
    engines = [r["engine"] for r in slam.ba_reports]
    assert engines[:5] == ["device"] * 5 and set(engines[5:]) == {"host"} and slam.ba_engine == "host"
    assert len(slam.ba_fallbacks) == 1 and slam.ba_fallbacks[0]["code"] == code and "injected" in slam.ba_fallbacks[0]["message"]
    assert len(slam._odo) == len(slam.keyframes) - 1                               # every keyframe's edge once: the device's, then the host's
    assert [o[1] for o in slam._odo] == slam._kf_pose[1:]
    c = np.array([-P[:, :3].T @ P[:, 3] for P in slam.poses])
    assert float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))) < 0.0057
    slam.close()


@pytest.mark.gpu
def test_resident_adjuster_fails_fast_when_its_workgroups_cannot_all_be_resident(gpu):
    """The adjuster's workgroups meet at spinning grid-wide barriers, so every one of them has to be resident: a grid larger than
    the device holds (compute units x workgroups of this kernel per unit, `mqs_slam_ba_resident_groups`) used to spin for 2 s and
    return MQS_E_TIMEOUT.  Now the library sizes its own grid from the device and refuses an explicit request beyond it at once."""
    import ctypes, time
    from mqslam_amd import _lib
    seq, objp, imgp, imgs = _rendered(gpu, 12)
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    slam.start(imgs[0], objp, imgp)
    for k in range(1, 12):
        slam.handle_new_frame(imgs[k])
    slam._bundle_adjust()                                                          # (warm: code object, occupancy query)
    g = ctypes.c_int32(0)
    _lib.check(_lib.lib().mqs_slam_ba_resident_groups(slam._h, ctypes.byref(g)))
    assert 1 <= g.value <= 4096
    rep = np.zeros(16)
    q = slam._ba_params(False, 0, 0)
    q.workgroups = g.value + 1
    t0 = time.perf_counter()
    rc = _lib.lib().mqs_slam_bundle_adjust(slam._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), None, 0)
    dt = time.perf_counter() - t0
    assert rc == -1 and dt < 0.010, (rc, dt)                                       # MQS_E_ARG, not a 2 s wait
    assert "resident" in _lib.lib().mqs_last_error().decode()
    q.workgroups = g.value                                                         # as many as fit: runs
    assert _lib.lib().mqs_slam_bundle_adjust(slam._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), None, 0) == 0 and rep[0] == 0.0
    q.workgroups = 0
    n_before = len(slam.ba_reports)
    slam._bundle_adjust()
    assert len(slam.ba_reports) == n_before + 1 and slam.ba_reports[-1]["engine"] == "device"
    slam.close()


@pytest.mark.gpu
def test_resident_adjuster_refuses_what_it_cannot_hold(gpu):
    """Errors of `mqs_slam_bundle_adjust`, none of them silent: a log that overflowed (round 4 clamped the count and adjusted a
    problem whose newest frames had no observations), bad parameters, a handle without a log."""
    import ctypes
    from mqslam_amd import _lib
    seq, objp, imgp, imgs = _rendered(gpu, 12)
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", ba_log_capacity=1024)
    slam.start(imgs[0], objp, imgp)
    with pytest.raises(RuntimeError, match="observation log is full"):
        for k in range(1, 12):
            slam.handle_new_frame(imgs[k])
            slam._bundle_adjust()
    with pytest.raises(RuntimeError, match="observation log is full"):
        slam.read_log()
    slam.close()
    ok = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    ok.start(imgs[0], objp, imgp)
    q = ok._ba_params(False, 0, 0)
    q.max_passes = 0
    rep = np.zeros(16)
    assert _lib.lib().mqs_slam_bundle_adjust(ok._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), None, 0) == -1      # MQS_E_ARG
    q = ok._ba_params(True, 3, 2)                                   # an edge that does not run forward / beyond the accepted frames
    assert _lib.lib().mqs_slam_bundle_adjust(ok._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), None, 0) == -1
    ok._bundle_adjust()                                             # one frame, the start-up landmarks: a problem of six unknowns
    assert ok.ba_reports[-1]["poses"] == 1 and ok.ba_reports[-1]["cost_after"] <= ok.ba_reports[-1]["cost_before"]
    ok.close()
    plain = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1)
    plain.start(imgs[0], objp, imgp)
    assert _lib.lib().mqs_slam_bundle_adjust(plain._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), None, 0) == -1
    plain.close()


@pytest.mark.gpu
def test_selection_forms_of_the_adjustment_in_the_device_loop(gpu):
    """The problem's poses as a selection of the accepted frames (round 6, `mqs_slam_bundle_adjust_window`): the default (every keyframe
    so far + every frame since the third keyframe from the end) against the whole-graph form and against a bounded history, on the
    rendered 60-frame sequence -- fewer poses per adjustment, the same accuracy.  On 200 frames of the reference's example sequence,
    16 seeds (profiles/r06): every frame 5.1 mm median / 8.0 worst at 1 575 frames/s, the default 4.8 / 7.2 at 3 130."""
    seq, objp, imgp, imgs = _rendered(gpu, 60)
    gt = seq.centres()

    def run(**kw):
        s = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", **kw)
        s.start(imgs[0], objp, imgp)
        rets = [s.handle_new_frame(imgs[k]) for k in range(1, 60)]
        s.finish()
        c = np.array([-P[:, :3].T @ P[:, 3] for P in s.poses])
        out = {"ok": all(r in (1, 2) for r in rets) and all(r["engine"] == "device" for r in s.ba_reports),
               "rmse": float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))), "poses": [r["poses"] for r in s.ba_reports],
               "first": [r["first_pose_of_the_window"] for r in s.ba_reports], "keyframes": len(s.keyframes)}
        s.close()
        return out
    full, default, bounded = run(ba_window_keyframes=None), run(), run(ba_window_keyframes=3, ba_history_keyframes=4)
    assert full["ok"] and default["ok"] and bounded["ok"] and default["keyframes"] >= 10
    assert full["poses"][-1] == 60 and default["poses"][-1] < 30 and max(bounded["poses"]) < 30       # measured 21 and <= 20
    assert set(default["first"]) == {0} and bounded["first"][-1] > 0                                   # every keyframe stays / the oldest leave
    # measured: 0.0025 (every frame), 0.0021 (default), 0.0019-0.0025 (bounded history); the plain loop: 0.0156
    assert full["rmse"] < 0.0057 and default["rmse"] < 0.0057 and default["rmse"] < full["rmse"] + 0.001 and bounded["rmse"] < 0.0057


@pytest.mark.gpu
def test_resident_adjusters_first_step_against_the_oracle_on_the_logged_problem(gpu):
    """The resident adjuster has a product-side twin (above); this is its check against the ORACLE (oracle/ba_np.py: explicit 2 x 6 /
    2 x 3 Jacobians, dense blocks, GTSAM 3.2.1's factor arithmetic restated) on the adjuster's own problem: the log, map and poses
    are read back in front of an adjustment, the problem is built HERE from the rules include/mqslam.h states (which landmarks take
    part, which observations, the gauge priors, the odometry edges), the oracle linearises it, adds the prior and between terms, damps
    with the first trial's lambda and solves; the library call then runs ONE Levenberg-Marquardt iteration without screens.  Cost at
    the start, cost after the step, every adjusted pose and landmark: equal.  Both forms: every frame, and a selection whose frame 0
    has left (two anchored poses, landmark priors at current values, carried frames)."""
    import ctypes
    from oracle import ba_np
    from mqslam_amd import _lib
    from mqslam_amd.bundle_adjustment import pose_from_world_to_camera, LM_LAMBDA_INITIAL
    seq, objp, imgp, imgs = _rendered(gpu, 40)
    for window, history, stop in ((None, None, 3), (2, 1, 6)):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", ba_window_keyframes=window,
                                              ba_history_keyframes=history)
        slam.ba_iterations, slam.ba_max_passes, slam.ba_gross_pixels = 1, 1, 0.0     # one iteration, no screen before or after
        inner, state = slam._bundle_adjust_device, {}

        def spy():
            if len(slam.ba_reports) == stop - 1:                                     # in front of the adjustment under test: the state it will see
                rep = np.zeros(40)
                _lib.check(_lib.lib().mqs_slam_flush(slam._h, rep.ctypes.data_as(_lib.c_f64p)))
                if rep[24] != 0.0 and slam._pending_keyframe is not None:            # (the keyframe's refined pose, as the trajectory holds it)
                    slam.poses[slam._pending_keyframe] = rep[28:40].reshape(3, 4).copy()
                state.update(log=slam.read_log(), pts=slam.objp.astype(np.float64), poses={f: slam.poses[f].copy() for f in slam._accepted},
                             accepted=list(slam._accepted), sel=slam._select_poses(), bad=slam.retired_landmarks(), kf_pose=list(slam._kf_pose))
                n = ctypes.c_int32(0)
                L = _lib.lib()
                _lib.check(L.mqs_slam_read_ba_edges(slam._h, None, None, None, 0, ctypes.byref(n)))
                fr, to, meas = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros((n.value, 12))
                if n.value:
                    _lib.check(L.mqs_slam_read_ba_edges(slam._h, fr.ctypes.data_as(_lib.c_i32p), to.ctypes.data_as(_lib.c_i32p), meas.ctypes.data_as(_lib.c_f64p), n.value, ctypes.byref(n)))
                state.update(odo=(fr, to, meas))
            return inner()
        slam._bundle_adjust_device = spy
        slam.start(imgs[0], objp, imgp)
        for k in range(1, 40):
            slam.handle_new_frame(imgs[k])
            if len(slam.ba_reports) == stop:
                break
        assert len(slam.ba_reports) == stop and state
        report = slam.ba_reports[-1]
        got_poses = {f: slam.poses[f].copy() for f in slam._accepted}
        got_pts = slam.objp.astype(np.float64)
        # ---- the problem, from the rules (include/mqslam.h: mqs_slam_bundle_adjust / _window) ----
        lm, ps, uv = state["log"]
        pts, acc = state["pts"], state["accepted"]
        N, P_all = len(pts), len(acc)
        sel = np.arange(P_all) if state["sel"] is None else np.asarray(state["sel"])
        P, first = len(sel), int(sel[0])
        pmap = -np.ones(P_all, int)
        pmap[sel] = np.arange(P)
        has = (lm >= 0) & (lm < N)
        out_cnt = np.bincount(lm[has & (pmap[ps] < 0)], minlength=N)
        m = slam.ba_border_margin
        inside = (uv[:, 0] >= m) & (uv[:, 0] <= seq.W - 1 - m) & (uv[:, 1] >= m) & (uv[:, 1] <= seq.H - 1 - m)
        fac = has & (pmap[ps] >= 0) & inside
        n_in = np.bincount(lm[fac], minlength=N)
        n0 = len(objp) if first == 0 else 0
        use = (((n_in >= 1) & (n_in + out_cnt >= slam.ba_min_observations)) | (np.arange(N) < n0)) & ~state["bad"][:N]
        fac &= use[np.clip(lm, 0, N - 1)]
        order = np.argsort(lm[fac], kind="stable")
        obs_pose, obs_uv = pmap[ps[fac]][order], uv[fac][order]
        obs_ptr = np.concatenate([[0], np.cumsum(np.bincount(lm[fac], minlength=N))])
        poses = np.stack([pose_from_world_to_camera(state["poses"][acc[j]]) for j in sel])
        if first == 0:
            poses[0] = pose_from_world_to_camera(slam._pose0)                         # the run's first pose starts at its start-up estimate
        prior_w, prior_xyz = np.zeros(N), pts.copy()
        prior_w[:n0], prior_xyz[:n0] = 1.0 / slam.ba_point_sigma ** 2, objp[:n0].astype(np.float32)
        if first != 0:
            prior_w[(out_cnt > 0)] = 1.0 / slam.ba_window_point_sigma ** 2
        anchors = [0]
        if first != 0:
            nxt = [j for j in sel[1:] if j in state["kf_pose"]]
            if nxt:
                anchors.append(int(pmap[nxt[0]]))
        calib = np.array([[seq.K[0, 0], seq.K[1, 1], 0.0, seq.K[0, 2], seq.K[1, 2], *np.asarray(seq.dist).reshape(-1)[:4]]])
        sigma = np.array([slam.ba_pixel_sigma])
        fr, to, meas = state["odo"]
        # (the edge this call brings: base keyframe -> this keyframe, measured from the poses as they stand, slam2.py:681-687)
        kf = state["kf_pose"]
        if len(kf) >= 2 and kf[-1] == P_all - 1 and (len(to) == 0 or to[-1] != kf[-1]):
            P1, P0 = np.vstack([state["poses"][acc[kf[-1]]], [0, 0, 0, 1.0]]), np.vstack([state["poses"][acc[kf[-2]]], [0, 0, 0, 1.0]])
            fr, to = np.append(fr, kf[-2]), np.append(to, kf[-1])
            meas = np.vstack([meas.reshape(-1, 12), pose_from_world_to_camera((P1 @ np.linalg.inv(P0))[:3])])
        keep = (pmap[fr] >= 0) & (pmap[to] >= 0)
        ofr, oto, omeas = pmap[fr[keep]], pmap[to[keep]], meas[keep]
        osig = np.tile(np.asarray(slam.ba_odometry_sigmas), (len(ofr), 1))
        psig = np.tile(np.asarray(slam.ba_pose_sigmas), (len(anchors), 1))
        lam = LM_LAMBDA_INITIAL
        assert report["poses"] == P and report["landmarks_adjusted"] == int(use.sum()) and report["observations"] == int(fac.sum())
        assert report["odometry_edges"] == len(fr) and (first == 0) == (window is None)
        # ---- the oracle's first trial ----
        cam = np.zeros(P, int)
        # (GTSAM 3.2.1's default damping, the adjuster's: lambda * I on every variable -- landmark blocks before their elimination, pose block after)
        S, g, c0, _, pieces = ba_np.sparse_linearize(poses, cam, calib, sigma, pts, obs_ptr, obs_pose, obs_uv, prior_w, prior_xyz, lam, additive=True)
        Hp, gp, cp = ba_np.sparse_pose_prior_terms(poses, anchors, poses[anchors], psig)
        Hb, gb, cb = ba_np.sparse_between_terms(poses, ofr, oto, omeas, osig)
        A = S + Hp + Hb
        d = np.linalg.solve(A + lam * np.eye(len(A)), g + gp + gb)
        new_poses = np.stack([ba_np.retract_pose(poses[j], d[6 * j:6 * j + 6]) for j in range(P)])
        new_pts = pts + ba_np.sparse_backsub(pieces, d) * use[:, None]
        c1 = (ba_np.sparse_cost(new_poses, cam, calib, sigma, new_pts, obs_ptr, obs_pose, obs_uv, prior_w, prior_xyz)
              + ba_np.sparse_pose_prior_terms(new_poses, anchors, poses[anchors], psig)[2] + ba_np.sparse_between_terms(new_poses, ofr, oto, omeas, osig)[2])
        assert report["lm_trials"] == 1 and report["lm_iterations"] == 1 and c1 < c0 + cp + cb
        assert report["cost_before"] == pytest.approx(c0 + cp + cb, rel=1e-9)
        assert report["cost_after"] == pytest.approx(c1, rel=1e-7)
        for k, j in enumerate(sel):
            R, c = new_poses[k, :9].reshape(3, 3), new_poses[k, 9:]
            np.testing.assert_allclose(got_poses[acc[j]], np.hstack([R.T, (-R.T @ c)[:, None]]), atol=2e-9)
        step = np.abs(new_pts - pts).max()
        assert step > 1e-5                                                           # (the step is not nothing)
        assert np.abs(got_pts[use] - new_pts[use]).max() < 2e-6                      # (the map holds float32 values, slam2.py:19)
        assert np.array_equal(got_pts[~use], pts[~use])
        # the frames the selection left out: carried along with the last selected pose in front of them
        h = lambda M: np.vstack([M, [0, 0, 0, 1.0]])
        a = 0
        for j in range(first, P_all):
            if pmap[j] >= 0:
                a = j
                continue
            want = (h(state["poses"][acc[j]]) @ np.linalg.inv(h(state["poses"][acc[a]])) @ h(got_poses[acc[a]]))[:3]
            np.testing.assert_allclose(got_poses[acc[j]], want, atol=1e-12)
        for j in range(first):                                                       # in front of the selection: untouched
            assert np.array_equal(got_poses[acc[j]], state["poses"][acc[j]])
        slam.close()


@pytest.mark.gpu
def test_four_hundred_frames_stay_on_the_device_engine(gpu):
    """More accepted frames than the resident adjuster holds poses (256; the reference's committed runs are 376 and 881 poses long,
    Work/SLAM/datasets/ICL_NUIM/*/traj_out.cam0-slam2.txt): the 400-frame rendering of the test sequence with the default selection and
    with the whole-graph form asked for -- every adjustment `engine: device`, no fall-back, the accuracy of the 60-frame run."""
    seq, objp, imgp, imgs = _rendered(gpu, 400)
    gt = seq.centres()
    for kw in ({}, {"ba_window_keyframes": None}):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe", reassociate=True, **kw)
        slam.start(imgs[0], objp, imgp)
        for k in range(1, 400):
            assert slam.handle_new_frame(imgs[k]) in (1, 2)
        slam.finish()
        assert len(slam._accepted) == 400 and len(slam.ba_reports) >= 10
        assert all(r["engine"] == "device" for r in slam.ba_reports) and not slam.ba_fallbacks and slam.ba_engine == "device"
        assert max(r["poses"] for r in slam.ba_reports) <= 256 and slam.ba_reports[-1]["accepted_frames"] == 400
        if kw:
            assert max(r["poses"] for r in slam.ba_reports) > 200                    # (everything while it fits, then every keyframe + the latest frames)
        c = np.array([-P[:, :3].T @ P[:, 3] for P in slam.poses])
        assert float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))) < 0.0057       # measured 0.0027
        slam.close()


@pytest.mark.gpu
def test_observation_log_of_the_device_loop(gpu):
    """The log the frame kernels keep for the adjuster (csrc/slam_frame.hip): one entry per kept track and accepted frame, pose
    indices = ranks among the accepted frames, a new landmark's entries reach back to its base keyframe (slam2.py:634-641), and
    every logged pixel of a landmark reprojects near it through the loop's own pose and map."""
    import torch
    seq = gpu.synthetic.PlaneSequence(frames=60)                                       # (the first half of the path)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    objp, imgp = objp[vis], imgp[vis]
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(30)]
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    slam._bundle_adjust = lambda: None                                                 # log only: the estimate stays the plain loop's
    slam.start(imgs[0], objp, imgp)
    rets = [2] + [slam.handle_new_frame(imgs[k]) for k in range(1, 30)]
    slam.finish()
    lm, ps, uv = slam.read_log()
    n_lm = len(slam.objp)
    assert all(r in (1, 2) for r in rets)
    assert ps.min() == 0 and ps.max() == 29 and lm.max() == n_lm - 1 and lm.min() >= -1
    assert np.array_equal(lm[:len(objp)], np.arange(len(objp))) and np.all(ps[:len(objp)] == 0)     # the first frame's associations
    per_pose = np.bincount(ps, minlength=30)
    assert per_pose[1:].min() >= 100                                                   # every frame logs all the tracks it keeps
    known = lm >= 0
    # a landmark made at a keyframe has an entry at that keyframe's base keyframe and at every frame in between
    kfs = [i for i, r in enumerate(rets) if r == 2]
    new = np.arange(len(objp), n_lm)
    first = np.full(n_lm, 10 ** 9); last = np.full(n_lm, -1); count = np.zeros(n_lm, int)
    np.minimum.at(first, lm[known], ps[known]); np.maximum.at(last, lm[known], ps[known]); np.add.at(count, lm[known], 1)
    assert np.all(np.isin(first[new], kfs)) and np.all(count[new] >= 2)
    assert np.all(count[new] == last[new] - first[new] + 1)                            # contiguous: base keyframe .. last sighting
    # reprojection of the logged pixels through the loop's own estimate (plain loop: a pixel or two)
    res = gpu.slam_device._reprojection_residuals(
        np.stack([gpu.bundle_adjustment.pose_from_world_to_camera(P) for P in slam.poses]), slam.objp.astype(np.float64),
        np.array([seq.K[0, 0], seq.K[1, 1], seq.K[0, 1], seq.K[0, 2], seq.K[1, 2], *np.asarray(seq.dist).reshape(-1)[:4]]),
        lm[known], ps[known], uv[known])
    # (the plain loop carries a few mistracked corners -- what the adjuster's screen is for: the tail is theirs)
    assert np.median(res) < 1.0 and np.percentile(res, 90) < 4.0 and np.percentile(res, 99) < 40.0
    slam.close()


@pytest.mark.gpu
def test_device_loop_with_the_reference_keyframe_sample(gpu):
    """max_homography_points="reference": the keyframe test on a random quarter of the tracks (slam2.py:48, 1088-1089), drawn on the
    device.  The loop still accepts every frame and stays on the path; which keyframes it takes depends on the draw (the study
    in profiles/r03/12_keyframe_sample_study.json: RMSE 0.3 % or 1.1 % of the path depending on the seed), so the bar is wider
    than the default's."""
    import torch
    seq = gpu.synthetic.PlaneSequence(frames=40)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    objp, imgp = objp[vis], imgp[vis]
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(40)]
    gt = seq.centres()
    path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())

    def run(seed, mh):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=seed, max_homography_points=mh)
        slam.start(imgs[0], objp, imgp)
        rets = [2] + [slam.handle_new_frame(imgs[k]) for k in range(1, 40)]
        traj = slam.trajectory()
        ratios = list(slam.homography_ratios) if hasattr(slam, "homography_ratios") else None
        slam.close()
        return rets, traj, ratios

    ref_runs = [run(seed, "reference") for seed in (0, 1, 2)]
    for rets, traj, _ in ref_runs:
        assert all(r in (1, 2) for r in rets) and sum(r == 2 for r in rets) >= 8
        assert np.sqrt(np.mean(np.linalg.norm(traj - gt, axis=1) ** 2)) < 0.02 * path
    assert any(not np.array_equal(ref_runs[0][1], r[1]) for r in ref_runs[1:])   # the run depends on the draw
    # all tracks (the default): the seed only moves the RANSAC draws, the trajectories stay within 0.2 % of the path of each other
    # (this 40-frame rendering is at the edge of what the plain loop tracks: 1.0-1.1 % of the path, 0.33 % for the 60-frame one)
    rm = [np.sqrt(np.mean(np.linalg.norm(run(seed, 0)[1] - gt, axis=1) ** 2)) for seed in (0, 2)]
    assert abs(rm[0] - rm[1]) < 2e-3 * path and max(rm) < 0.02 * path


@pytest.mark.gpu
def test_device_loop_rejects_a_broken_frame_and_recovers(gpu):
    """A frame that shares nothing with its predecessor (noise) is rejected by the lost-tracks gate without touching the state;
    the following frame is tracked from the last good image."""
    import torch
    seq = gpu.synthetic.PlaneSequence(frames=60)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(8)]
    noise = torch.from_numpy(np.random.default_rng(0).integers(0, 255, (seq.H, seq.W), dtype=np.uint8)).cuda()
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=3)
    slam.start(imgs[0], objp[vis], imgp[vis])
    rets = [slam.handle_new_frame(imgs[k]) for k in range(1, 5)]
    n_before = len(slam.tracks()[0])
    assert all(r in (1, 2) for r in rets)
    assert slam.handle_new_frame(noise) == 0 and slam.poses[-1] is None and int(slam.reports[-1][1]) in (1, 2, 3, 4, 5)
    assert len(slam.tracks()[0]) == n_before                    # untouched
    assert slam.handle_new_frame(imgs[5]) in (1, 2)              # tracked from frame 4's image
    slam.finish()
    with pytest.raises(ValueError):
        slam.handle_new_frame(imgs[6].cpu())
    slam.close()


@pytest.mark.gpu
def test_device_loop_to_bundle_adjustment_files_end_to_end(gpu, tmp_path):
    """The device-resident loop with the recorder on: the BA_info.* / traj_out / map_out set it writes passes the reader's two
    validators and the CLI-compatible bundle adjuster improves on it -- BASELINE configs[4] end to end, the loop's state on the
    GPU."""
    import subprocess
    import run_slam_loop
    io = gpu.ba_io
    info = io.BundleAdjustmentInfoContainer(str(tmp_path), "dloop", 1)
    out = run_slam_loop.run_device(40, ba_info=info, out_files=(str(tmp_path), "dloop", 30))
    assert out["accepted"] == 40
    fn = io.create_filenames(str(tmp_path), "dloop", 1)
    data = io.load_data(fn, 30)
    io.validate_data_integrity(data, 1)
    ok, _ = io.validate_sufficiently_constrained(data, True)
    assert ok
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bundle_adjust.py"), str(tmp_path), "dloop", "1", "30", "1", "1", "0", "1", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    gt = gpu.synthetic.PlaneSequence(frames=40).centres()
    before = np.array([p[9:] for _, p in io.load_trajectory(fn.trajectories_in[0])])
    after = np.array([p[9:] for _, p in io.load_trajectory(fn.trajectories_out[0])])
    e0 = np.sqrt(np.mean(np.sum((before - gt) ** 2, axis=1)))
    e1 = np.sqrt(np.mean(np.sum((after - gt) ** 2, axis=1)))
    path = np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()
    assert e1 < 0.01 * path and e1 < 1.5 * e0 + 1e-3
    line = [l for l in r.stdout.splitlines() if l.startswith("cost")][0]
    assert float(line.split()[3]) < float(line.split()[1])


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["pinned", "pageable"])
def test_frame_ingest_on_a_side_stream_gives_the_same_run(gpu, source):
    """`slam_device.FrameUploader` (csrc/slam_ingest.hip): the frames go from host memory to the device INSIDE the loop (the reference
    reads every frame inside its loop, slam2.py:1209-1213), on a stream of their own while the loop's kernels work on the frames
    before -- from one pinned capture buffer, or from ordinary arrays through pinned staging slots; the loop's stream waits for a
    frame's upload on the device.  Same decisions and the same poses, bit for bit, as the run whose frames were on the device
    beforehand -- with a frame the tracker loses in between (a rejected frame gives its ring slot back at once, its predecessor stays
    the previous image)."""
    import torch
    seq, objp, imgp, imgs = _rendered(gpu, 30)
    imgs = list(imgs)
    imgs[17] = torch.zeros_like(imgs[17])
    host = [im.cpu().numpy() for im in imgs]

    def run(make):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=2, bundle_adjust="keyframe")
        rets = []
        for k, img in enumerate(make(slam)):
            if k == 0:
                slam.start(img, objp, imgp)
                rets.append(2)
            else:
                rets.append(slam.handle_new_frame(img))
        slam.finish()
        poses = [None if P is None else np.array(P) for P in slam.poses]
        free = None if slam._ingest_free is None else sorted(slam._ingest_free)
        slam.close()
        return rets, poses, free
    base = run(lambda slam: imgs)
    src = torch.from_numpy(np.stack(host)).pin_memory() if source == "pinned" else host
    other = run(lambda slam: gpu.slam_device.FrameUploader(slam, src, ahead=2))
    assert other[0] == base[0] and 2 in base[0][1:] and 0 in base[0]
    for a, b in zip(base[1], other[1]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    assert len(other[2]) == 4                                                        # of the 5 ring slots only the last image's is still held
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=2)
    with pytest.raises(ValueError):
        gpu.slam_device.FrameUploader(slam, torch.zeros((2, seq.H, seq.W), dtype=torch.uint8))      # a torch source has to be pinned
    slam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [False, True], ids=["prepared", "enqueued"])
@pytest.mark.parametrize("frames_from", ["device tensors", "ingest ring"])
def test_next_pairs_pyramid_ahead_of_its_frame_gives_the_same_run(gpu, frames_from, pipeline):
    """`mqs_slam_prepare_next`: the tracker's pyramid of the NEXT image pair is built on a side stream under the current frame's pose
    kernels; the next `mqs_slam_track` finds it by the two image pointers and runs the tracker alone.  Same decisions, poses and tracks,
    bit for bit, as with the pyramid built inside every call -- including a frame the tracker loses (its pair was prepared and never
    comes: the following frame is tracked from the lost frame's predecessor, pyramid built in the call) and the keyframes' top-ups.
    The same holds with the pair also TRACKED ahead (behind the current frame's hypotheses; the default wherever the pyramid is ahead) and,
    `pipeline`, with the next frame's pose kernels ENQUEUED behind the current frame's decision (`mqs_slam_pipeline`: they run behind an
    ordinary frame and do nothing behind a keyframe or a lost frame -- all three occur here)."""
    import torch
    seq, objp, imgp, imgs = _rendered(gpu, 34)
    imgs = list(imgs)
    imgs[17] = torch.zeros_like(imgs[17])
    host = [im.cpu().numpy() for im in imgs]

    def run(ahead):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=2, bundle_adjust="keyframe", reassociate=True)      # (BA and the matcher's re-association behind every keyframe)
        slam.prepare_next = ahead
        slam.pipeline = pipeline
        rets = []
        if frames_from == "device tensors":
            slam.start(imgs[0], objp, imgp)
            rets = [2] + [slam.handle_new_frame(imgs[k], imgs[k + 1] if k + 1 < len(imgs) else None) for k in range(1, len(imgs))]
        else:
            for k, img in enumerate(gpu.slam_device.FrameUploader(slam, host, ahead=2)):
                rets.append((slam.start(img, objp, imgp) is not None and 2) if k == 0 else slam.handle_new_frame(img))
        slam.finish()
        out = (rets, [None if P is None else np.array(P) for P in slam.poses], [np.array(a) for a in slam.tracks()], slam.objp.copy())
        slam.close()
        return out
    plain, ahead = run(False), run(True)
    assert plain[0] == ahead[0] and 0 in plain[0] and plain[0].count(2) >= 5
    for a, b in zip(plain[1], ahead[1]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    for a, b in zip(plain[2], ahead[2]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(plain[3], ahead[3])

@pytest.mark.gpu
def test_a_frame_enqueued_ahead_has_to_be_the_next_one_tracked(gpu):
    """`mqs_slam_pipeline`'s contract: behind an ordinary frame the frame named by `mqs_slam_set_next` is already running -- tracking
    another one is MQS_E_ARG (not a silently different run), and so is switching the pipeline off in between."""
    seq, objp, imgp, imgs = _rendered(gpu, 8)
    slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=2)
    slam.start(imgs[0], objp, imgp)
    k = 1
    while slam.handle_new_frame(imgs[k], imgs[k + 1]) != 1 or k == 1:       # (the first call creates the side stream: nothing runs ahead of it)
        k += 1
    L = gpu._lib.lib()
    assert L.mqs_slam_pipeline(slam._h, 0) == -1 and b"enqueued" in L.mqs_last_error()
    with pytest.raises(RuntimeError, match="enqueued"):
        slam.handle_new_frame(imgs[k + 2])
    slam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1], ids=["four wavefronts, a barrier per pivot", "one wavefront, panels of four pivots on the matrix pipe"])
def test_resident_adjusters_diagonal_tile_factor(gpu, form):
    """The 32 x 32 diagonal-tile factor of the resident adjuster's Cholesky (`chol_block.h`; `mqs_debug_factor32`): L in the lower
    triangle, inv(L)'s strictly lower part transposed above it -- both forms against numpy on well- and badly-conditioned matrices, the
    scales a reduced camera system has (1e-3 .. 1e8 on the diagonal), and the flag on a matrix that is not positive definite."""
    import ctypes
    L_ = gpu._lib.lib()
    rng = np.random.default_rng(7)
    def run(A):
        out, bad = np.zeros((32, 32)), ctypes.c_int32(0)
        gpu._lib.check(L_.mqs_debug_factor32(np.ascontiguousarray(A).ctypes.data_as(gpu._lib.c_f64p), out.ctypes.data_as(gpu._lib.c_f64p), form, ctypes.byref(bad)))
        return out, bad.value
    for trial in range(6):
        M = rng.standard_normal((32, 40))
        scale = np.diag(10.0 ** rng.uniform(-1.5, 4.0, 32)) if trial % 2 else np.eye(32)
        A = scale @ (M @ M.T + (1e-6 if trial == 5 else 1.0) * np.eye(32)) @ scale
        out, bad = run(A)
        assert bad == 0
        Lr = np.linalg.cholesky(A)
        Li = np.linalg.inv(Lr)
        tol = 1e-12 if trial < 5 else 1e-7                    # (trial 5: condition number ~1e8 and more)
        np.testing.assert_allclose(np.tril(out), Lr, rtol=0, atol=tol * np.abs(Lr).max())
        up = np.triu(out, 1)                                  # out[c][j] = inv(L)[j][c], j > c
        want = np.triu(Li.T, 1)
        np.testing.assert_allclose(up, want, rtol=0, atol=tol * max(1.0, np.abs(Li).max()) * (1e3 if trial == 5 else 1.0))
        # what the adjuster does with it: X = B inv(L)^T solves X L^T = B
        B = rng.standard_normal((32, 32))
        Linv = np.tril(up.T, -1) + np.diag(1.0 / np.diag(out))
        np.testing.assert_allclose((B @ Linv.T) @ np.tril(out).T, B, rtol=0, atol=(1e-10 if trial < 5 else 1e-4) * np.abs(B).max())
    A = rng.standard_normal((32, 32)); A = A + A.T                # symmetric, indefinite
    assert run(A)[1] == 1


@pytest.mark.gpu
def test_two_cameras_on_one_device_each_run_as_if_alone(gpu):
    """The reference is a MULTIPLE-quadrotor system: several loops on one device, one handle and one host thread per camera (the library
    calls release the interpreter lock while they wait).  Two cameras with their own seeds, BA per keyframe and re-association, running
    at the same time: each reports the run it reports alone -- decisions, poses, tracks, map, bit for bit (the handles share nothing
    but the device; the adjuster's launches of one device are serialised by the library)."""
    import threading
    seq, objp, imgp, imgs = _rendered(gpu, 40)

    def run(seed, out, start=None):
        slam = gpu.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=seed, bundle_adjust="keyframe", reassociate=True)
        if start is not None:
            start.wait()
        slam.start(imgs[0], objp, imgp)
        rets = [2] + [slam.handle_new_frame(imgs[k], imgs[k + 1] if k + 1 < len(imgs) else None) for k in range(1, len(imgs))]
        slam.finish()
        out[seed] = (rets, [None if P is None else np.array(P) for P in slam.poses], [np.array(a) for a in slam.tracks()], slam.objp.copy())
        slam.close()
    alone, both = {}, {}
    for seed in (3, 4):
        run(seed, alone)
    for rep in range(3):
        start = threading.Event()
        th = [threading.Thread(target=run, args=(seed, both, start)) for seed in (3, 4)]
        for t in th:
            t.start()
        start.set()
        for t in th:
            t.join()
        for seed in (3, 4):
            a, b = alone[seed], both[seed]
            assert a[0] == b[0] and a[0].count(2) >= 5
            for x, y in zip(a[1], b[1]):
                assert (x is None) == (y is None) and (x is None or np.array_equal(x, y))
            for x, y in zip(a[2], b[2]):
                np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(a[3], b[3])
