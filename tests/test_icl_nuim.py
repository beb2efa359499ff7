"""The reference's own SLAM example run on real images: slam2.py's ExampleUsage for the ICL-NUIM living-room sequence
(slam2.py:924-933).  The reference commits the images (a 200-frame subset), the initialisation (init_pose.txt, init_points.pcd),
the renderer's exact trajectory (traj_groundtruth3.txt) and -- the golden vector -- the trajectory slam2.py itself wrote for these
images with OpenCV 2.4's goodFeaturesToTrack / calcOpticalFlowPyrLK / solvePnPRansac inside (traj_out.cam0-slam2.txt).
tests/golden/make_icl_nuim.py extracted the first 80 frames (grey, OpenCV's BGR2GRAY formula) and those files' rows.

What is pinned here:
  * oracle/features_np.py's pyramidal Lucas-Kanade on REAL images against the renderer's exact geometry (CPU);
  * the product's corner detector and tracker against that oracle on the same real images (GPU);
  * the product's whole loop -- detect, track, solvePnPRansac, keyframe test, triangulation -- against the trajectory the
    reference's loop produced on the same frames (GPU): not bit for bit (RANSAC draws, OpenCV's fixed-point LK), but frame by frame
    within the distance the reference's own output keeps from the exact trajectory.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
FIX = os.path.join(HERE, "golden", "icl_nuim_traj3n", "sequence.npz")


@pytest.fixture(scope="module")
def seq():
    d = np.load(FIX)
    return {k: d[k] for k in d.files}


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def project_through_tum_row(row, pts, K):
    """TUM row = the camera's pose in the world (slam2.py:698-741 writes the inverse of P): pixels of `pts` and their depths."""
    R = quat_to_R(row[4:8]).T
    X = (pts - row[1:4]) @ R.T @ K.T
    return X[:, :2] / X[:, 2:3], X[:, 2]


def test_fixture_is_the_reference_s_example_run(seq):
    assert seq["frames"].shape[1:] == (480, 640) and seq["frames"].dtype == np.uint8 and len(seq["frames"]) >= 80
    assert seq["K"][1, 1] < 0                                                   # the data set's negative fy (slam2.py:1047)
    assert len(seq["init_points"]) == 23
    # the first row of the reference's output and of the exact trajectory both ARE the initial pose
    P0 = seq["init_pose"]
    c0 = -P0[:3, :3].T @ P0[:3, 3]
    assert np.abs(seq["traj_slam2"][0, 1:4] - c0).max() < 1e-5 and np.abs(seq["traj_groundtruth"][0, 1:4] - c0).max() < 1e-5
    uv_row, z = project_through_tum_row(seq["traj_groundtruth"][0], seq["init_points"], seq["K"])
    X = np.c_[seq["init_points"], np.ones(23)] @ P0[:3].T @ seq["K"].T
    assert np.abs(uv_row - X[:, :2] / X[:, 2:3]).max() < 1e-3 and (z > 0).all()
    # the reference's own run stays within millimetres of the exact trajectory over these frames (results_ate-slam2.txt: 0.134 m
    # rmse over the whole sequence; the drift starts behind frame 85)
    e = np.linalg.norm(seq["traj_slam2"][:80, 1:4] - seq["traj_groundtruth"][:80, 1:4], axis=1)
    assert np.sqrt(np.mean(e ** 2)) < 0.005 and e.max() < 0.015


def test_oracle_lucas_kanade_follows_real_corners_to_the_renderer_s_geometry(seq):
    """oracle/features_np.py (the restatement of OpenCV 2.4's calcOpticalFlowPyrLK that the GPU tracker is held to) on the
    reference's real, noisy images: the 23 initial points -- known 3-D corners of the room -- tracked frame to frame land where
    the exact trajectory projects them."""
    from oracle import features_np as Fn
    K, pts = seq["K"], seq["init_points"]
    cur, _ = project_through_tum_row(seq["traj_groundtruth"][0], pts, K)
    cur = cur.astype(np.float32)
    alive = np.ones(len(pts), bool)
    for k in range(1, 6):
        nxt, st, err = Fn.calc_optical_flow_pyr_lk(seq["frames"][k - 1], seq["frames"][k], cur)
        want, _ = project_through_tum_row(seq["traj_groundtruth"][k], pts, K)
        alive &= st.astype(bool).ravel()
        e = np.linalg.norm(nxt - want, axis=1)
        assert alive.sum() >= 20
        assert np.median(e[alive]) < 0.1 and np.percentile(e[alive], 80) < 0.3       # measured: median 0.03-0.05 px, p90 0.09-0.16
        cur = nxt.astype(np.float32)


def start_mask(seq):
    """slam2.py:29-40 keypoint_mask over the projected initial points: discs of keypoint_coverage_radius = 12 px cleared."""
    uv, _ = project_through_tum_row(seq["traj_groundtruth"][0], seq["init_points"], seq["K"])
    H, W = seq["frames"].shape[1:]
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.ones((H, W), np.uint8)
    for p in uv:
        mask[(xx - int(p[0])) ** 2 + (yy - int(p[1])) ** 2 <= 12 * 12] = 0          # cv2.circle through the Python 2 binding: truncated centre
    return mask


def test_oracle_detector_fills_the_reference_s_quota_on_the_first_frame(seq):
    """slam2.py:1172-1174 asks goodFeaturesToTrack for 300 - 23 = 277 corners outside the discs around the 23 initial points, and
    the reference's record of a run on this sequence (BA_info.measurements.point2D3DAssocs: 296 tracked points in frame 0, the 23
    initial ones among them) shows it got at least 273.  The quality threshold is relative to the largest response UNDER THE MASK (OpenCV 2.4
    featureselect.cpp: minMaxLoc(eig, 0, &maxVal, 0, 0, mask)); relative to the whole image's maximum -- the strongest corners are
    the masked ones -- this frame has 216 candidates at the 12 px spacing."""
    from oracle import features_np as Fn
    I, mask = seq["frames"][0], start_mask(seq)
    eig = Fn.corner_min_eigenval(I)
    assert eig[mask != 0].max() < 0.8 * eig.max()                                # the strongest corners are under the discs
    got = Fn.good_features_to_track(I, 277, 0.01, 12.0, mask)
    assert len(got) == 277 and np.all(mask[got[:, 1].astype(int), got[:, 0].astype(int)] != 0)
    # what the whole image's maximum would give (the restatement before round 4)
    assert len(Fn.good_features_to_track(I, 277, 0.01 * float(eig.max() / eig[mask != 0].max()), 12.0, mask)) < 230


def fundamental_from_exact_poses(seq, a, b):
    """x_b^T F x_a = 0 for the exact poses of frames a and b (pixels)."""
    K = seq["K"]
    Ra, Rb = quat_to_R(seq["traj_groundtruth"][a, 4:8]).T, quat_to_R(seq["traj_groundtruth"][b, 4:8]).T
    ca, cb = seq["traj_groundtruth"][a, 1:4], seq["traj_groundtruth"][b, 1:4]
    t = Rb @ (ca - cb)
    E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ (Rb @ Ra.T)
    return np.linalg.inv(K).T @ E @ np.linalg.inv(K)


def sampson_distance(Fm, pa, pb):
    x1, x2 = np.c_[pa, np.ones(len(pa))], np.c_[pb, np.ones(len(pb))]
    l2, l1 = x1 @ Fm.T, x2 @ Fm
    return np.abs(np.sum(x2 * l2, axis=1)) / np.sqrt(l2[:, 0] ** 2 + l2[:, 1] ** 2 + l1[:, 0] ** 2 + l1[:, 1] ** 2)


def test_oracle_detector_and_tracker_obey_the_exact_epipolar_geometry(seq):
    """The oracle's corners of frame 30 (goodFeaturesToTrack as slam2.py calls it) tracked through four real frames by the oracle's
    Lucas-Kanade (slam2.py:381-382: status and err < max_OF_error kept): start and end of every surviving track against the
    epipolar geometry of the two frames' EXACT poses (13 mm of baseline) -- measured median 0.04 px, 90 % within 0.33 px; the
    few per cent beyond a pixel are corners sliding along edges, which the loop's RANSAC is there for."""
    from oracle import features_np as Fn
    a, b = 30, 34
    p0 = Fn.good_features_to_track(seq["frames"][a], 300, 0.01, 12.0)
    cur, alive = p0.copy(), np.ones(len(p0), bool)
    for k in range(a, b):
        nxt, st, err = Fn.calc_optical_flow_pyr_lk(seq["frames"][k], seq["frames"][k + 1], cur)
        alive &= st.astype(bool).ravel() & (err.ravel() < 12.0)
        cur = nxt.astype(np.float32)
    assert len(p0) >= 150 and alive.mean() > 0.85
    s = sampson_distance(fundamental_from_exact_poses(seq, a, b), p0[alive].astype(np.float64), cur[alive].astype(np.float64))
    assert np.median(s) < 0.1 and np.percentile(s, 90) < 0.6


@pytest.mark.gpu
def test_detector_and_tracker_equal_the_oracle_on_the_real_images(seq, gpu):
    from oracle import features_np as Fn
    I, J = seq["frames"][0], seq["frames"][1]
    mask = start_mask(seq)
    ref_m = Fn.good_features_to_track(I, 277, 0.01, 12.0, mask)
    np.testing.assert_array_equal(gpu.features.goodFeaturesToTrack(I, 277, 0.01, 12.0, None, mask), ref_m)     # under the reference's mask
    ref = Fn.good_features_to_track(I, 300, 0.01, 12.0)
    got = gpu.features.goodFeaturesToTrack(I, 300, 0.01, 12.0)
    assert len(ref) >= 150
    np.testing.assert_array_equal(got, ref)                                     # the same corners in the same order
    rn, rs, re = Fn.calc_optical_flow_pyr_lk(I, J, ref)
    gn, gs, ge = gpu.features.calcOpticalFlowPyrLK(I, J, ref)
    np.testing.assert_array_equal(gs.ravel(), rs.ravel())
    ok = rs.ravel().astype(bool)
    assert ok.mean() > 0.9
    assert np.abs(gn[ok] - rn[ok]).max() < 5e-3                                 # float32 window sums in a different order


@pytest.mark.gpu
def test_loop_reproduces_the_reference_s_trajectory_on_its_example_sequence(gpu):
    """BASELINE configs[4] on data the reference holds: the device-resident loop over the 80 frames against the trajectory
    slam2.py wrote for them (and against the exact one), over four RANSAC seeds.  Measured on the round's last tree over 32 seeds
    (profiles/r04/17): median 4.1 mm rmse from the reference's trajectory (quartiles 3.8 / 5.6, worst 22 mm), whose own distance
    from the exact one is 4.4 mm and 0.088 degrees over a path of 0.32 m; ours a median 3.8 mm and 0.076 degrees.  The plain loop --
    the reference's as much as this one -- has nothing that pulls a bad keyframe back: at earlier trees of the round a fifth of the
    seeds ended at 43-46 mm.  Every run accepts every frame."""
    import run_icl_nuim
    outs = [run_icl_nuim.run(80, seed=seed) for seed in range(4)]
    for out in outs:
        assert out["accepted"] == 80 and 2 <= out["keyframes"] <= 10
        assert out["reference_vs_groundtruth_rmse_m"] < 0.005
        # within twice the worst of the 32-seed study (22 mm from the reference's trajectory, 23 mm from the exact one; round 5, 8 seeds:
        # 2.8-7.0 mm from the exact one, profiles/r05): a regression to 30 mm on any seed fails
        assert out["ours_vs_reference_rmse_m"] < 0.03 and out["ours_vs_groundtruth_rmse_m"] < 0.03
        assert out["frames_per_s"] > 500
    d_ref = sorted(o["ours_vs_reference_rmse_m"] for o in outs)
    d_gt = sorted(o["ours_vs_groundtruth_rmse_m"] for o in outs)
    assert d_ref[1] < 0.008 and d_gt[1] < 0.008                                  # at least two of the four agree with the reference to millimetres
    assert 0.5 * (d_ref[1] + d_ref[2]) < 0.008 and 0.5 * (d_gt[1] + d_gt[2]) < 0.008          # the median of the four (measured 4.1 / 3.5 mm)
    assert min(o["ours_vs_reference_max_m"] for o in outs) < 0.02
    assert sorted(o["orientation_rmse_deg"]["ours_vs_reference"] for o in outs)[1] < 0.2          # measured 0.09, 0.09, 0.18, 0.25 degrees


@pytest.mark.gpu
def test_host_driven_loop_on_the_example_sequence(gpu):
    """slam_loop.MonoSlam (the loop with its state on the host, one library call per OpenCV call of slam2.py) over the same frames:
    measured 6.2 mm from the exact trajectory, 5.6 mm from the reference's."""
    import run_icl_nuim
    out = run_icl_nuim.run(80, seed=0, device=False)
    assert out["accepted"] == 80 and out["keyframes"] >= 2
    assert out["ours_vs_groundtruth_rmse_m"] < 0.05 and out["ours_vs_reference_rmse_m"] < 0.05


@pytest.mark.gpu
def test_example_run_leaves_the_reference_s_output_files(gpu, tmp_path):
    """--out: what slam2.py's write_output leaves behind (:698-741) -- the trajectory in TUM format and the map as PCD, readable by
    the reference-format readers, the trajectory's first row the initial pose like the reference's own file."""
    import run_icl_nuim
    out = run_icl_nuim.run(20, seed=0, out_dir=str(tmp_path))
    io = gpu.ba_io
    traj = io.load_trajectory(out["written"][0])
    pts = io.load_map(out["written"][1])
    d = np.load(run_icl_nuim.FIX)
    assert len(traj) == 20 and len(pts) == out["landmarks"]
    assert abs(traj[0][0] - 1.0 / 30) < 1e-9 and np.abs(np.asarray(traj[0][1])[9:] - d["traj_slam2"][0, 1:4]).max() < 1e-4
    assert np.abs(np.asarray(traj[10][1])[9:] - d["traj_slam2"][10, 1:4]).max() < 0.01


@pytest.mark.gpu
def test_loop_with_adjustment_per_keyframe_on_the_example_sequence(gpu):
    """... and with the bundle adjustment per keyframe EVERY seed stays within a centimetre of the exact trajectory (measured 4-10 mm
    rmse over 80 frames, 4-6 mm over 200 where the reference's committed run has drifted to 171 mm)."""
    import run_icl_nuim
    for seed in range(4):
        out = run_icl_nuim.run(80, bundle_adjust="keyframe", seed=seed)
        assert out["accepted"] == 80 and out["engine"] == "device"
        assert out["ours_vs_groundtruth_rmse_m"] < 0.015 and out["ours_vs_groundtruth_max_m"] < 0.035
        assert out["ours_vs_reference_rmse_m"] < 0.018
    # round 4's host-built adjustment on the same frames: the resident adjuster IS that adjustment (same trajectory to the digits printed)
    dev, host = run_icl_nuim.run(80, bundle_adjust="keyframe", seed=0), run_icl_nuim.run(80, bundle_adjust="keyframe", seed=0, engine="host")
    assert host["engine"] == "host" and abs(dev["ours_vs_groundtruth_rmse_m"] - host["ours_vs_groundtruth_rmse_m"]) < 2e-4
    assert dev["keyframe_frames"] == host["keyframe_frames"]


def test_reference_s_committed_run_drifts_over_the_200_frames():
    """The golden vector of the longer run: over the 200 frames the reference commits, the trajectory slam2.py wrote for them leaves
    the exact one by 0.171 m rmse (0.31 m at the worst frame) -- 4.3 mm over the first 80.  (results_ate-slam2.txt says 0.134 m for the
    run as a whole: that figure is after evaluate_ate.py's rigid alignment, this one before it.)"""
    d = np.load(FIX)
    a, g = d["traj_slam2_all"][:200], d["traj_groundtruth_all"][:200]
    e = np.linalg.norm(a[:, 1:4] - g[:, 1:4], axis=1)
    assert abs(np.sqrt(np.mean(e ** 2)) - 0.17097) < 2e-4 and abs(e.max() - 0.31047) < 2e-4
    assert np.sqrt(np.mean(e[:80] ** 2)) < 0.0044 and np.argmax(e > 0.02) > 80          # the drift starts behind frame 85
    rest = os.path.join(os.path.dirname(FIX), "sequence_rest.npz")
    r = np.load(rest)
    assert int(r["first"]) == 80 and r["frames"].shape == (120, 480, 640) and r["frames"].dtype == np.uint8


@pytest.mark.gpu
def test_all_200_frames_of_the_example_sequence(gpu):
    """The part of the reference's example run its 80-frame fixture does not see: frames 85-199, where the reference's committed
    trajectory drifts to 0.171 m.  The plain loop -- slam2.py's flow -- drifts as well (52-205 mm over 8 seeds, a median of 81; 52 mm
    at seed 0); with the bundle adjustment per keyframe every seed stays within a centimetre of the exact trajectory: the adjustment
    over EVERY accepted frame (round 5: ~20 adjustments of up to 200 poses) 4.2-8.0 mm over 16 seeds, the default selection (round 6:
    every keyframe + the frames since the third keyframe from the end, 20-35 poses) 3.4-7.2 mm at twice the frame rate
    (profiles/r06).  Bars within twice the measurements."""
    import run_icl_nuim
    plain = run_icl_nuim.run(200, seed=0)
    assert plain["accepted"] == 200 and plain["keyframes"] >= 10
    assert abs(plain["reference_vs_groundtruth_rmse_m"] - 0.17097) < 2e-4
    assert 0.026 < plain["ours_vs_groundtruth_rmse_m"] < 0.105                   # it drifts, like the reference's run (measured 52 mm at this seed)
    errs = []
    for seed in range(4):
        out = run_icl_nuim.run(200, bundle_adjust="keyframe", seed=seed)
        assert out["accepted"] == 200 and out["engines"] == ["device"] and not out["fallbacks"]
        assert 15 <= out["poses_in_the_last_adjustment"] <= 70 and max(out["poses_per_adjustment"]) <= 100      # (the first adjustment: no keyframes to select from yet)
        assert out["ours_vs_groundtruth_rmse_m"] < 0.010 and out["ours_vs_groundtruth_max_m"] < 0.03
        assert out["orientation_rmse_deg"]["ours_vs_groundtruth"] < 0.5
        errs.append(out["ours_vs_groundtruth_rmse_m"])
    assert max(errs) < 0.06 * plain["reference_vs_groundtruth_rmse_m"]           # a thirtieth of the reference's drift
    every = run_icl_nuim.run(200, bundle_adjust="keyframe", seed=0, window=None)  # the whole graph behind every keyframe, as the reference's tool adjusts a recording
    assert every["engines"] == ["device"] and every["poses_in_the_last_adjustment"] == 200
    assert every["ours_vs_groundtruth_rmse_m"] < 0.010 and every["ours_vs_groundtruth_max_m"] < 0.03


@pytest.mark.gpu
def test_optional_second_pass_screen_on_the_example_sequence(gpu):
    """Off by default, because it is not in slam2.py's flow: `second_pass_screen` (mqs_slam_set_second_pass_screen) keeps a freshly
    triangulated point whose reprojection error in the current frame exceeds the bound out of the keyframe's second solvePnP -- the
    use slam2.py:1092 announces for max_2nd_solvePnP_reproj_error (1 px) and never makes.  Seeds 9 and 12 take their first keyframe
    at frame 36, where ONE of 215 points carries 325 px of error into the plain least squares (profiles/r04/17): 43 mm without the
    screen, 5-6 mm with it (tree 5c528d2; since Lucas-Kanade reads the image border these seeds take their first keyframe later
    and the plain flow survives it); over 32 seeds 2.6-6.9 mm with the screen (plain: 2.6-23 mm; the reference's own run: 4.4 mm).  The host-driven
    loop takes the same option."""
    import run_icl_nuim
    for seed in (9, 12):
        plain = run_icl_nuim.run(80, seed=seed)
        screened = run_icl_nuim.run(80, seed=seed, screen=1.0)
        if plain["keyframe_frames"][1] == 36:                                                           # (these seeds' first keyframe today)
            assert plain["ours_vs_groundtruth_rmse_m"] > 0.03                                          # slam2.py's flow, faithfully
        assert screened["accepted"] == 80 and screened["ours_vs_groundtruth_rmse_m"] < 0.009
        assert screened["orientation_rmse_deg"]["ours_vs_groundtruth"] < 0.2
    host = run_icl_nuim.run(80, seed=9, device=False, screen=1.0)
    assert host["accepted"] == 80 and host["ours_vs_groundtruth_rmse_m"] < 0.012


@pytest.mark.gpu
def test_the_reference_s_two_program_workflow_on_its_example_sequence(gpu):
    """slam2.py records, bundle_adjust adjusts the recording (the reference's ReadMe workflow): the device loop with the BA_info
    recorder, the 12-file set, tools/bundle_adjust.py with the reference's command line, the adjusted trajectory back.  On these 80
    frames the post-hoc adjustment does not beat the loop (measured 4.9 -> 8.5 mm from the exact trajectory) -- nor does the
    reference's own: its committed slam2 / slam2-BA trajectories are 4.3 / 6.5 mm from it over the same frames (profiles/r04/17)."""
    import run_icl_nuim
    out = run_icl_nuim.run_posthoc(80, seed=0)
    assert out["tool_return_code"] == 0 and out["poses_in_the_adjusted_file"] == 80
    assert out["loop_vs_groundtruth_rmse_m"] < 0.01 and out["after_the_tool_vs_groundtruth_rmse_m"] < 0.02


@pytest.mark.gpu
def test_keyframe_test_ratio_of_the_device_loop_is_find_homography_s(seq, gpu, monkeypatch):
    """keyframe_test (slam2.py:43-59) inside the device loop on the real tracks: the ratio w0 / w2 the decision kernel reports is
    that of cv2.findHomography(method = 0) as restated on the host -- normalised DLT, then the Levenberg-Marquardt refinement of the
    transfer error (fundam.cpp: estimator.refine) -- over the frame's kept tracks; with parallax the DLT alone differs in the third
    digit, the digit the 1.04 threshold looks at (the switch MQS_SLAM_HOMOGRAPHY_REFINE=0 gives that)."""
    import torch
    import run_icl_nuim
    L = gpu.slam_loop
    K, dist = seq["K"], seq["dist"]
    H, W = seq["frames"].shape[1:]
    uv, vis = run_icl_nuim.start_points(K, (H, W), seq["init_pose"], seq["init_points"])
    imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in seq["frames"][:80]]

    def ratios(refine, jacobi=False):
        monkeypatch.setenv("MQS_SLAM_HOMOGRAPHY_REFINE", "1" if refine else "0")
        monkeypatch.setenv("MQS_SLAM_NULL_VECTOR_JACOBI", "1" if jacobi else "0")
        s = gpu.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=0)
        s.start(imgs[0], seq["init_points"][vis], uv[vis])
        out = []
        for k in range(1, 80):
            r = s.handle_new_frame(imgs[k])
            rep = s.reports[-1]
            s.finish()
            if r == 2:
                out.append((k, None, None, None))
                continue
            p, b, lm, tid = s.tracks()
            u1 = gpu.camera.undistort_points(b.astype(np.float64), K, dist)
            u2 = gpu.camera.undistort_points(p.astype(np.float64), K, dist)
            w0 = np.linalg.svd(L.homography_dlt(u1, u2), compute_uv=False)
            w1 = np.linalg.svd(L.find_homography(u1, u2), compute_uv=False)
            out.append((k, float(rep[10]), w0[0] / w0[2], w1[0] / w1[2]))
        s.close()
        return out
    with_refine = ratios(True)
    assert sum(1 for r in with_refine if r[1] is None) >= 1                       # the sequence has keyframes in its 80 frames
    got = np.array([r[1] for r in with_refine if r[1] is not None])
    dlt = np.array([r[2] for r in with_refine if r[1] is not None])
    full = np.array([r[3] for r in with_refine if r[1] is not None])
    assert np.abs(got - full).max() < 1e-7                                         # measured: 1e-10
    assert np.abs(dlt - full).max() > 5e-4                                         # the refinement matters on this sequence (measured 1.2e-3 over all tracks;
                                                                                   # 6e-3 and more on the reference's 75-track samples)
    without = ratios(False)
    got0 = np.array([r[1] for r in without if r[1] is not None])
    dlt0 = np.array([r[2] for r in without if r[1] is not None])
    assert np.abs(got0 - dlt0).max() < 1e-9
    # the DLT's null vector: inverse iteration (the default) against the Jacobi sweeps it falls back to when it does not settle
    sweeps = ratios(False, jacobi=True)
    assert [r[0] for r in sweeps if r[1] is not None] == [r[0] for r in without if r[1] is not None]
    assert np.abs(np.array([r[1] for r in sweeps if r[1] is not None]) - got0).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [(10, 11), (60, 62)])
def test_detect_track_match_step_of_slam_py_on_the_real_frames(pair, seq, gpu):
    """slam.py's frame step (main_loop :57-226: FAST corners of the new frame, LK flow of the old frame's corners, match_OF_based =
    radiusMatch of the flow points against the FAST points + ratio test + one match per FAST point) on two real frames: the GPU
    path returns the oracle's matches -- same (query, train) pairs, the match indices bit for bit -- and the matched pairs obey
    the epipolar geometry of the EXACT poses of the two frames (the matcher itself has no fixture in the reference: this is what
    real data can say about it)."""
    from oracle import features_np as Fn, matching_np as Mn
    a, b = pair
    I, J = seq["frames"][a], seq["frames"][b]
    F = gpu.slam_frontend
    left, _ = gpu.features.FastFeatureDetector().detect_arrays(I)
    assert len(left) > 300
    right_fast, matches, _, mean_flow, _, _ = F.main_loop(left, I, J, set())
    # the oracle's statements over the same frames
    np.testing.assert_array_equal(Fn.fast_detect(I)[0], left)
    np.testing.assert_array_equal(Fn.fast_detect(J)[0], right_fast)
    flow, st, err = Fn.calc_optical_flow_pyr_lk(I, J, left)
    ref = Mn.match_OF_based(flow, right_fast, err.reshape(-1), st.reshape(-1), 2.0, 0.7)
    assert len(ref) > 150
    got_pairs = sorted((m.queryIdx, t) for t, m in matches.items())
    ref_pairs = sorted((m.queryIdx, t) for t, m in ref.items())
    # the two LK implementations differ by < 5e-3 px: a flow point that sits within that of the 2 px radius or of the ratio bound
    # may fall on either side; everything else is the same pair
    common = set(got_pairs) & set(ref_pairs)
    assert len(common) >= 0.995 * max(len(got_pairs), len(ref_pairs))               # measured: every pair (406-490 per frame pair)
    # epipolar geometry from the exact trajectory: x2^T E x1 = 0 with E = [t]x R for the relative pose
    q = np.array([left[i] for i, _ in got_pairs], dtype=np.float64)
    r = np.array([right_fast[j] for _, j in got_pairs], dtype=np.float64)
    sampson = sampson_distance(fundamental_from_exact_poses(seq, a, b), q, r)
    assert np.median(sampson) < 0.3 and np.mean(sampson < 1.5) > 0.95        # FAST corners are integer pixels
    assert np.abs(mean_flow).max() < 20
