#!/usr/bin/env python3
"""The reference's own SLAM example run (slam2.py:924-933: the ICL-NUIM living-room sequence, 4th trajectory) through this
build's device-resident loop, against what the reference COMMITTED as that run's output (traj_out.cam0-slam2.txt, written by
slam2.py with OpenCV 2.4's goodFeaturesToTrack / calcOpticalFlowPyrLK / solvePnPRansac) and against the renderer's exact
trajectory.  Reads the fixture tests/golden/icl_nuim_traj3n/sequence.npz (tests/golden/make_icl_nuim.py).

    python tools/run_icl_nuim.py [frames] [--ba [--window K [--history H]] [--reference-noise] [--host-ba]] [--host] [--seed S] [--out DIR]
(--out: trajectory and map in the reference's formats; --host-ba: round 4's host-built adjustment; more than 80 frames: sequence_rest.npz too)
"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mqslam_amd

FIX = os.environ.get("MQS_ICL_FIXTURE") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                       "icl_nuim_traj3n", "sequence.npz")      # frames 0 .. 79


def centres_from_tum(rows):
    """TUM rows: timestamp tx ty tz qx qy qz qw with (t, q) the camera's pose IN the world (slam2.py:698-741 writes the inverse of P)."""
    return np.asarray(rows)[:, 1:4].copy()


def rotations_from_tum(rows):
    """World -> camera rotation of every TUM row (the file holds the camera's orientation IN the world: the transpose)."""
    q = np.asarray(rows)[:, 4:8]
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    return np.transpose(R, (0, 2, 1))


def angle_deg(Ra, Rb):
    c = (np.einsum("nij,nij->n", Ra, Rb) - 1.0) / 2.0
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


def start_points(K, shape, P_init, pts):
    """slam2.py:1054-1059: the predefined points projected through the initial pose, the visible ones kept (no rounding)."""
    X = np.c_[pts, np.ones(len(pts))] @ P_init[:3].T @ K.T
    uv = X[:, :2] / X[:, 2:3]
    vis = (X[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < shape[1]) & (uv[:, 1] >= 0) & (uv[:, 1] < shape[0])
    return uv, vis


FIX_REST = os.path.join(os.path.dirname(FIX), "sequence_rest.npz")      # frames 80 .. 199 (`make_icl_nuim.py rest`)


def load_sequence(frames=None):
    """The fixture as a dict; more than the first file's 80 frames: the second file's frames behind them (200 in all), and the
    trajectory rows from the files' complete tables."""
    f = np.load(FIX)
    d = {k: f[k] for k in f.files}
    if frames is not None and frames > len(d["frames"]):
        if not os.path.exists(FIX_REST):
            raise FileNotFoundError("%s (frames 80..199 of the reference's example sequence: tests/golden/make_icl_nuim.py rest)" % FIX_REST)
        r = np.load(FIX_REST)
        assert int(r["first"]) == len(d["frames"])
        d["frames"] = np.concatenate([d["frames"], r["frames"]])
        d["traj_slam2"], d["traj_groundtruth"] = d["traj_slam2_all"][:len(d["frames"])], d["traj_groundtruth_all"][:len(d["frames"])]
    return d


REFERENCE_NOISE = mqslam_amd.slam_device.REFERENCE_NOISE      # BA_info.noise.*-slam2.txt beside the reference's recording of this sequence


def run(frames=None, bundle_adjust=None, seed=0, device=True, reassociate=False, window="default", out_dir=None, screen=None, noise=None, engine="device",
        history=None, window_point_sigma="default", carry=True, check=False, upload=None, screen_iterations="default", prepare_next=True, pipeline=True):
    """upload: None -- every frame is on the device before the clock starts (the loop's kernels alone); "pinned" -- the frames lie in ONE pinned host
    buffer and go to the device inside the timed loop, on a side stream under the previous frames' kernels (`slam_device.FrameUploader`);
    "pageable" -- they lie in ordinary numpy arrays and pass through pinned staging slots on the uploader's thread.
    noise="reference": the in-loop adjuster's noise models take the values of the reference's own noise files for this sequence
    (instead of this build's defaults: a tighter prior on the first pose, a looser one on the start-up points)."""
    import torch
    d = load_sequence(frames)
    imgs_h = d["frames"] if frames is None else d["frames"][:frames]
    n = len(imgs_h)
    K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
    H, W = imgs_h.shape[1:]
    uv, vis = start_points(K, (H, W), P_init, pts)
    objp, imgp = pts[vis], uv[vis]
    t0 = time.perf_counter()
    if device:
        if upload is None:
            imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in imgs_h]
        elif upload == "pinned":
            src = torch.from_numpy(np.ascontiguousarray(imgs_h)).pin_memory()
        else:
            src = [np.ascontiguousarray(f) for f in imgs_h]
        torch.cuda.synchronize()
        slam = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, bundle_adjust=bundle_adjust, reassociate=reassociate,
                                                     max_homography_points="reference", second_pass_screen=screen,
                                                     ba_engine=engine, ba_history_keyframes=history, ba_check=check,
                                                     ba_noise="reference" if noise == "reference" else None,
                                                     **({} if window == "default" else {"ba_window_keyframes": window}),
                                                     **({} if screen_iterations == "default" else {"ba_screen_iterations": screen_iterations}))
        if window_point_sigma != "default":
            slam.ba_window_point_sigma = window_point_sigma
        slam.ba_carry = carry
        slam.prepare_next = prepare_next
        slam.pipeline = pipeline
        t0 = time.perf_counter()
        if upload is None:
            slam.start(imgs[0], objp, imgp)
            rets = [2]
            for k in range(1, n):
                rets.append(slam.handle_new_frame(imgs[k], imgs[k + 1] if k + 1 < n else None))
        else:
            up = mqslam_amd.slam_device.FrameUploader(slam, src)
            rets = []
            for k, img in enumerate(up):
                if k == 0:
                    slam.start(img, objp, imgp)
                    rets.append(2)
                else:
                    rets.append(slam.handle_new_frame(img))
        slam.finish()
    else:
        slam = mqslam_amd.slam_loop.MonoSlam(K, dist, (H, W), seed=seed, second_pass_screen=screen)
        slam.start(imgs_h[0], objp, imgp)
        rets = [2]
        for k in range(1, n):
            rets.append(slam.handle_new_frame(imgs_h[k]))
    dt = time.perf_counter() - t0
    c = np.array([(-P[:, :3].T @ P[:, 3]) if P is not None else [np.nan] * 3 for P in slam.projection_matrices()])
    ok = np.isfinite(c[:, 0])
    ref, gt = centres_from_tum(d["traj_slam2"][:n]), centres_from_tum(d["traj_groundtruth"][:n])
    err = lambda a, b: np.linalg.norm(a - b, axis=1)
    Rs = np.array([P[:, :3] if P is not None else np.eye(3) for P in slam.projection_matrices()])
    Rref, Rgt = rotations_from_tum(d["traj_slam2"][:n]), rotations_from_tum(d["traj_groundtruth"][:n])
    path = float(np.sum(np.linalg.norm(np.diff(gt, axis=0), axis=1)))
    out = {"frames": n, "frame_ingest": upload or "resident before the clock starts", "accepted": int(ok.sum()), "keyframes": int(sum(1 for r in rets if r == 2)), "landmarks": int(len(slam.objp)),
           "keyframe_frames": [k for k, r in enumerate(rets) if r == 2],
           "path_length_m": round(path, 4), "frames_per_s": round(n / dt, 1),
           "ours_vs_groundtruth_rmse_m": round(float(np.sqrt(np.mean(err(c[ok], gt[ok]) ** 2))), 5),
           "ours_vs_groundtruth_max_m": round(float(err(c[ok], gt[ok]).max()), 5),
           "reference_vs_groundtruth_rmse_m": round(float(np.sqrt(np.mean(err(ref, gt) ** 2))), 5),
           "reference_vs_groundtruth_max_m": round(float(err(ref, gt).max()), 5),
           "orientation_rmse_deg": {"ours_vs_groundtruth": round(float(np.sqrt(np.mean(angle_deg(Rs[ok], Rgt[ok]) ** 2))), 4),
                                    "reference_vs_groundtruth": round(float(np.sqrt(np.mean(angle_deg(Rref, Rgt) ** 2))), 4),
                                    "ours_vs_reference": round(float(np.sqrt(np.mean(angle_deg(Rs[ok], Rref[ok]) ** 2))), 4)},
           "ours_vs_reference_rmse_m": round(float(np.sqrt(np.mean(err(c[ok], ref[ok]) ** 2))), 5),
           "ours_vs_reference_max_m": round(float(err(c[ok], ref[ok]).max()), 5),
           "every_10th_frame_ours_ref_gt_error_mm": [[k, round(1e3 * float(err(c[k:k + 1], gt[k:k + 1])[0]), 2), round(1e3 * float(err(ref[k:k + 1], gt[k:k + 1])[0]), 2)]
                                                     for k in range(0, n, 10) if ok[k]]}
    if device and bundle_adjust:
        out["ms_per_adjustment_first_to_last"] = [round(r["build_ms"] + r["adjust_ms"] + r["write_back_ms"], 2) for r in slam.ba_reports][::max(1, len(slam.ba_reports) // 8)]
        out["poses_in_the_last_adjustment"] = slam.ba_reports[-1]["poses"] if slam.ba_reports else 0
        co = np.array([(-P[:, :3].T @ P[:, 3]) if P is not None else [np.nan] * 3 for P in slam.poses_online])
        out["online_vs_groundtruth_rmse_m"] = round(float(np.sqrt(np.mean(err(co[ok], gt[ok]) ** 2))), 5)
        out["landmarks_screened_out"] = int(slam.retired_landmarks().sum())
        out["engine"] = slam.ba_engine
        out["engines"] = sorted(set(r["engine"] for r in slam.ba_reports))
        out["poses_per_adjustment"] = [r["poses"] for r in slam.ba_reports]
        out["fallbacks"] = slam.ba_fallbacks
        if check:
            out["twin_max_pose_difference"] = float(max(np.abs(c["host_poses"] - c["device_poses"]).max() for c in slam.ba_checks))
    if out_dir:
        # what slam2.py's write_output leaves behind (:698-741): traj_out.cam0-<name>.txt in TUM format (30 fps, as the reference's
        # run) and map_out-<name>.pcd -- the inputs of the reference's own evaluation scripts
        os.makedirs(out_dir, exist_ok=True)
        fn = mqslam_amd.ba_io.create_filenames(out_dir, "mqslam", 1)
        mqslam_amd.ba_io.save_slam_output(fn, 30, slam.projection_matrices(), np.asarray(slam.objp, dtype=np.float64))
        out["written"] = [fn.trajectories_in[0], fn.map_in]
    if hasattr(slam, "close"):
        slam.close()
    return out


def run_posthoc(frames=80, seed=0, screen=None):
    """The reference's two-program workflow on its own data with this build's drop-ins: the loop records what slam2.py records for
    the bundle adjuster (BA_info.* files, trajectory, map: slam2.py:743-865, 698-741), `tools/bundle_adjust.py` -- the counterpart of
    the reference's C++ tool, same command line -- adjusts the recording, and both trajectories are held against the exact one."""
    import tempfile, torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bundle_adjust as ba_tool
    io = mqslam_amd.ba_io
    d = load_sequence(frames)
    n = frames
    K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
    H, W = d["frames"].shape[1:]
    uv, vis = start_points(K, (H, W), P_init, pts)
    gt = centres_from_tum(d["traj_groundtruth"][:n])
    with tempfile.TemporaryDirectory() as tmp:
        info = io.BundleAdjustmentInfoContainer(tmp, "mqslam", 1)
        slam = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, ba_info=info, max_homography_points="reference", second_pass_screen=screen)
        imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"][:n]]
        slam.start(imgs[0], pts[vis], uv[vis])
        for k in range(1, n):
            slam.handle_new_frame(imgs[k])
        slam.finish()
        info.write_all()
        info.write_noise(point2D=1.0)
        fn = io.create_filenames(tmp, "mqslam", 1)
        io.save_slam_output(fn, 30, slam.projection_matrices(), slam.objp)
        c0 = np.array([-P[:, :3].T @ P[:, 3] for P in slam.projection_matrices()])
        slam.close()
        rc = ba_tool.main(["bundle_adjust.py", tmp, "mqslam", "1", "30", "1", "0", "0", "1", "0"])       # odometry on, full optimisation
        traj = io.load_trajectory(os.path.join(tmp, "traj_out.cam0-mqslam-BA.txt"))
        c1 = np.array([np.asarray(p)[9:] for _, p in traj])
    e = lambda c: float(np.sqrt(np.mean(np.sum((c - gt[:len(c)]) ** 2, axis=1))))
    return {"frames": n, "tool_return_code": rc, "loop_vs_groundtruth_rmse_m": round(e(c0), 5), "after_the_tool_vs_groundtruth_rmse_m": round(e(c1), 5),
            "poses_in_the_adjusted_file": len(c1)}


if __name__ == "__main__":
    if "--posthoc" in sys.argv:
        a = [x for x in sys.argv[1:] if not x.startswith("--")]
        print(json.dumps(run_posthoc(int(a[0]) if a else 80)))
        sys.exit(0)
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
    window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else "default"      # (--window 0: every accepted frame)
    history = int(sys.argv[sys.argv.index("--history") + 1]) if "--history" in sys.argv else None
    screen = float(sys.argv[sys.argv.index("--screen") + 1]) if "--screen" in sys.argv else None
    a = [x for i, x in enumerate(sys.argv[1:], 1) if not x.startswith("--") and sys.argv[i - 1] not in ("--seed", "--window", "--out", "--screen", "--history")]
    out_dir = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    a = [x for x in a if x != out_dir]
    print(json.dumps(run(int(a[0]) if a else None, "keyframe" if "--ba" in sys.argv else None, seed, "--host" not in sys.argv,
                         "--reassociate" in sys.argv, window, out_dir, screen, noise="reference" if "--reference-noise" in sys.argv else None,
                         engine="host" if "--host-ba" in sys.argv else "device", history=history)))
