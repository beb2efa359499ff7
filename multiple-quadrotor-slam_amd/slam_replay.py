"""
Replay of the triangulation step of the reference's per-keyframe SLAM loop from
RECORDED feature tracks instead of images (the images are not in the reference; the tracks, poses and
the resulting map are).

At every keyframe `handle_new_frame` (Work/SLAM/application/own/slam2.py:541-590) undistorts the
not-yet-triangulated image points of the previous keyframe and of the current frame
(cv2.undistortPoints :551-552), triangulates them with `iterative_LS_triangulation` against the two
poses (:553-555, again at :582-584 with the refined pose -- the one that is kept), and keeps the points
with status >= 0 (:589).  The recorded BA_info files list, for every step, the landmarks created there
(`point3DAddedIdxs`) and their 2-D observations in every frame since the previous keyframe
(`point2D3DAssocs`, slam2.py:628-635); the trajectory file holds the (refined) poses.  Replaying the
step on those inputs must therefore reproduce the reference's own map (`map_out-<name>.pcd`, stored
as float32) -- which tests/test_replay.py asserts to 1e-5.

`triangulate(u0_pixels, u1_pixels, K, dist, P0, P1)` is pluggable so that the tests can run the same
replay with the oracle; the default is the GPU path (camera.undistort_points + triangulation).
"""
import time

import numpy as np

from . import camera
from . import triangulation


def world_to_camera(pose12):
    """Trajectory poses are camera-to-world (IO.hpp:430-436); triangulation wants P = [R^T | -R^T t]."""
    R = pose12[:9].reshape(3, 3)
    t = pose12[9:]
    return np.concatenate([R.T, (-R.T @ t).reshape(3, 1)], axis=1)


def gpu_triangulate(p0, p1, K, dist, P0, P1):
    u0 = camera.undistort_points(p0, K, dist)
    u1 = camera.undistort_points(p1, K, dist)
    return triangulation.iterative_LS_triangulation(u0, P0, u1, P1)


def replay_keyframe_triangulation(data, cam=0, triangulate=gpu_triangulate):
    """
    data: ba_io.BAData.  Returns dict(points (N,3) float64 with NaN where not replayed, status (N,) int32,
    keyframes = [(step, previous keyframe frame, n points, seconds)], n_triangulated).
    """
    cal = data.calibrations[cam]
    if cal[2] != 0.0:
        raise NotImplementedError("shear is not part of the OpenCV camera matrix used by slam2.py")
    K = np.array([[cal[0], 0.0, cal[3]], [0.0, cal[1], cal[4]], [0.0, 0.0, 1.0]])
    dist = np.array([cal[5], cal[6], cal[7], cal[8]])
    n = len(data.points3D)
    points = np.full((n, 3), np.nan)
    status = np.zeros(n, dtype=np.int32)
    keyframes = []
    for s in range(1, len(data.point3DAddedIdxs)):
        new = data.point3DAddedIdxs[s]
        if not new:
            continue
        new_set = set(new)
        assocs = [a for a in data.point2D3DAssocs[cam][s] if a[2] in new_set]
        f0 = min(a[0] for a in assocs)                      # tracking_history[0]: the previous keyframe
        o0 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in assocs if a[0] == f0}
        o1 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in assocs if a[0] == s}
        ids = [p for p in new if p in o0 and p in o1]
        if not ids:
            continue
        p0 = np.array([o0[p] for p in ids], dtype=np.float64)
        p1 = np.array([o1[p] for p in ids], dtype=np.float64)
        P0 = world_to_camera(data.poses[cam][f0][1])
        P1 = world_to_camera(data.poses[cam][s][1])
        t0 = time.perf_counter()
        x, st = triangulate(p0, p1, K, dist, P0, P1)
        dt = time.perf_counter() - t0
        points[ids] = x
        status[ids] = st
        keyframes.append((s, f0, len(ids), dt))
    return dict(points=points, status=status, keyframes=keyframes,
                n_triangulated=int(np.isfinite(points[:, 0]).sum()))


# ---------------------------------------------------------------------------------------------------
# Full per-frame replay: pose (solvePnP) -> triangulate -> refined pose -> re-triangulate
# ---------------------------------------------------------------------------------------------------
def camera_to_world(P):
    """[R | t] (world -> camera) -> trajectory pose12 (camera-to-world R row-major, t)."""
    R, t = P[:, :3], P[:, 3]
    return np.concatenate([R.T.reshape(-1), -R.T @ t])


def gpu_solve_pnp(objp, imgp, intr, P_start):
    from . import pnp
    return pnp.solve_pnp_pose(objp, imgp, intr, P_start)[0]


def replay_frames(data, cam=0, solve_pnp=gpu_solve_pnp, triangulate=gpu_triangulate, chained=True, fused=None):
    """
    Replays `handle_new_frame` (Work/SLAM/application/own/slam2.py:360-695) for every recorded frame from the
    recorded 2-D tracks alone: the only 3-D input is the initial map (the landmarks of step 0) and the first pose.

      every frame   pose = solvePnP(tracked already-triangulated landmarks, start = previous pose)   (:453-490; the
                    recorded tracks hold the RANSAC inliers only, so the RANSAC pass itself has nothing to reject)
      keyframes     (frames that add landmarks) triangulate the new points against the base keyframe with that pose
                    (:551-555), keep status == 1 (:556), cast to float32 (:19), refine the pose on old + new points
                    (:576-577), re-triangulate with the refined pose (:582-584) and keep status >= 0 (:589).

    chained=True uses the replay's OWN poses / map throughout (errors may accumulate); False restarts every frame
    from the recorded previous pose and recorded map (isolates the per-frame arithmetic).
    fused: one `pnp.keyframe_step` call per frame (one launch: both poses, both triangulations, the undistortions) instead
    of up to eight host-pointer calls -- the default whenever the two solvers are the GPU ones; same results.
    Returns dict(poses (F, 12) camera-to-world, points (N, 3) with NaN where never triangulated, status,
                 frames = [(frame, n tracked, n new, seconds)]).
    """
    cal = data.calibrations[cam]
    if cal[2] != 0.0:
        raise NotImplementedError("shear is not part of the OpenCV camera matrix used by slam2.py")
    K = np.array([[cal[0], 0.0, cal[3]], [0.0, cal[1], cal[4]], [0.0, 0.0, 1.0]])
    dist = np.array([cal[5], cal[6], cal[7], cal[8]])
    intr = np.array([cal[0], cal[1], cal[3], cal[4], cal[5], cal[6], cal[7], cal[8], 0.0])
    n = len(data.points3D)
    F = len(data.point2D3DAssocs[cam])
    points = np.full((n, 3), np.nan)
    status = np.zeros(n, dtype=np.int32)
    init = list(data.point3DAddedIdxs[0])
    points[init] = data.points3D[init]                                  # the initial map (slam2.py:1150-1160)
    recorded = np.array([data.poses[cam][f][1] for f in range(F)])
    poses = np.full((F, 12), np.nan)
    poses[0] = recorded[0]
    added = {}
    for s, ids in enumerate(data.point3DAddedIdxs):
        for p in ids:
            added[p] = s
    if fused is None:
        fused = solve_pnp is gpu_solve_pnp and triangulate is gpu_triangulate
    frames = []
    if fused:
        # The recorded file structures are decoded into per-frame arrays first (that is reading the recording, not the
        # frame step); the timed step is then what the live loop does per frame: gather the tracked landmarks, ONE library
        # call, scatter the new landmarks.
        from . import pnp as _pnp
        prep = []
        t_prep0 = time.perf_counter()
        for f in range(1, F):
            assocs = data.point2D3DAssocs[cam][f]
            old = [(i2, p3) for (fr, i2, p3) in assocs if fr == f and added[p3] < f]
            rec = dict(ids_old=np.array([p3 for _, p3 in old], dtype=np.int64),
                       uv_old=np.array([data.points2D[cam][f][i2] for i2, _ in old], dtype=np.float64).reshape(-1, 2),
                       ids=None)
            new = data.point3DAddedIdxs[f]
            if new:
                new_set = set(new)
                asn = [a for a in assocs if a[2] in new_set]
                f0 = min(a[0] for a in asn)
                o0 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in asn if a[0] == f0}
                o1 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in asn if a[0] == f}
                ids = [p for p in new if p in o0 and p in o1]
                if ids:
                    rec.update(ids=np.array(ids, dtype=np.int64), f0=f0,
                               p0=np.array([o0[p] for p in ids], dtype=np.float64),
                               p1=np.array([o1[p] for p in ids], dtype=np.float64))
            prep.append(rec)
        prep_seconds = time.perf_counter() - t_prep0             # decoding the recording: reported beside the frames' times
        src_pts = points if chained else data.points3D
        step = _pnp.KeyframeStepper(intr)                      # the library call with its buffers and pointers set up once
        Pw = np.full((F, 3, 4), np.nan)                        # world -> camera matrices of the poses so far
        Pw[0] = world_to_camera(poses[0])
        if not chained:
            for f in range(F):
                Pw[f] = world_to_camera(recorded[f])
        for f in range(1, F):
            rec = prep[f - 1]
            t0 = time.perf_counter()
            X_old = src_pts[rec["ids_old"]]
            n_new = 0
            if rec["ids"] is not None:
                P1, x2, st2 = step(X_old, rec["uv_old"], Pw[f - 1], rec["p0"], rec["p1"], Pw[rec["f0"]])
                keep = st2 >= 0
                kept = rec["ids"][keep]
                points[kept] = x2[keep].astype(np.float32)
                status[kept] = st2[keep]
                n_new = len(kept)
            else:
                P1, _, _ = step(X_old, rec["uv_old"], Pw[f - 1])
            if chained:
                Pw[f] = P1
            R = P1[:, :3]
            poses[f, :9] = R.T.reshape(-1)
            poses[f, 9:] = -R.T @ P1[:, 3]
            frames.append((f, len(rec["ids_old"]), n_new, time.perf_counter() - t0))
        return dict(poses=poses, points=points, status=status, frames=frames, prep_seconds=prep_seconds)
    for f in range(1, F):
        t0 = time.perf_counter()
        assocs = data.point2D3DAssocs[cam][f]
        old = [(i2, p3) for (fr, i2, p3) in assocs if fr == f and added[p3] < f]
        ids_old = [p3 for _, p3 in old]
        uv_old = np.array([data.points2D[cam][f][i2] for i2, _ in old], dtype=np.float64)
        src_pts = points if chained else data.points3D
        X_old = np.asarray(src_pts[ids_old], dtype=np.float64)
        P_prev = world_to_camera(poses[f - 1] if chained else recorded[f - 1])
        new = data.point3DAddedIdxs[f]
        n_new = 0
        P1 = solve_pnp(X_old, uv_old, intr, P_prev)
        if new:
            new_set = set(new)
            asn = [a for a in assocs if a[2] in new_set]
            f0 = min(a[0] for a in asn)                                 # tracking_history[0]: the base keyframe
            o0 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in asn if a[0] == f0}
            o1 = {a[2]: data.points2D[cam][a[0]][a[1]] for a in asn if a[0] == f}
            ids = [p for p in new if p in o0 and p in o1]
            p0 = np.array([o0[p] for p in ids], dtype=np.float64)
            p1 = np.array([o1[p] for p in ids], dtype=np.float64)
            P0 = world_to_camera(poses[f0] if chained else recorded[f0])
            x1, st1 = triangulate(p0, p1, K, dist, P0, P1)
            ok = np.asarray(st1) == 1
            X_all = np.concatenate([X_old, np.asarray(x1)[ok].astype(np.float32).astype(np.float64)])
            uv_all = np.concatenate([uv_old, p1[ok]])
            P2 = solve_pnp(X_all, uv_all, intr, P1)
            ids_ok = [p for p, k in zip(ids, ok) if k]
            x2, st2 = triangulate(p0[ok], p1[ok], K, dist, P0, P2)
            keep = np.asarray(st2) >= 0
            kept = [p for p, k in zip(ids_ok, keep) if k]
            points[kept] = np.asarray(x2)[keep].astype(np.float32)
            status[kept] = np.asarray(st2)[keep]
            n_new = len(kept)
            P1 = P2
        poses[f] = camera_to_world(P1)
        frames.append((f, len(old), n_new, time.perf_counter() - t0))
    return dict(poses=poses, points=points, status=status, frames=frames)
