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
