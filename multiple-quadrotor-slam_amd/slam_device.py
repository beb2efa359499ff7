"""
The reference's monocular per-frame loop (Work/SLAM/application/own/slam2.py:360-695 `handle_new_frame`, :1021-1253
`main`; BASELINE configs[4]) with its state RESIDENT ON THE DEVICE: the live tracks, their base-keyframe positions, the map
and the poses never leave the GPU; a frame is one library call (`mqs_slam_track`, csrc/slam_frame.hip) that enqueues

    pyramidal LK of the live tracks -> status / error filter + the two early gates -> RANSAC hypotheses, selection, solvePnP
    on the inliers -> outlier-ratio and reprojection gates, commit of the kept tracks, keyframe test

waits for one small result block and, on a keyframe, enqueues without waiting the keyframe step (triangulate the free
tracks against the base keyframe, refine the pose, re-triangulate), the map update, the coverage mask, goodFeaturesToTrack
and the top-up.  `slam_loop.MonoSlam` is the same state machine driven from the host (about ten host-pointer calls per
frame): same gates and thresholds (slam2.py:1070-1098); the keyframe test runs on all tracks by default, on the reference's random
quarter of them (slam2.py:48, 1088-1089) with max_homography_points="reference"; that sample and the RANSAC draws come from a
counter-based generator on the device (the reference: numpy's and OpenCV's global generators).

Images are device tensors (torch uint8, H x W, contiguous); a live system would upload them on a side stream while the
previous frame is tracked.
"""
import ctypes
import time

import numpy as np

from . import _lib
from .camera import _intr
from .slam_loop import (CORNER_QUALITY_LEVEL, HOMOGRAPHY_CONDITION_THRESHOLD, KEYPOINT_COVERAGE_RADIUS, MAX_AMOUNT_KEYPOINTS,
                        MAX_LOST_TRACKS_RATIO, MAX_OF_ERROR, MAX_SOLVEPNP_OUTLIER_RATIO, MAX_SOLVEPNP_REPROJ_ERROR)

REASONS = {0: "", 1: "lost track of too many points", 2: "fewer than 8 triangulated tracks", 3: "no RANSAC model",
           4: "PnP outlier ratio", 5: "reprojection error"}


class DeviceMonoSlam:
    def __init__(self, cameraMatrix, distCoeffs, image_shape, seed=0, device=0, max_landmarks=1 << 16, verbose=False,
                 ba_info=None, max_homography_points=0):
        """max_homography_points: size of keyframe_test's random sample of the tracks (slam2.py:48): 0 = every track (default),
        "reference" = the reference's max(4, target_amount_keypoints / 4) (:1088-1089).  On the rendered test sequence the quarter
        makes the run depend on the draw -- trajectory RMSE 0.017-0.020 for half of the seeds, 0.057-0.068 for the other half,
        against 0.020 for every seed with all tracks (profiles/r03/12_keyframe_sample_study.json) -- hence the default.
        ba_info: an optional `ba_io.BundleAdjustmentInfoContainer`; the loop then records what the reference records for
        the bundle adjuster (slam2.py:519-522, 634-641, 681-687, 1167-1169, 1204).  Recording reads the live tracks back
        after every frame (one more synchronisation per frame): the recorder's lists live on the host."""
        self.K = np.asarray(cameraMatrix, dtype=np.float64)
        self.dist = np.asarray(distCoeffs, dtype=np.float64).reshape(-1)[:4]
        self.shape = tuple(image_shape)
        H, W = self.shape
        target = int(round(W * H / (np.pi * KEYPOINT_COVERAGE_RADIUS ** 2)))         # slam2.py:1081
        self.target_keypoints = min(MAX_AMOUNT_KEYPOINTS, target)
        self.verbose = verbose
        self._intr = np.ascontiguousarray(_intr(self.K, self.dist), dtype=np.float64)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        _lib.check(L.mqs_slam_create(int(device), W, H, self._intr.ctypes.data_as(_lib.c_f64p), self.target_keypoints,
                                     float(KEYPOINT_COVERAGE_RADIUS), float(CORNER_QUALITY_LEVEL), int(max_landmarks),
                                     ctypes.c_uint64(int(seed)), ctypes.byref(self._h)))
        # keyframe_test's random sample of the tracks (slam2.py:48): "reference" = max(4, target_amount_keypoints / 4) (:1088-1089)
        self.max_homography_points = (max(4, self.target_keypoints // 4) if max_homography_points == "reference"
                                      else int(max_homography_points))
        _lib.check(L.mqs_slam_set_thresholds(self._h, MAX_OF_ERROR, MAX_LOST_TRACKS_RATIO, MAX_SOLVEPNP_REPROJ_ERROR,
                                             MAX_SOLVEPNP_OUTLIER_RATIO, HOMOGRAPHY_CONDITION_THRESHOLD, self.max_homography_points))
        self._track = L.mqs_slam_track
        self._res = np.zeros(40)
        self._pres = self._res.ctypes.data_as(_lib.c_f64p)
        self.poses = []                  # per frame: 3x4 world -> camera matrix, None for a rejected frame
        self.keyframes = []
        self.timing = []
        self.reports = []                # per frame: the first 12 result fields
        self._pending_keyframe = None    # frame index whose refined pose arrives with the next result block
        self._prev = None
        self._max_landmarks = int(max_landmarks)
        self.ba_info = ba_info
        self.history = []                # since the base keyframe: (frame, track ids, image points)
        self._key_pose = None
        if ba_info is not None:
            ba_info.set_calibration(self.K, self.dist)

    def close(self):
        if self._h:
            _lib.lib().mqs_slam_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _img_ptr(img, shape, sync=True):
        import torch
        if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous()
                and tuple(img.shape) == tuple(shape)):
            raise ValueError("images are contiguous uint8 device tensors of shape %r" % (tuple(shape),))
        # the library reads the image on its own stream and takes no event (include/mqslam.h, "Streams"): whatever the caller
        # enqueued on torch's current stream to produce it (an upload, a render) has to be complete first
        if sync:
            torch.cuda.current_stream(img.device).synchronize()
        return ctypes.c_void_p(img.data_ptr())

    def start(self, img, init_objp, init_imgp):
        """slam2.py:1136-1180: pose of the first frame from known 3-D points, then the first batch of free tracks."""
        o = np.ascontiguousarray(init_objp, dtype=np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(init_imgp, dtype=np.float32).reshape(-1, 2)
        pose = np.zeros(12)
        _lib.check(_lib.lib().mqs_slam_start(self._h, self._img_ptr(img, self.shape), o.ctypes.data_as(_lib.c_f32p),
                                             m.ctypes.data_as(_lib.c_f32p), len(o), pose.ctypes.data_as(_lib.c_f64p)))
        self.poses.append(pose.reshape(3, 4).copy())
        self.keyframes.append(0)
        self._prev = img
        if self.ba_info is not None:                         # slam2.py:1167-1169, 1184-1185
            pts, _, _, tid = self.tracks()
            self.ba_info.set_point3DAddedIdxs(np.arange(len(o)))
            self.ba_info.add_points2D_3Dassoc(m, np.arange(len(o)), 0)
            self.history = [(0, tid.copy(), pts.copy())]
            self._key_pose = self.poses[0]
        return self.poses[0]

    def _take_keyframe_report(self):
        r = self._res
        if r[24] != 0.0 and self._pending_keyframe is not None:
            self.poses[self._pending_keyframe] = r[28:40].reshape(3, 4).copy()
            self._pending_keyframe = None

    def handle_new_frame(self, img):
        """Returns 0 (rejected), 1 (frame) or 2 (keyframe), like the reference's `ret`."""
        t0 = time.perf_counter()
        if self.ba_info is not None:
            self.ba_info.next_step()                         # slam2.py:1204: one step per frame, rejected ones included
        rc = self._track(self._h, self._img_ptr(self._prev, self.shape, sync=False), self._img_ptr(img, self.shape), self._pres)
        if rc != 0:
            _lib.check(rc)
        r = self._res
        self._take_keyframe_report()
        decision = int(r[0])
        if decision and self.ba_info is not None:
            self.poses.append(r[12:24].reshape(3, 4).copy())
            if decision == 2:
                self.keyframes.append(len(self.poses) - 1)
            self._record(decision, int(r[11]))
            self._prev = img
            self.reports.append(r[:12].copy())
            self.timing.append(time.perf_counter() - t0)
            return decision
        if decision == 0:
            if self.verbose:
                print("REJECTED:", REASONS.get(int(r[1]), "?"))
            self.poses.append(None)
        else:
            self.poses.append(r[12:24].reshape(3, 4).copy())
            if decision == 2:
                self._pending_keyframe = len(self.poses) - 1
                self.keyframes.append(len(self.poses) - 1)
            self._prev = img             # slam2.py keeps the previous image of a rejected frame
        self.reports.append(r[:12].copy())
        self.timing.append(time.perf_counter() - t0)
        return decision

    def _record(self, decision, landmarks_before):
        """The recorder's share of a frame (see MonoSlam._frame), from the tracks as the frame -- and, on a keyframe, its
        keyframe branch -- left them."""
        frame_idx = len(self.poses) - 1
        if decision == 2:
            rep = np.zeros(40)
            _lib.check(_lib.lib().mqs_slam_flush(self._h, rep.ctypes.data_as(_lib.c_f64p)))      # the keyframe branch has run
            if rep[24] != 0.0:
                self.poses[frame_idx] = rep[28:40].reshape(3, 4).copy()
        pts, _, lm, tid = self.tracks()
        old = (lm >= 0) & (lm < landmarks_before)            # landmark tracks as the frame's pose saw them (slam2.py:519-522)
        self.history.append((frame_idx, tid.copy(), pts.copy()))
        self.ba_info.add_points2D_3Dassoc(pts[old], lm[old], frame_idx)
        if decision != 2:
            return
        new = lm >= landmarks_before
        if new.any():
            # slam2.py:634-641: the new landmarks and their image points in every frame since the base keyframe
            ids, new_tid = lm[new].astype(np.int64), tid[new]
            self.ba_info.set_point3DAddedIdxs(ids)
            for ev_frame, ev_tid, ev_pts in self.history:
                pos = {t: k for k, t in enumerate(ev_tid)}
                sel = np.array([pos[t] for t in new_tid], dtype=np.int64)
                self.ba_info.add_points2D_3Dassoc(ev_pts[sel], ids, ev_frame)
        P1 = np.vstack([self.poses[frame_idx], [0, 0, 0, 1.0]])                                   # slam2.py:681-687
        P0 = np.vstack([self._key_pose, [0, 0, 0, 1.0]])
        self.ba_info.add_odometry(P1 @ np.linalg.inv(P0), self.history[0][0], frame_idx)
        self.history = [(frame_idx, tid.copy(), pts.copy())]
        self._key_pose = self.poses[frame_idx]

    def finish(self):
        """Waits for the last keyframe branch and takes its report."""
        _lib.check(_lib.lib().mqs_slam_flush(self._h, self._pres))
        self._take_keyframe_report()

    # ---- state read-back (tests, recorders) --------------------------------------------
    def tracks(self):
        """(pts (n, 2) f32, base (n, 2) f32, landmark id or -1 (n,), track id (n,))."""
        cap = 512
        pts, base = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        lm, tid = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = ctypes.c_int32(0)
        _lib.check(_lib.lib().mqs_slam_read_tracks(self._h, pts.ctypes.data_as(_lib.c_f32p), base.ctypes.data_as(_lib.c_f32p),
                                                   lm.ctypes.data_as(_lib.c_i32p), tid.ctypes.data_as(_lib.c_i32p), cap, ctypes.byref(n)))
        k = min(n.value, cap)
        return pts[:k], base[:k], lm[:k], tid[:k]

    @property
    def objp(self):
        """The map, float32 (n, 3) like slam2.py:19."""
        n = ctypes.c_int32(0)
        _lib.check(_lib.lib().mqs_slam_read_map(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        _lib.check(_lib.lib().mqs_slam_read_map(self._h, out.ctypes.data_as(_lib.c_f32p), n.value, ctypes.byref(n)))
        return out[:n.value]

    def projection_matrices(self):
        return [None if p is None else p.copy() for p in self.poses]

    def trajectory(self):
        """Camera centres (F, 3), NaN for rejected frames."""
        out = np.full((len(self.poses), 3), np.nan)
        for i, P in enumerate(self.poses):
            if P is not None:
                out[i] = -P[:, :3].T @ P[:, 3]
        return out
