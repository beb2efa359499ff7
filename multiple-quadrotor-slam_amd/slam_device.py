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

REASSOCIATE_RADIUS = 2.0         # pixels: a re-detected corner within this distance of a lost landmark's projection (slam.py:29 max_radius_OF_to_FAST["FAST"])
REASSOCIATE_RATIO = 0.7          # slam.py:30 max_dist_ratio["FAST"]
# BA_info.noise.*-slam2.txt beside the reference's recording of its ICL-NUIM example sequence (Work/SLAM/datasets/ICL_NUIM/living_room_traj3n_frei_png)
REFERENCE_NOISE = {"point3D": 0.2, "pose": (0.02, 0.02, 0.02, 0.1, 0.1, 0.1), "odometry": (0.05, 0.05, 0.05, 0.2, 0.2, 0.2), "point2D": 1.0}



class _BaParams(ctypes.Structure):                       # include/mqslam.h: mqs_slam_ba_params
    _fields_ = ([(k, ctypes.c_int32) for k in ("max_iterations", "min_observations", "max_passes", "add_odometry_edge", "edge_from", "edge_to",
                                                "damping", "workgroups", "screen_iterations", "reserved")] +
                [(k, ctypes.c_double) for k in ("outlier_px", "gross_px", "border_margin_px", "min_depth_ratio", "point_sigma", "pixel_sigma")] +
                [("pose_sigmas", ctypes.c_double * 6), ("odometry_sigmas", ctypes.c_double * 6)] +
                [(k, ctypes.c_double) for k in ("lambda_initial", "lambda_factor", "lambda_upper", "abs_tol", "rel_tol")])


class _BaWindow(ctypes.Structure):                       # include/mqslam.h: mqs_slam_ba_window
    _fields_ = [("n_poses", ctypes.c_int32), ("second_anchor", ctypes.c_int32), ("carry_unselected", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("poses", ctypes.POINTER(ctypes.c_int32)), ("seen_outside_sigma", ctypes.c_double)]


MQS_E_TIMEOUT, MQS_E_CAPACITY = -6, -7                   # include/mqslam.h


REASONS = {0: "", 1: "lost track of too many points", 2: "fewer than 8 triangulated tracks", 3: "no RANSAC model",
           4: "PnP outlier ratio", 5: "reprojection error"}


def _reprojection_residuals(poses_c2w, points, calib, lm, pose_idx, uv):
    """|projection - measurement| in pixels per observation; poses camera-to-world pose12, calib = fx fy s u0 v0 k1 k2 p1 p2
    (the Cal3DS2 model of csrc/ba_math.h), vectorised numpy: the screen of the in-loop adjustment, not a product kernel."""
    R = poses_c2w[pose_idx, :9].reshape(-1, 3, 3)
    d = points[lm] - poses_c2w[pose_idx, 9:]
    X = np.einsum("nji,nj->ni", R, d)                                  # R^T (p - c)
    z = np.where(X[:, 2] > 1e-9, X[:, 2], 1e-9)
    x, y = X[:, 0] / z, X[:, 1] / z
    fx, fy, sk, u0, v0, k1, k2, p1, p2 = calib
    r2 = x * x + y * y
    g = 1.0 + k1 * r2 + k2 * r2 * r2
    xd = g * x + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = g * y + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y)
    return np.hypot(fx * xd + sk * yd + u0 - uv[:, 0], fy * yd + v0 - uv[:, 1])


class _Slot:
    """An image in the handle's ingest ring (`FrameUploader`): what `DeviceMonoSlam.start` / `handle_new_frame` take instead of a
    device tensor."""
    __slots__ = ("index", "source", "ptr", "next")

    def __init__(self, index, source):
        self.index, self.source, self.ptr = index, source, None      # source: the host array, kept alive until the frame is done
        self.next = None                                             # the slot of the frame behind this one, once posted (FrameUploader)


class FrameUploader:
    """Frame ingest of the device-resident loop: the reference reads every frame INSIDE its loop (slam2.py:1209-1213: cv2.imread per
    iteration); here a frame arrives in host memory and goes to the device on a stream of its own while the loop's kernels work on
    the frames before it (csrc/slam_ingest.hip: a ring of device images owned by the handle, the library's worker thread enqueues the
    copies; the loop's stream waits for a frame's upload ON THE DEVICE).  Iterating yields one ring slot per frame, `ahead` uploads
    posted in front of the one being processed:

        for img in FrameUploader(slam, frames):
            slam.handle_new_frame(img)                  # (the first one: slam.start(img, ...))

    frames: a sequence of H x W uint8 numpy arrays (ordinary memory: staged through the slot's pinned buffer by the worker thread), or
    ONE pinned uint8 torch tensor [n, H, W] (a capture buffer a driver delivers into: no staging copy)."""

    def __init__(self, slam, frames, ahead=2):
        self._slam, self._frames, self._n, self._ahead = slam, frames, len(frames), max(1, int(ahead))
        self._pinned = not isinstance(frames, (list, tuple, np.ndarray))
        if self._pinned:
            import torch
            if not (isinstance(frames, torch.Tensor) and frames.dtype == torch.uint8 and frames.is_pinned() and frames.dim() == 3 and frames.is_contiguous()
                    and tuple(frames.shape[1:]) == tuple(slam.shape)):
                raise ValueError("a torch source is one pinned contiguous uint8 tensor [n, H, W]")
        slam._ingest_enable(self._ahead + 3)            # in use at once: the previous image, the current one, `ahead` uploads, one spare

    def __len__(self):
        return self._n

    def _post(self, k):
        s = self._slam
        if self._pinned:
            src = self._frames[k]
            ptr = src.data_ptr()
        else:
            src = np.ascontiguousarray(self._frames[k], dtype=np.uint8)
            if src.shape != tuple(s.shape):
                raise ValueError("frames are H x W uint8 arrays of shape %r" % (tuple(s.shape),))
            ptr = src.ctypes.data
        slot = _Slot(s._ingest_free.pop(), src)
        _lib.check(_lib.lib().mqs_slam_upload(s._h, slot.index, ctypes.c_void_p(ptr), int(self._pinned)))
        return slot

    def __iter__(self):
        from collections import deque
        q, k, last = deque(), 0, None
        while k < self._n or q:
            while k < self._n and len(q) <= self._ahead:
                slot = self._post(k)
                if q:
                    q[-1].next = slot
                elif last is not None:
                    last.next = slot
                q.append(slot)
                k += 1
            last = q.popleft()
            yield last


class DeviceMonoSlam:
    def __init__(self, cameraMatrix, distCoeffs, image_shape, seed=0, device=0, max_landmarks=1 << 16, verbose=False,
                 ba_info=None, max_homography_points=0, bundle_adjust=None, ba_iterations=10, ba_log_capacity=1 << 20,
                 reassociate=False, ba_window_keyframes=3, second_pass_screen=None, ba_engine="device", ba_check=False,
                 ba_history_keyframes=None, ba_noise=None, ba_screen_iterations=3):
        """max_homography_points: size of keyframe_test's random sample of the tracks (slam2.py:48): 0 = every track (default),
        "reference" = the reference's max(4, target_amount_keypoints / 4) (:1088-1089).  On the rendered test sequence the quarter
        makes the run depend on the draw -- trajectory RMSE 0.017-0.020 for half of the seeds, 0.057-0.068 for the other half,
        against 0.020 for every seed with all tracks (profiles/r03/12_keyframe_sample_study.json) -- hence the default.
        bundle_adjust="keyframe": BASELINE configs[4] as written -- detect -> track -> triangulate -> BUNDLE-ADJUST per keyframe,
        inside the loop: the frame kernels log every observation the reference's recorder is handed (slam2.py:519-522, 634-641,
        1167-1169) in device memory; behind every keyframe the log becomes a `ba_io.SparseProblem` (all accepted frames so far, all
        landmarks; gauge as the reference's tool fixes it, bundle_adjust.cpp:268-282: a pose prior on the first frame, point priors
        on the first frame's landmarks), `sparse_ba.SparseBundleAdjuster` runs Levenberg-Marquardt on it (<= ba_iterations), and
        the adjusted landmarks and poses go back into the loop's live state (`mqs_slam_write_back`), so the next frames are
        tracked against the adjusted map.  `poses` then holds the ADJUSTED pose of every accepted frame up to the last
        keyframe; `poses_online` keeps each frame's pose as it was first estimated.
        ba_engine: "device" (default) -- the adjustment is ONE library call on the resident state (`mqs_slam_bundle_adjust`,
        csrc/slam_ba.hip: a persistent launch builds the problem from the device log, runs every Levenberg-Marquardt trial and the
        screens, writes the result back; the host sees a 16-double report and the adjusted poses); "host" -- round 4's path, kept as
        its twin for the tests and for ba_window_keyframes: the log comes to the host, numpy builds the problem, `sparse_ba.
        SparseBundleAdjuster` adjusts it, `mqs_slam_write_back` returns it.  ba_check=True runs the host twin first WITHOUT writing
        anything back and keeps (host, device) result pairs in `ba_checks`.
        ba_window_keyframes=K (>= 1; default 3; None / 0: every accepted frame -- what the reference's tool does with a finished recording,
        bundle_adjust.cpp:190-330): the problem's poses are a SELECTION of the accepted frames (`mqs_slam_bundle_adjust_window`) --
        every frame since the K-th keyframe from the end, and of the frames in front of those the KEYFRAMES only (all of them;
        ba_history_keyframes=H: the last H).  The frames left out keep their pose relative to the keyframe in front of them; what they
        saw of a landmark counts towards its three sightings and -- `ba_window_point_sigma` -- stays with it as a prior at its current
        value.  While frame 0 is selected the gauge is the plain adjustment's; once it is not, the first selected pose and the next
        selected keyframe are held by priors at their current values.  The cost of an adjustment then grows with the keyframes of the
        run (or, with H, not at all) instead of with its frames, and runs of any length stay on the device engine: beyond 256
        accepted frames a selection is made in any case (every keyframe + the latest frames that fit).  On the 200 frames of the
        reference's example sequence, 16 seeds (profiles/r06): every frame 5.1 mm median / 8.0 worst at 1 575 frames/s; K = 3:
        4.8 / 7.2 at 3 130; K = 2: 5.1 / 9.4 at 3 440; K = 4: 4.5 / 7.3 at 2 890.
        second_pass_screen=px (None / 0: off, the reference's flow): at a keyframe a freshly triangulated point whose reprojection error
        in the current frame exceeds px is not handed to the second solvePnP (slam2.py:576-577) -- the use slam2.py:1092 announces for
        max_2nd_solvePnP_reproj_error (= 1 px) and never makes; see mqs_slam_set_second_pass_screen.
        ba_screen_iterations=K (default 3; 0: off): the residual screen behind an adjustment also looks behind every K of its
        iterations (`mqs_slam_ba_params.screen_iterations`): a pass that carries a mistracked corner is thrown away after K trials
        instead of after `ba_iterations` -- 76 % of the trials of a run on the reference's example sequence were spent in such
        passes.  16 seeds, 200 frames: 4.6 mm median / 7.2 worst at 3 510 frames/s against 4.8 / 7.2 at 3 140 with K = 0
        (K = 2: 5.0 / 8.5 at 3 540; profiles/r06).
        ba_noise: the adjustment's noise models, as the reference keeps them in the four BA_info.noise.* files beside a recording
        (`ba_io.load_data` reads them: poseNoise, odometryNoise, point3DNoise, point2DNoise) -- a dict with any of "point3D" (sigma of the prior on the start-up landmarks), "pose" (6: prior
        on the first pose, rotation then translation), "odometry" (6: keyframe -> keyframe between-factors), "point2D" (pixels); or
        "reference": the values of the reference's ICL-NUIM recording (REFERENCE_NOISE below).  Default: this build's (a 0.05 m
        start-up prior instead of 0.2 -- inside the loop the start-up points are what the first pose was COMPUTED from; a 2 mrad /
        1 mm first-pose prior; the reference's odometry and pixel sigmas): profiles/r05/07 has the measurement behind the deviation.
        reassociate=True: behind every keyframe's top-up the new corners are matched (BFMatcher.radiusMatch on pixel positions +
        ratio test + one match per corner: the reference's match_OF_based, slam.py:81-127, cv2_helpers.py:296-339) against the
        PROJECTIONS of the landmarks that are in the map but not tracked any more; a matched corner takes its landmark up again
        instead of starting a new one (`mqs_slam_reassociate`).
        ba_info: an optional `ba_io.BundleAdjustmentInfoContainer`; the loop then records what the reference records for
        the bundle adjuster (slam2.py:519-522, 634-641, 681-687, 1167-1169, 1204).  Recording reads the live tracks back
        after every frame (one more synchronisation per frame): the recorder's lists live on the host."""
        self.K = np.asarray(cameraMatrix, dtype=np.float64)
        d5 = np.zeros(5)
        dc = np.asarray(distCoeffs, dtype=np.float64).reshape(-1)
        d5[:min(5, dc.size)] = dc[:5]
        self.dist = d5[:4]                       # k1 k2 p1 p2: the bundle adjuster's camera model (IO.hpp:230-236 has no k3)
        if bundle_adjust and d5[4] != 0.0:
            raise ValueError("bundle_adjust: the adjuster's camera model (fx fy s u0 v0 k1 k2 p1 p2, IO.hpp:230-236) has no k3; got k3 = %g" % d5[4])
        self.shape = tuple(image_shape)
        H, W = self.shape
        target = int(round(W * H / (np.pi * KEYPOINT_COVERAGE_RADIUS ** 2)))         # slam2.py:1081
        self.target_keypoints = min(MAX_AMOUNT_KEYPOINTS, target)
        self.verbose = verbose
        self._intr = np.ascontiguousarray(_intr(self.K, d5), dtype=np.float64)          # the frame kernels take k3 as well
        self._h = ctypes.c_void_p()
        self._device = int(device)
        L = _lib.lib()
        _lib.check(L.mqs_slam_create(int(device), W, H, self._intr.ctypes.data_as(_lib.c_f64p), self.target_keypoints,
                                     float(KEYPOINT_COVERAGE_RADIUS), float(CORNER_QUALITY_LEVEL), int(max_landmarks),
                                     ctypes.c_uint64(int(seed)), ctypes.byref(self._h)))
        # keyframe_test's random sample of the tracks (slam2.py:48): "reference" = max(4, target_amount_keypoints / 4) (:1088-1089)
        self.max_homography_points = (max(4, self.target_keypoints // 4) if max_homography_points == "reference"
                                      else int(max_homography_points))
        _lib.check(L.mqs_slam_set_thresholds(self._h, MAX_OF_ERROR, MAX_LOST_TRACKS_RATIO, MAX_SOLVEPNP_REPROJ_ERROR,
                                             MAX_SOLVEPNP_OUTLIER_RATIO, HOMOGRAPHY_CONDITION_THRESHOLD, self.max_homography_points))
        if second_pass_screen:
            _lib.check(L.mqs_slam_set_second_pass_screen(self._h, float(second_pass_screen)))
        self._track = L.mqs_slam_track
        self._res = np.zeros(40)
        self._pres = self._res.ctypes.data_as(_lib.c_f64p)
        self.poses = []                  # per frame: 3x4 world -> camera matrix, None for a rejected frame
        self.keyframes = []
        self.timing = []
        self.reports = []                # per frame: the first 12 result fields
        self._pending_keyframe = None    # frame index whose refined pose arrives with the next result block
        self._prev = None
        self._ingest_free = None         # free slots of the ingest ring (FrameUploader), None: no ring
        self.prepare_next = True         # the next pair's pyramid ahead of its frame, where the next image is known (handle_new_frame)
        self.pipeline = True             # ... and that frame's pose kernels enqueued behind this frame's (mqs_slam_pipeline): no host round trip
        self._pipeline_on = False        #     between two ordinary frames; off with the recorder (it reads the tracks after every frame)
        self._max_landmarks = int(max_landmarks)
        self.ba_info = ba_info
        if bundle_adjust not in (None, "keyframe"):
            raise ValueError("bundle_adjust: None or 'keyframe'")
        self.bundle_adjust = bundle_adjust
        self.ba_iterations = int(ba_iterations)
        self.ba_min_observations = 3
        self.ba_outlier_pixels = 4.0     # a landmark with a residual beyond this after an adjustment is a mistracked corner
        self.ba_gross_pixels = 40.0      # ... and with one beyond this BEFORE the adjustment it does not enter it
        self.ba_max_passes = 4           # adjust, screen, adjust again from the same start: at most this many adjustments
        self.ba_screen_iterations = int(ba_screen_iterations or 0)    # K > 0: the screen also looks behind every K iterations of an adjustment (mqs_slam_ba_params.screen_iterations)
        self.ba_border_margin = 10.0     # observations nearer to the image border than half a 21 x 21 tracker window stay out of the adjustment
        self.ba_min_depth_ratio = 0.02   # a landmark closer to one of its cameras than this fraction of the median landmark depth sits out
        # the noise models of the adjustment (the reference keeps them in the four BA_info.noise.* files beside a recording; its
        # ICL-NUIM run: point3D 0.2, pose (0.02 x 3, 0.1 x 3), odometry (0.05 x 3, 0.2 x 3), point2D 1.0)
        # prior on the start-up landmarks (bundle_adjust.cpp:277-281).  The reference's tool takes 0.2 m from its noise file -- a gauge, for a
        # recording adjusted afterwards.  Inside the loop the start-up points are what the loop itself treats as EXACT (the first pose is
        # computed from them, slam2.py:1136-1180): with 0.25 m the adjuster moved the 23 exact model points of the reference's example
        # sequence by 109 mm rms and ended 7.5 mm from the exact trajectory over its first 80 frames where the plain loop ends at 3.5;
        # with 0.05 m: 3.7 mm there, 5.1 (from 5.3) over all 200 frames, 2.4 (from 3.0) on the rendered 60 frames (profiles/r05)
        self.ba_point_sigma = 0.05
        self.ba_pose_sigmas = (0.002, 0.002, 0.002, 0.001, 0.001, 0.001)     # prior on the first pose (:273), rotation then translation
        self.ba_odometry_sigmas = (0.05, 0.05, 0.05, 0.2, 0.2, 0.2)  # between-factors keyframe -> keyframe (:301-309)
        self.ba_pixel_sigma = 1.0
        if ba_noise is not None:
            nz = REFERENCE_NOISE if ba_noise == "reference" else dict(ba_noise)
            unknown = set(nz) - {"point3D", "pose", "odometry", "point2D"}
            if unknown:
                raise ValueError("ba_noise: unknown keys %r (point3D, pose, odometry, point2D)" % sorted(unknown))
            self.ba_point_sigma = float(nz.get("point3D", self.ba_point_sigma))
            self.ba_pose_sigmas = tuple(float(v) for v in nz.get("pose", self.ba_pose_sigmas))
            self.ba_odometry_sigmas = tuple(float(v) for v in nz.get("odometry", self.ba_odometry_sigmas))
            self.ba_pixel_sigma = float(nz.get("point2D", self.ba_pixel_sigma))
            if len(self.ba_pose_sigmas) != 6 or len(self.ba_odometry_sigmas) != 6:
                raise ValueError("ba_noise: pose and odometry sigmas are 6 values (rotation, then translation)")
        self.ba_window_keyframes = ba_window_keyframes              # None: every frame so far; K >= 1: every frame since the K-th keyframe from the end + the keyframes in front
        self.ba_history_keyframes = ba_history_keyframes            # None: every keyframe in front of the dense part; H >= 0: the last H of them
        self.ba_window_point_sigma = 0.02                           # selection: prior on a landmark that frames outside the problem have seen (0 / None: none)
        self.ba_carry = True                                        # selection: frames left out keep their pose relative to the selected pose in front of them
        self._kf_pose = [0]                                         # pose index (rank among the accepted frames) of every keyframe
        self._ba_bad = np.zeros(0, bool)
        if ba_engine not in ("device", "host"):
            raise ValueError("ba_engine: 'device' or 'host'")
        self.ba_engine = ba_engine
        self.ba_check = bool(ba_check)
        self.ba_checks = []
        self.ba_fallbacks = []           # adjustments the resident adjuster gave up on (timeout / capacity): the host-built path took over
        self.ba_workgroups = 0           # 0: the library's choice
        self._pending_online = None      # keyframe whose refined (pre-adjustment) pose arrives with the next result block
        self.reassociate = bool(reassociate)
        self.poses_online = []           # with bundle_adjust: the pose of each frame as first estimated (poses: adjusted)
        self._accepted = []              # frame index of every accepted frame, in order (= the log's pose indices)
        self.ba_reports = []
        self._odo = []                   # (from pose index, to pose index, measured relative pose12): one edge per keyframe
        self._odo_last_to = -1           # device engine: pose index the last odometry edge ends at
        self.reassociated = 0
        if bundle_adjust:
            _lib.check(L.mqs_slam_log_enable(self._h, int(ba_log_capacity)))
            self._log_cap = int(ba_log_capacity)
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

    def _ingest_enable(self, slots):
        if self._ingest_free is None:
            _lib.check(_lib.lib().mqs_slam_ingest_enable(self._h, int(slots)))
            self._ingest_free = list(range(int(slots)))

    def _release(self, img):
        """The loop is done with an image: a ring slot goes back to the free list."""
        if isinstance(img, _Slot) and self._ingest_free is not None:
            self._ingest_free.append(img.index)

    def _img_ptr(self, img, shape, sync=True):
        if isinstance(img, _Slot):
            if img.ptr is None:                              # first use: the loop's stream waits (on the device) for the slot's upload
                p = ctypes.c_void_p()
                _lib.check(_lib.lib().mqs_slam_wait_upload(self._h, img.index, ctypes.byref(p)))
                img.ptr = p
            return img.ptr
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
        """slam2.py:1136-1180: pose of the first frame from known 3-D points, then the first batch of free tracks.
        img: a contiguous uint8 device tensor, or a slot of the ingest ring (`FrameUploader`)."""
        o = np.ascontiguousarray(init_objp, dtype=np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(init_imgp, dtype=np.float32).reshape(-1, 2)
        pose = np.zeros(12)
        _lib.check(_lib.lib().mqs_slam_start(self._h, self._img_ptr(img, self.shape), o.ctypes.data_as(_lib.c_f32p),
                                             m.ctypes.data_as(_lib.c_f32p), len(o), pose.ctypes.data_as(_lib.c_f64p)))
        self.poses.append(pose.reshape(3, 4).copy())
        self.poses_online.append(pose.reshape(3, 4).copy())
        self._accepted.append(0)
        self._n0 = len(o)
        self._objp0 = o.astype(np.float64)               # where the gauge is anchored: the KNOWN start-up points and the first pose,
        self._pose0 = pose.reshape(3, 4).copy()          # never their adjusted values (an anchor that follows the estimate drifts)
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
        if r[24] != 0.0 and self._pending_online is not None:      # the device adjuster ran behind this keyframe: `poses` is adjusted already
            self.poses_online[self._pending_online] = r[28:40].reshape(3, 4).copy()
            self._pending_online = None

    def handle_new_frame(self, img, next_img=None):
        """Returns 0 (rejected), 1 (frame) or 2 (keyframe), like the reference's `ret`.
        img: a contiguous uint8 device tensor, or a slot of the ingest ring (`FrameUploader`).
        next_img (device tensors; ring slots know their successor): the frame behind this one, if it is on the device already -- the
        tracker's pyramid of the pair (img, next_img) is then built on a side stream under this frame's pose kernels, and the
        pair is tracked there as soon as this frame's hypotheses are out (`mqs_slam_set_next`; `prepare_next = False` switches
        both off: same results, bit for bit)."""
        t0 = time.perf_counter()
        if self.ba_info is not None:
            self.ba_info.next_step()                         # slam2.py:1204: one step per frame, rejected ones included
        p_prev, p_img = self._img_ptr(self._prev, self.shape, sync=False), self._img_ptr(img, self.shape)
        want = bool(self.pipeline and self.prepare_next and self.ba_info is None)
        if want != self._pipeline_on:
            _lib.check(_lib.lib().mqs_slam_pipeline(self._h, 1 if want else 0))
            self._pipeline_on = want
        if self.prepare_next:
            # the pyramid of the NEXT pair on the library's side stream, under this frame's pose kernels: named here, enqueued by the track call
            # behind its own launches (mqs_slam_set_next / mqs_slam_prepare_next)
            nxt = next_img if next_img is not None else getattr(img, "next", None)
            if isinstance(nxt, _Slot) and isinstance(img, _Slot):
                _lib.check(_lib.lib().mqs_slam_set_next(self._h, img.index, None, nxt.index))
            elif nxt is not None and not isinstance(nxt, _Slot) and not isinstance(img, _Slot):
                _lib.check(_lib.lib().mqs_slam_set_next(self._h, -1, self._img_ptr(nxt, self.shape), -1))
        rc = self._track(self._h, p_prev, p_img, self._pres)
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
            self._release(self._prev)
            self._prev = img
            self.reports.append(r[:12].copy())
            self.timing.append(time.perf_counter() - t0)
            return decision
        if decision == 0:
            if self.verbose:
                print("REJECTED:", REASONS.get(int(r[1]), "?"))
            self.poses.append(None)
            self.poses_online.append(None)
            self._release(img)               # (its kernels have run: the call waited for the result block)
        else:
            self.poses.append(r[12:24].reshape(3, 4).copy())
            self.poses_online.append(self.poses[-1].copy())
            self._accepted.append(len(self.poses) - 1)
            if decision == 2:
                self._pending_keyframe = len(self.poses) - 1
                self.keyframes.append(len(self.poses) - 1)
                self._kf_pose.append(len(self._accepted) - 1)
                if self.reassociate:
                    self._reassociate(img)
                if self.bundle_adjust:
                    self._bundle_adjust()
            self._release(self._prev)        # (ingest ring: the image before this one is not needed any more)
            self._prev = img                 # slam2.py keeps the previous image of a rejected frame
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

    # ---- bundle adjustment inside the loop ---------------------------------------------
    def read_log(self):
        """The observation log so far: (landmark (n,), pose index (n,), pixel (n, 2))."""
        n = _lib.c_i64(0)
        L = _lib.lib()
        _lib.check(L.mqs_slam_read_log(self._h, None, None, None, 0, ctypes.byref(n)))
        m = int(n.value)
        lm, ps, uv = np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros((m, 2), np.float64)
        if m:
            _lib.check(L.mqs_slam_read_log(self._h, lm.ctypes.data_as(_lib.c_i32p), ps.ctypes.data_as(_lib.c_i32p),
                                           uv.ctypes.data_as(_lib.c_f64p), m, ctypes.byref(n)))
        return lm, ps, uv

    BA_DEVICE_MAX_POSES = 256            # include/mqslam.h: MQS_SLAM_BA_MAX_POSES (the resident adjuster stages every camera of the PROBLEM in LDS)

    def _select_poses(self):
        """The poses of this adjustment's problem: None (every accepted frame), or ascending pose indices -- the keyframes in front of
        the dense part (the last `ba_history_keyframes` of them) and every frame since the `ba_window_keyframes`-th keyframe from the
        end.  More than the adjuster holds: the oldest plain frames go first, then the oldest keyframes."""
        P, cap = len(self._accepted), self.BA_DEVICE_MAX_POSES
        Kw, H, kp = self.ba_window_keyframes, self.ba_history_keyframes, self._kf_pose
        if not Kw and P <= cap:
            return None
        dense_start = (kp[-Kw] if len(kp) >= Kw else 0) if Kw else max(0, P - cap // 2)
        hist = [k for k in kp if k < dense_start]
        if H is not None:
            hist = hist[max(0, len(hist) - H):] if H > 0 else []
        sel = hist + list(range(dense_start, P))
        over = len(sel) - cap
        if over > 0:
            kset, keep = set(kp), []
            for j in sel:
                if over > 0 and j not in kset and j != P - 1:
                    over -= 1
                    continue
                keep.append(j)
            sel = keep[over:] if over > 0 else keep
        if len(sel) == P:
            return None
        return np.asarray(sel, dtype=np.int32)

    def _second_anchor(self, sel):
        """Once frame 0 is not a pose of the problem: the next selected keyframe behind the first selected pose (two poses held by
        priors fix the seven gauge freedoms of a monocular window, scale included)."""
        if sel is None or sel[0] == 0:
            return -1
        kset = set(self._kf_pose)
        return next((int(j) for j in sel[1:] if int(j) in kset), -1)

    def _hand_over_to_the_host_engine(self):
        """The host-built path takes the adjustment over where it stands (after a launch of the resident adjuster that gave up or did
        not hold the problem) -- the retired landmarks and the odometry edges (measured when their keyframes were taken) come from the
        device, once."""
        L = _lib.lib()
        self._ba_bad = self.retired_landmarks()
        n = ctypes.c_int32(0)
        _lib.check(L.mqs_slam_read_ba_edges(self._h, None, None, None, 0, ctypes.byref(n)))
        m = n.value
        fr, to, meas = np.zeros(max(m, 1), np.int32), np.zeros(max(m, 1), np.int32), np.zeros((max(m, 1), 12))
        if m:
            _lib.check(L.mqs_slam_read_ba_edges(self._h, fr.ctypes.data_as(_lib.c_i32p), to.ctypes.data_as(_lib.c_i32p), meas.ctypes.data_as(_lib.c_f64p), m, ctypes.byref(n)))
        self._odo = [(int(fr[k]), int(to[k]), meas[k].copy()) for k in range(m)]
        self.ba_engine = "host"

    def _bundle_adjust(self):
        if self.ba_engine == "host":
            return self._bundle_adjust_host()
        if self.ba_check:
            host = self._bundle_adjust_host(write_back=False)
        rep = self._bundle_adjust_device()
        if rep is None:                                              # the launch gave up or did not hold the problem; nothing was written
            self._hand_over_to_the_host_engine()
            return self._bundle_adjust_host()
        if self.ba_check:
            self.ba_checks.append({"frame": self._accepted[-1], "host_poses": host[0], "host_points": host[1], "host_report": host[2],
                                   "device_poses": np.stack([self.poses[f] for f in self._accepted]), "device_points": self.objp.astype(np.float64),
                                   "device_report": rep, "host_retired": host[3], "device_retired": self.retired_landmarks()})

    def _ba_params(self, add_edge, e_from, e_to):
        from .bundle_adjustment import LM_ABS_TOL, LM_LAMBDA_FACTOR, LM_LAMBDA_INITIAL, LM_LAMBDA_UPPER, LM_REL_TOL
        q = _BaParams()
        q.max_iterations, q.min_observations, q.max_passes = int(self.ba_iterations), int(self.ba_min_observations), int(self.ba_max_passes)
        q.add_odometry_edge, q.edge_from, q.edge_to = int(add_edge), int(e_from), int(e_to)
        q.damping, q.workgroups = 0, int(self.ba_workgroups)
        q.screen_iterations, q.reserved = int(self.ba_screen_iterations or 0), 0
        q.outlier_px, q.gross_px = float(self.ba_outlier_pixels), float(self.ba_gross_pixels or 0.0)
        q.border_margin_px, q.min_depth_ratio = float(self.ba_border_margin or 0.0), float(self.ba_min_depth_ratio)
        q.point_sigma, q.pixel_sigma = float(self.ba_point_sigma), float(self.ba_pixel_sigma)
        for k in range(6):
            q.pose_sigmas[k] = float(self.ba_pose_sigmas[k])
            q.odometry_sigmas[k] = float(self.ba_odometry_sigmas[k])
        q.lambda_initial, q.lambda_factor, q.lambda_upper = LM_LAMBDA_INITIAL, LM_LAMBDA_FACTOR, LM_LAMBDA_UPPER
        q.abs_tol, q.rel_tol = LM_ABS_TOL, LM_REL_TOL
        return q

    def _bundle_adjust_device(self):
        """Behind a keyframe: `mqs_slam_bundle_adjust(_window)` -- one persistent launch on the handle's stream (behind the keyframe
        branch it does not wait for from here), one wait for its report and the adjusted poses.  None: the launch gave up (a grid-wide
        wait) or its resident lists do not hold the problem -- nothing was written back, the caller takes the host-built path."""
        t0 = time.perf_counter()
        kf = self._accepted[-1]
        add_edge, e_from, e_to = False, 0, 0
        if self.keyframes and self.keyframes[-1] == kf and len(self.keyframes) >= 2 and self._odo_last_to != len(self._accepted) - 1:
            add_edge, e_from, e_to = True, self._kf_pose[-2], len(self._accepted) - 1
        q = self._ba_params(add_edge, e_from, e_to)
        P = len(self._accepted)
        sel = self._select_poses()
        first = 0 if sel is None else int(sel[0])
        rep, poses = np.zeros(16), np.zeros((P - first, 12))
        L = _lib.lib()
        if sel is None:
            rc = L.mqs_slam_bundle_adjust(self._h, ctypes.byref(q), rep.ctypes.data_as(_lib.c_f64p), poses.ctypes.data_as(_lib.c_f64p), P)
        else:
            w = _BaWindow()
            w.n_poses, w.second_anchor, w.carry_unselected = len(sel), self._second_anchor(sel), int(bool(self.ba_carry))
            w.poses = sel.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            w.seen_outside_sigma = float(self.ba_window_point_sigma or 0.0) if sel[0] != 0 else 0.0      # (while frame 0 is a pose the gauge is the plain adjustment's: no such prior)
            rc = L.mqs_slam_bundle_adjust_window(self._h, ctypes.byref(q), ctypes.byref(w), rep.ctypes.data_as(_lib.c_f64p),
                                                 poses.ctypes.data_as(_lib.c_f64p), P - first)
        if rc in (MQS_E_TIMEOUT, MQS_E_CAPACITY):
            msg = L.mqs_last_error()
            self.ba_fallbacks.append({"frame": kf, "code": int(rc), "message": msg.decode() if msg else ""})
            return None
        _lib.check(rc)
        if add_edge:
            self._odo_last_to = e_to
        if self._pending_keyframe is not None:                       # its refined pose as first estimated comes with the next result block
            self._pending_online, self._pending_keyframe = self._pending_keyframe, None
        P34 = poses.reshape(P - first, 3, 4)
        for k, f in enumerate(self._accepted[first:]):               # (views of this call's own array: nothing else writes it)
            self.poses[f] = P34[k]
        out = {"frame": kf, "poses": int(rep[1]), "first_pose_of_the_window": first, "landmarks": int(rep[2]), "landmarks_adjusted": int(rep[3]),
               "observations": int(rep[4]), "passes": int(rep[5]), "landmarks_screened_out": int(rep[6]), "lm_iterations": int(rep[7]),
               "cost_before": float(rep[8]), "cost_after": float(rep[9]), "lm_trials": int(rep[10]), "repeated_observations_left_out": int(rep[11]),
               "odometry_edges": int(rep[12]), "grid_barriers": int(rep[13]), "engine": "device", "accepted_frames": P,
               "build_ms": 0.0, "adjust_ms": round(1e3 * (time.perf_counter() - t0), 3), "write_back_ms": 0.0}
        self.ba_reports.append(out)
        return out

    def retired_landmarks(self):
        """bool (landmarks,): the landmarks the in-loop adjuster has retired (mistracked corners, points at a camera centre)."""
        if self.ba_engine == "host":
            return self._ba_bad.copy()
        n = ctypes.c_int32(0)
        L = _lib.lib()
        _lib.check(L.mqs_slam_read_ba_flags(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint8)
        _lib.check(L.mqs_slam_read_ba_flags(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n.value, ctypes.byref(n)))
        return out[:n.value].astype(bool)

    def _bundle_adjust_host(self, write_back=True):
        """Behind a keyframe: the whole history so far through the sparse bundle adjuster, the result back into the live state
        (write_back=False: the live state and `poses` stay as they are;
        returns (adjusted [R | t] of the accepted frames, adjusted landmarks, report, retired flags): the device engine's twin)."""
        from . import ba_io, sparse_ba
        from .bundle_adjustment import pose_from_world_to_camera
        t0 = time.perf_counter()
        rep = np.zeros(40)
        _lib.check(_lib.lib().mqs_slam_flush(self._h, rep.ctypes.data_as(_lib.c_f64p)))          # the keyframe branch has run
        if rep[24] != 0.0 and self._pending_keyframe is not None:
            self.poses[self._pending_keyframe] = rep[28:40].reshape(3, 4).copy()
            self.poses_online[self._pending_keyframe] = self.poses[self._pending_keyframe].copy()
            self._pending_keyframe = None
        # the odometry edge of this keyframe (slam2.py:681-687: base keyframe -> keyframe, from the poses as estimated now)
        kf = self._accepted[-1]
        odo_all = list(self._odo)
        if self.keyframes and self.keyframes[-1] == kf and len(self.keyframes) >= 2 and (not self._odo or self._odo[-1][1] != len(self._accepted) - 1):
            base = self.keyframes[-2]
            P1, P0 = np.vstack([self.poses[kf], [0, 0, 0, 1.0]]), np.vstack([self.poses[base], [0, 0, 0, 1.0]])
            odo_all.append((self._accepted.index(base), len(self._accepted) - 1, pose_from_world_to_camera((P1 @ np.linalg.inv(P0))[:3])))
            if write_back or self.ba_check:
                self._odo = odo_all                                   # (the twin keeps its own edge list: measured once, like the device's)
        lm, ps, uv = self.read_log()
        pts = self.objp.astype(np.float64)
        N, P_all = len(pts), len(self._accepted)
        # the poses of the problem: every accepted frame, or the selection (`_select_poses`; csrc/slam_ba.hip is this path's twin)
        sel = self._select_poses()
        sel = np.arange(P_all, dtype=np.int32) if sel is None else sel
        w0, P = int(sel[0]), len(sel)
        pmap = np.full(P_all, -1, np.int64)
        pmap[sel] = np.arange(P)
        has_lm = (lm >= 0) & (lm < N)
        outside = has_lm & (ps >= 0) & (ps < P_all) & (pmap[np.clip(ps, 0, P_all - 1)] < 0)
        out_cnt = np.bincount(lm[outside], minlength=N)             # sightings from accepted frames that are not poses of the problem
        known = has_lm & ~outside                                    # (free tracks that have not become landmarks: lm < 0)
        if self.ba_border_margin:
            # an observation closer to the image border than half a tracker window was measured on a window that reads the
            # border-extended pyramid -- mirrored content (OpenCV's tracker, and this one since round 4, follows a point until its
            # window has left the image altogether): good enough for RANSAC to sort out, not for a least-squares adjustment
            H_, W_ = self.shape
            m = self.ba_border_margin
            known &= (uv[:, 0] >= m) & (uv[:, 0] <= W_ - 1 - m) & (uv[:, 1] >= m) & (uv[:, 1] <= H_ - 1 - m)
        lm, ps, uv = lm[known], pmap[ps[known]], uv[known]
        if len(self._ba_bad) < N:
            self._ba_bad = np.concatenate([self._ba_bad, np.zeros(N - len(self._ba_bad), bool)])
        # the gauge: while frame 0 is a pose of the problem, the reference's (bundle_adjust.cpp:268-282: a prior on pose 0 at its start-up
        # estimate, priors on the start-up landmarks); once it is not, priors on the first selected pose and on the next selected keyframe
        # at their current values (two poses fix the seven gauge freedoms of a monocular map, scale included)
        anchors = [0]
        a2 = self._second_anchor(sel)
        if a2 >= 0 and pmap[a2] > 0:
            anchors.append(int(pmap[a2]))
        accepted_w = [self._accepted[j] for j in sel]
        poses = np.stack([pose_from_world_to_camera(self.poses[f]) for f in accepted_w])
        if w0 == 0:
            poses[0] = pose_from_world_to_camera(self._pose0)         # the pose prior sits at the initial value of pose 0 (bundle_adjust.cpp:273)
        calib = np.array([[self.K[0, 0], self.K[1, 1], self.K[0, 1], self.K[0, 2], self.K[1, 2], self.dist[0], self.dist[1],
                           self.dist[2], self.dist[3]]])
        n0_w = self._n0 if w0 == 0 else 0                            # the start-up landmarks are the gauge only while frame 0 is a pose of the problem
        prior_xyz = pts.copy()
        prior_xyz[:n0_w] = self._objp0[:n0_w]
        prior_w = np.where(np.arange(N) < n0_w, 1.0 / self.ba_point_sigma ** 2, 0.0)             # noise.point3D of the reference's runs
        if self.ba_window_point_sigma and len(sel) < P_all and w0 != 0:
            # what the frames outside the problem know about a landmark stays with it as a prior at its adjusted value: two anchor
            # poses alone leave the scale of a monocular window to drift (measured: 54-97 mm over 200 frames against 5 mm)
            prior_w = np.where((out_cnt[:N] > 0) & (np.arange(N) >= n0_w), 1.0 / self.ba_window_point_sigma ** 2, prior_w)
        per_lm = np.bincount(lm, minlength=N)
        odo = [(int(pmap[a]), int(pmap[b]), m) for a, b, m in odo_all if pmap[a] >= 0 and pmap[b] >= 0]
        t1 = time.perf_counter()
        passes, dropped, hist_all = 0, 0, None
        movable = np.arange(N) >= n0_w
        screened_at_start = not self.ba_gross_pixels
        while True:
            # a landmark joins the adjustment once it has been seen from a THIRD frame (fresh from its triangulation it constrains
            # nothing but the relative pose of its two keyframes), and sits out for good once an adjustment has shown it to be a
            # mistracked corner: GTSAM's factors are plain least squares (bundle_adjust.cpp:289-298: no robust kernel), the
            # reference runs them once over a finished recording -- inside the loop one bad track that passed the depth checks
            # drags the two keyframes it was triangulated from by centimetres before anything else has seen it
            use = (((per_lm >= 1) & (per_lm + out_cnt[:N] >= self.ba_min_observations)) | ~movable) & ~self._ba_bad[:N]
            keep = use[lm]
            l2, p2, u2 = lm[keep], ps[keep], uv[keep]
            order = np.argsort(l2, kind="stable")
            ptr = np.concatenate([[0], np.cumsum(np.bincount(l2, minlength=N))]).astype(np.int64)
            problem = ba_io.SparseProblem(
                poses=poses, pose_cam=np.zeros(P, np.int32), pose_key=[(0, f) for f in accepted_w], calib=calib, sigma=np.array([float(self.ba_pixel_sigma)]),
                points=pts.copy(), obs_ptr=ptr, obs_pose=p2[order].astype(np.int32), obs_uv=u2[order], prior_w=prior_w, prior_xyz=prior_xyz,
                pose_prior_idx=np.array(anchors, np.int32), pose_prior_sigmas=np.tile(np.asarray(self.ba_pose_sigmas, dtype=np.float64), (len(anchors), 1)),
                odo_from=np.array([o[0] for o in odo], np.int32), odo_to=np.array([o[1] for o in odo], np.int32),
                odo_meas=np.array([o[2] for o in odo]).reshape(-1, 12), odo_sigmas=np.tile(np.asarray(self.ba_odometry_sigmas, dtype=np.float64), (len(odo), 1)))
            ba = sparse_ba.SparseBundleAdjuster(problem, device="cuda:%d" % self._device)
            if not screened_at_start:
                # before anything is adjusted: an observation that misses the CURRENT estimate by tens of pixels is not noise the
                # adjustment could average out (the loop's own estimate is good to a pixel or two) -- its landmark sits out at
                # once, before ten LM iterations have spread a 1e7 cost over every pose it touches
                screened_at_start = True
                worst0, zmin = ba.worst_residuals(with_min_depth=True)
                gross = ~(worst0 <= self.ba_gross_pixels) & movable & use
                # ... and a landmark triangulated AT a camera centre (two views a millimetre apart: the reference's filters -- converged,
                # in front of both cameras, slam2.py:556-590 -- let it through) has a depth of ~0 in the cameras that see it: the first
                # step of an adjustment moves it behind one of them, GTSAM's cheirality cost (2 fx per component) makes every trial at
                # every lambda a million times worse than the start, and nothing is ever adjusted again (the ICL-NUIM run, seed 3)
                seen = np.isfinite(zmin) & use
                if seen.any():
                    gross |= (zmin < self.ba_min_depth_ratio * np.median(zmin[seen])) & movable & use
                if gross.any():
                    self._ba_bad[:N] |= gross
                    dropped += int(gross.sum())
                    continue
            # the iterations in legs of `ba_screen_iterations` (0: one leg), the residual screen behind every leg that has neither met
            # the stop rule nor used the iterations up (csrc/slam_ba.hip has the reasons)
            hist, early, it_done, K = None, False, 0, int(self.ba_screen_iterations or 0)
            while True:
                left = self.ba_iterations - it_done
                cap = K if 0 < K < left else left
                leg = ba.optimize(iters=cap, mode="lm")
                done = len(leg) - 1
                it_done += done
                hist = leg if hist is None else hist + leg[1:]
                dec = abs(leg[-2] - leg[-1]) if done >= 1 else 0.0
                ended = done < cap or (done >= 1 and (dec < sparse_ba.LM_ABS_TOL or dec / max(leg[-2], 1e-300) < sparse_ba.LM_REL_TOL))
                if ended or it_done >= self.ba_iterations or K <= 0:
                    break
                if passes + 1 >= self.ba_max_passes:
                    continue
                bad = ~(ba.worst_residuals() <= self.ba_outlier_pixels) & movable & use
                if bad.any():
                    self._ba_bad[:N] |= bad
                    dropped += int(bad.sum())
                    early = True
                    break
            if early:
                hist_all = hist[:1] if hist_all is None else hist_all
                passes += 1
                continue
            if len(hist) == 1 and passes + 1 < self.ba_max_passes:
                # no trial at any lambda was accepted: if that is cheirality -- a first step that puts a landmark behind one of its
                # cameras -- the landmark is found at the trial estimate and sits out; the adjustment is redone without it
                ba.step(-sparse_ba.LM_LAMBDA_INITIAL)
                flipped = np.isinf(ba.worst_residuals(ba.poses_new, ba.points_new)) & movable & use
                if flipped.any():
                    self._ba_bad[:N] |= flipped
                    dropped += int(flipped.sum())
                    passes += 1
                    continue
            hist_all = hist if hist_all is None else hist_all[:1] + hist[1:]
            passes += 1
            # the screen: pixel residuals of the adjusted estimate (`mqs_sba_worst_residual_dev`; `_reprojection_residuals` above is
            # its numpy twin); a landmark with one beyond the bound sits out from now on and the adjustment is redone from the start
            bad = ~(ba.worst_residuals() <= self.ba_outlier_pixels) & movable & use
            if passes >= self.ba_max_passes or not bad.any():
                break
            self._ba_bad[:N] |= bad
            dropped += int(bad.sum())
        new_poses, new_pts = ba.poses.cpu().numpy(), ba.points.cpu().numpy()
        new_pts[~use] = pts[~use]                                    # landmarks that sat out keep their values
        t2 = time.perf_counter()
        adjusted = {}
        for k, f in enumerate(accepted_w):                           # camera-to-world pose12 -> [R | t] world -> camera
            R, c = new_poses[k, :9].reshape(3, 3), new_poses[k, 9:]
            adjusted[f] = np.hstack([R.T, (-R.T @ c)[:, None]])
        if P < P_all - w0 and self.ba_carry:
            # the accepted frames behind the first selected pose that are not poses of the problem keep their pose RELATIVE to the last
            # selected pose in front of them: M_j <- M_j inv(A_before) A_after
            h = lambda M: np.vstack([M, [0, 0, 0, 1.0]])
            a = 0
            for j in range(w0, P_all):
                if pmap[j] >= 0:
                    a = int(pmap[j])
                    continue
                f, fa = self._accepted[j], accepted_w[a]
                adjusted[f] = (h(self.poses[f]) @ np.linalg.inv(h(self.poses[fa])) @ h(adjusted[fa]))[:3]
        report = {"frame": self._accepted[-1], "poses": P, "first_pose_of_the_window": w0, "landmarks": N, "landmarks_adjusted": int(use.sum()),
                                "observations": int(keep.sum()), "passes": passes, "landmarks_screened_out": dropped,
                                "lm_iterations": len(hist_all) - 1, "cost_before": hist_all[0], "cost_after": hist_all[-1], "accepted_frames": P_all,
                                "build_ms": round(1e3 * (t1 - t0), 3), "adjust_ms": round(1e3 * (t2 - t1), 3), "engine": "host"}
        if not write_back:
            retired = self._ba_bad.copy()                             # (the twin's retired set lives on, as the device's does)
            every = np.stack([adjusted.get(f, self.poses[f]) for f in self._accepted])
            return every, np.asarray(new_pts, dtype=np.float32).astype(np.float64), report, retired
        for f, M in adjusted.items():
            self.poses[f] = M
        # the live state's two poses: the last accepted frame's, and the base keyframe's of the live tracks (behind a keyframe the
        # same frame; in finish() behind plain frames it is the last KEYFRAME's -- the tracks' base points belong to that frame)
        last = np.ascontiguousarray(self.poses[self._accepted[-1]], dtype=np.float64)
        key = np.ascontiguousarray(self.poses[self.keyframes[-1]], dtype=np.float64)
        _lib.check(_lib.lib().mqs_slam_write_back(self._h, np.ascontiguousarray(new_pts).ctypes.data_as(_lib.c_f64p), N,
                                                  last.ctypes.data_as(_lib.c_f64p), key.ctypes.data_as(_lib.c_f64p)))
        report["write_back_ms"] = round(1e3 * (time.perf_counter() - t2), 3)
        self.ba_reports.append(report)

    def _reassociate(self, img):
        n = ctypes.c_int32(0)
        _lib.check(_lib.lib().mqs_slam_reassociate(self._h, ctypes.c_float(REASSOCIATE_RADIUS), ctypes.c_double(REASSOCIATE_RATIO),
                                                   ctypes.byref(n)))
        self.reassociated += int(n.value)

    def finish(self):
        """Waits for the last keyframe branch and takes its report; with bundle_adjust, one last adjustment over everything
        (the frames behind the last keyframe included)."""
        _lib.check(_lib.lib().mqs_slam_flush(self._h, self._pres))
        self._take_keyframe_report()
        if self.bundle_adjust and self.keyframes and self._accepted and self._accepted[-1] != self.keyframes[-1]:
            self._bundle_adjust()

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
