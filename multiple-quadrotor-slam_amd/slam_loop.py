"""
The reference's monocular per-frame loop (Work/SLAM/application/own/slam2.py:360-695 `handle_new_frame`, :1021-1253
`main`) on the gfx950 path end to end -- BASELINE configs[4]: detect -> track -> pose -> triangulate per keyframe.

    frame      calcOpticalFlowPyrLK from the previous frame (:381), drop status == 0 / error >= max_OF_error (:382),
               reject the frame when too many tracks are lost (:385-387) or < 8 triangulated points remain (:437);
               solvePnPRansac on the already-triangulated tracks (:453-454), outlier-ratio gate (:461-468), solvePnP on
               the inliers from the RANSAC pose (:489-490), reprojection-error gate (:493-497)
    keyframe   (keyframe_test :43-59: homography between the base keyframe's and the current undistorted points is
               far enough from a pure scaling, singular-value ratio w0 / w2 > 1.04)
               undistort + iterative-LS triangulation of the not-yet-triangulated tracks against the base keyframe
               (:551-555), keep status == 1 (:556), refine the pose on old + new points (:576-577), re-triangulate
               (:582-584), keep status >= 0 (:589); then top the tracks up with goodFeaturesToTrack under the coverage
               mask of the current points (:657-672) and make this frame the new base keyframe.

Bookkeeping is restated with plain arrays (track id -> landmark id or -1) instead of the reference's index sets; the
decisions and their thresholds are the reference's (:1070-1098).  Deviations, because no image set / OpenCV run of the
reference exists to compare with: the homography of the keyframe test is a normalised DLT (the estimator behind
cv2.findHomography(method=0)) over ALL tracked points by default -- the reference's random quarter of them (:48,
1088-1089) is `max_homography_points="reference"`, drawn with the reference's call from a seeded legacy generator; it
makes the outcome depend on the draw (profiles/r03/12_keyframe_sample_study.json) -- and RANSAC draws come from a
seeded numpy generator.  Every numeric step runs on the GPU library (features, pnp, camera, triangulation); nothing falls back.
"""
import time

import numpy as np

from . import camera
from . import features
from . import pnp
from . import triangulation

# slam2.py:1070-1098
MAX_OF_ERROR = 12.0
MAX_LOST_TRACKS_RATIO = 0.5
KEYPOINT_COVERAGE_RADIUS = int(MAX_OF_ERROR)
MAX_AMOUNT_KEYPOINTS = 300
CORNER_QUALITY_LEVEL = 0.01
HOMOGRAPHY_CONDITION_THRESHOLD = 1.04
MAX_SOLVEPNP_REPROJ_ERROR = 2.0
MAX_SOLVEPNP_OUTLIER_RATIO = 0.33


def keypoint_mask(shape, points, radius=KEYPOINT_COVERAGE_RADIUS):
    """slam2.py:29-40: ones with a filled disc of zeros around every point.  The centre is truncated, as the Python 2 binding of
    cv2.circle does with the float32 pair it is handed (a Point is parsed with "ii": a float goes through __int__); OpenCV's
    filled circle of radius 12 (drawing.cpp: the midpoint algorithm) covers exactly the pixels with dx^2 + dy^2 <= 144."""
    H, W = shape
    mask = np.ones((H, W), dtype=np.uint8)
    p = np.trunc(np.asarray(points, dtype=np.float64).reshape(-1, 2)).astype(np.int64)
    if len(p) == 0:
        return mask
    r = int(radius)
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    dy, dx = yy[(xx * xx + yy * yy) <= r * r], xx[(xx * xx + yy * yy) <= r * r]
    X = (p[:, 0:1] + dx[None, :]).ravel()
    Y = (p[:, 1:2] + dy[None, :]).ravel()
    ok = (X >= 0) & (X < W) & (Y >= 0) & (Y < H)
    mask[Y[ok], X[ok]] = 0
    return mask


def homography_dlt(p1, p2):
    """Normalised DLT (least squares over all pairs): the first half of cv2.findHomography(method=0) (`find_homography`)."""
    def norm(p):
        c = p.mean(axis=0)
        s = np.sqrt(2.0) / max(np.mean(np.linalg.norm(p - c, axis=1)), 1e-12)
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T
    a, Ta = norm(np.asarray(p1, dtype=np.float64))
    b, Tb = norm(np.asarray(p2, dtype=np.float64))
    n = len(a)
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2], A[0::2, 2] = a, 1.0
    A[0::2, 6:8], A[0::2, 8] = -b[:, 0:1] * a, -b[:, 0]
    A[1::2, 3:5], A[1::2, 5] = a, 1.0
    A[1::2, 6:8], A[1::2, 8] = -b[:, 1:2] * a, -b[:, 1]
    Hn = np.linalg.eigh(A.T @ A)[1][:, 0].reshape(3, 3)      # null vector of the 2n x 9 system
    H = np.linalg.inv(Tb) @ Hn @ Ta
    return H / H[2, 2]


HOMOGRAPHY_REFINE_ITERATIONS = 10        # fundam.cpp (2.4): estimator.refine(M, m, &matH, 10)
HOMOGRAPHY_REFINE_STOP = 1e-12           # relative step below which the remaining iterations change nothing that matters


def homography_refine(H, p1, p2, max_iter=HOMOGRAPHY_REFINE_ITERATIONS):
    """The second half of cv2.findHomography(method=0) (OpenCV 2.4 fundam.cpp: `estimator.refine(M, m, &matH, 10)` whenever there
    are more than four pairs): Levenberg-Marquardt on the eight free entries of H (h33 = 1) over the transfer error
    sum |p2 - proj(H p1)|^2, with CvLevMarq's schedule -- lambda 1e-3, (J^T J) with its diagonal scaled by (1 + lambda), a step
    that raises the error is retried with lambda x 10, an accepted one divides lambda by 10, at most `max_iter` accepted steps.
    OpenCV stops early only when the relative step falls below DBL_EPSILON (in practice never); here at 1e-12: the steps left
    over move H by less than that.  With parallax between the two views the least-squares homography is a compromise, and the
    algebraic (DLT) and geometric minimisers differ in the third digit of w0 / w2 -- the digit keyframe_test's 1.04 looks at."""
    p1, p2 = np.asarray(p1, dtype=np.float64), np.asarray(p2, dtype=np.float64)
    if len(p1) <= 4:
        return H / H[2, 2]
    h = (H / H[2, 2]).ravel()[:8].copy()

    def transfer(h):
        Hm = np.append(h, 1.0).reshape(3, 3)
        q = np.c_[p1, np.ones(len(p1))] @ Hm.T
        return q, (q[:, :2] / q[:, 2:3] - p2)

    q, r = transfer(h)
    err = float(np.sum(r * r))
    lam_lg10 = -3
    for _ in range(max_iter):
        # J^T J and J^T r from their 29 distinct sums (rows of J: [a 0 c] for x, [0 a d] for y with a = (p1, 1) / z, c = -a_xy x, d = -a_xy y)
        w = 1.0 / q[:, 2]
        x, y = q[:, 0] * w, q[:, 1] * w
        a = np.c_[p1 * w[:, None], w]
        c, dd = -a[:, :2] * x[:, None], -a[:, :2] * y[:, None]
        A = np.zeros((8, 8))
        A[0:3, 0:3] = A[3:6, 3:6] = a.T @ a
        A[0:3, 6:8], A[3:6, 6:8] = a.T @ c, a.T @ dd
        A[6:8, 0:3], A[6:8, 3:6] = A[0:3, 6:8].T, A[3:6, 6:8].T
        A[6:8, 6:8] = c.T @ c + dd.T @ dd
        g = np.concatenate([a.T @ r[:, 0], a.T @ r[:, 1], c.T @ r[:, 0] + dd.T @ r[:, 1]])
        accepted = False
        while lam_lg10 <= 16:
            An = A.copy()
            An[np.diag_indices(8)] *= 1.0 + 10.0 ** lam_lg10
            try:
                step = np.linalg.solve(An, g)
            except np.linalg.LinAlgError:
                lam_lg10 += 1
                continue
            q2, r2 = transfer(h - step)
            err2 = float(np.sum(r2 * r2))
            if err2 > err:
                lam_lg10 += 1
                continue
            accepted = True
            break
        if not accepted:
            break
        rel = float(np.linalg.norm(step) / max(np.linalg.norm(h), 1e-300))
        h, q, r, err = h - step, q2, r2, err2
        lam_lg10 = max(lam_lg10 - 1, -16)
        if rel < HOMOGRAPHY_REFINE_STOP:
            break
    return np.append(h, 1.0).reshape(3, 3)


def find_homography(p1, p2):
    """cv2.findHomography(p1, p2) with method = 0 (slam2.py:54): normalised DLT, then the LM refinement of the transfer error."""
    return homography_refine(homography_dlt(p1, p2), p1, p2)


def keyframe_test(points1, points2, K, dist, max_points=None, rng=None):
    """slam2.py:43-59.  max_points / rng: the reference's random sample of the input points (:48,
    `np.random.permutation(len(points1))[:max_num_homography_points]`); rng is a `numpy.random.RandomState` -- the legacy
    generator behind `np.random.permutation`, seeded here where the reference leaves the global one unseeded."""
    if len(points1) < 4:
        return False
    points1 = np.asarray(points1, dtype=np.float64)
    points2 = np.asarray(points2, dtype=np.float64)
    if max_points is not None and max_points > 0:
        idxs = (rng if rng is not None else np.random).permutation(len(points1))[:max_points]
        points1, points2 = points1[idxs], points2[idxs]
    u1 = camera.undistort_points(points1, K, dist)
    u2 = camera.undistort_points(points2, K, dist)
    w = np.linalg.svd(find_homography(u1, u2), compute_uv=False)
    return w[0] / w[2] > HOMOGRAPHY_CONDITION_THRESHOLD


class MonoSlam:
    def __init__(self, cameraMatrix, distCoeffs, image_shape, seed=0, verbose=False, ba_info=None, max_homography_points=0,
                 second_pass_screen=None):
        """max_homography_points: keyframe_test's random sample of the tracks (slam2.py:48): 0 = all tracks (default; see
        slam_device.DeviceMonoSlam for the measurement behind it), "reference" = max(4, target_amount_keypoints / 4) (:1088-1089).
        ba_info: an optional `ba_io.BundleAdjustmentInfoContainer`; the loop then records what the reference records
        for the bundle adjuster (slam2.py:519-522, 634-641, 681-687, 1167-1169, 1204)."""
        self.K = np.asarray(cameraMatrix, dtype=np.float64)
        self.dist = np.asarray(distCoeffs, dtype=np.float64).reshape(-1)[:4]
        self.shape = tuple(image_shape)
        H, W = self.shape
        target = int(round(W * H / (np.pi * KEYPOINT_COVERAGE_RADIUS ** 2)))         # slam2.py:1081
        self.target_keypoints = min(MAX_AMOUNT_KEYPOINTS, target)
        self.rng = np.random.default_rng(seed)
        # keyframe_test's random sample (slam2.py:1088-1089: target_amount_keypoints / 4, at least 4), drawn like the reference
        # draws it (np.random.permutation) from a seeded legacy generator
        self.second_pass_screen = float(second_pass_screen) if second_pass_screen else 0.0
        self.max_homography_points = (max(4, self.target_keypoints // 4) if max_homography_points == "reference"
                                      else int(max_homography_points))
        self.legacy_rng = np.random.RandomState(seed)
        self.verbose = verbose
        self.objp = np.zeros((0, 3), dtype=np.float32)       # the map (float32 like slam2.py:19)
        self.poses = []                                      # per frame: (rvec, tvec) or None when rejected
        self.keyframes = []
        self.timing = []
        self.ba_info = ba_info
        self.history = []                                    # since the base keyframe: (frame, track ids, image points)
        if ba_info is not None:
            ba_info.set_calibration(self.K, self.dist)

    def _log(self, *a):
        if self.verbose:
            print(*a)

    def _P(self, rvec, tvec):
        return np.hstack([pnp.Rodrigues(rvec), np.asarray(tvec, dtype=np.float64).reshape(3, 1)])

    def _top_up(self, img, pts):
        to_add = max(0, self.target_keypoints - len(pts))
        return features.goodFeaturesToTrack(img, to_add, CORNER_QUALITY_LEVEL, KEYPOINT_COVERAGE_RADIUS, None,
                                            keypoint_mask(self.shape, pts))

    def start(self, img, init_objp, init_imgp):
        """slam2.py:1136-1180: pose of the first frame from known 3-D points, then the first batch of free tracks."""
        self.objp = np.asarray(init_objp, dtype=np.float32).reshape(-1, 3)
        imgp = np.asarray(init_imgp, dtype=np.float32).reshape(-1, 2)
        ret, rvec, tvec = pnp.solvePnP(self.objp, imgp, self.K, self.dist)
        self.poses.append((rvec, tvec))
        self.keyframes.append(0)
        self.rvec_keyfr, self.tvec_keyfr = rvec, tvec
        extra = self._top_up(img, imgp)
        self.pts = np.concatenate([imgp, extra])             # current image points of the live tracks
        self.base_pts = self.pts.copy()                      # their image points in the base keyframe
        self.lm = np.concatenate([np.arange(len(imgp)), -np.ones(len(extra), dtype=np.int64)])   # track -> landmark or -1
        self.tid = np.arange(len(self.pts), dtype=np.int64)  # track ids (stable while a track lives)
        self.next_tid = len(self.pts)
        self.prev_img = img
        if self.ba_info is not None:                         # slam2.py:1167-1169, 1184-1185
            self.ba_info.set_point3DAddedIdxs(np.arange(len(imgp)))
            self.ba_info.add_points2D_3Dassoc(imgp, np.arange(len(imgp)), 0)
            self.history = [(0, self.tid.copy(), self.pts.copy())]
        return rvec, tvec

    def handle_new_frame(self, img):
        """Returns 0 (rejected), 1 (frame) or 2 (keyframe), like the reference's `ret`."""
        t0 = time.perf_counter()
        r = self._frame(img)
        self.timing.append(time.perf_counter() - t0)
        return r

    def _frame(self, img):
        K, dist = self.K, self.dist
        frame_idx = len(self.poses)
        if self.ba_info is not None:
            self.ba_info.next_step()                         # slam2.py:1204: one step per frame, rejected ones included
        new_pts, st, err = features.calcOpticalFlowPyrLK(self.prev_img, img, self.pts)
        keep = (st.ravel() == 1) & (err.ravel() < MAX_OF_ERROR)
        lost = 1.0 - keep.mean() if len(keep) else 1.0
        if lost > MAX_LOST_TRACKS_RATIO:
            self._log("REJECTED: lost track of too many points", lost)
            self.poses.append(None)
            return 0
        pts, base, lm, tid = new_pts[keep], self.base_pts[keep], self.lm[keep], self.tid[keep]
        tri = lm >= 0
        if tri.sum() < 8:
            self._log("REJECTED: fewer than 8 triangulated tracks")
            self.poses.append(None)
            return 0
        objp_t, imgp_t = self.objp[lm[tri]], pts[tri]
        rvec_, tvec_, inliers = pnp.solvePnPRansac(objp_t, imgp_t, K, dist,
                                                   minInliersCount=int(np.ceil((1 - MAX_SOLVEPNP_OUTLIER_RATIO) * tri.sum())),
                                                   reprojectionError=MAX_SOLVEPNP_REPROJ_ERROR, seed=int(self.rng.integers(1 << 30)))
        if inliers is None:
            self.poses.append(None)
            return 0
        inliers = inliers.ravel()
        outlier_ratio = (tri.sum() - len(inliers)) / float(tri.sum())
        if outlier_ratio > MAX_SOLVEPNP_OUTLIER_RATIO or len(inliers) < 8:
            self._log("REJECTED: PnP outlier ratio", outlier_ratio)
            self.poses.append(None)
            return 0
        objp_i, imgp_i = objp_t[inliers], imgp_t[inliers]
        ret, rvec, tvec = pnp.solvePnP(objp_i, imgp_i, K, dist, rvec_, tvec_, useExtrinsicGuess=True)
        reproj, _ = camera.reprojection_error(objp_i.astype(np.float64), imgp_i.astype(np.float64), K, dist, rvec, tvec)
        if reproj > MAX_SOLVEPNP_REPROJ_ERROR:
            self._log("REJECTED: reprojection error", reproj)
            self.poses.append(None)
            return 0
        # keep the inlier tracks and the not-yet-triangulated ones
        tri_idx = np.nonzero(tri)[0]
        sel = np.zeros(len(pts), dtype=bool)
        sel[tri_idx[inliers]] = True
        sel |= ~tri
        pts, base, lm, tid = pts[sel], base[sel], lm[sel], tid[sel]
        tri = lm >= 0
        if self.ba_info is not None:                         # slam2.py:519-522
            self.history.append((frame_idx, tid.copy(), pts.copy()))
            self.ba_info.add_points2D_3Dassoc(pts[tri], lm[tri], frame_idx)
        result = 1
        if keyframe_test(base, pts, K, dist, self.max_homography_points, self.legacy_rng):
            result = 2
            non = np.nonzero(~tri)[0]
            if len(non):
                imgp0, imgp1 = base[non].astype(np.float64), pts[non].astype(np.float64)
                u0, u1 = camera.undistort_points(imgp0, K, dist), camera.undistort_points(imgp1, K, dist)
                P0 = self._P(self.rvec_keyfr, self.tvec_keyfr)
                x1, s1 = triangulation.iterative_LS_triangulation(u0, P0, u1, self._P(rvec, tvec))
                ok = np.nonzero(np.asarray(s1) == 1)[0]
                if self.second_pass_screen and len(ok):
                    # optional, not in slam2.py's flow (its :1092 max_2nd_solvePnP_reproj_error is defined for this place and unused):
                    # fresh points that miss their own measurement in this frame by more than the bound stay out of the second solvePnP
                    uvp, _, _ = camera.project_points(np.asarray(x1)[ok].astype(np.float32).astype(np.float64), K, dist, self._P(rvec, tvec), imgp1[ok])
                    ok = ok[np.linalg.norm(uvp - imgp1[ok], axis=1) <= self.second_pass_screen]
                if len(ok):
                    obj_all = np.concatenate([objp_i, np.asarray(x1)[ok].astype(np.float32)])
                    img_all = np.concatenate([imgp_i, imgp1[ok].astype(np.float32)])
                    ret, rvec, tvec = pnp.solvePnP(obj_all, img_all, K, dist, rvec, tvec, useExtrinsicGuess=True)
                    x2, s2 = triangulation.iterative_LS_triangulation(u0[ok], P0, u1[ok], self._P(rvec, tvec))
                    good = np.nonzero(np.asarray(s2) >= 0)[0]
                    ids = len(self.objp) + np.arange(len(good))
                    self.objp = np.concatenate([self.objp, np.asarray(x2)[good].astype(np.float32)])
                    lm[non[ok[good]]] = ids
                    if self.ba_info is not None and len(good):
                        # slam2.py:634-641: the new landmarks and their image points in every frame since the base keyframe
                        new_tid = tid[non[ok[good]]]
                        self.ba_info.set_point3DAddedIdxs(ids)
                        for ev_frame, ev_tid, ev_pts in self.history:
                            pos = {t: k for k, t in enumerate(ev_tid)}
                            sel_ev = np.array([pos[t] for t in new_tid], dtype=np.int64)
                            self.ba_info.add_points2D_3Dassoc(ev_pts[sel_ev], ids, ev_frame)
                # tracks that failed to triangulate are dropped (slam2.py:596-612)
                done = lm >= 0
                pts, base, lm, tid = pts[done], base[done], lm[done], tid[done]
            extra = self._top_up(img, pts)
            pts = np.concatenate([pts, extra])
            lm = np.concatenate([lm, -np.ones(len(extra), dtype=np.int64)])
            tid = np.concatenate([tid, self.next_tid + np.arange(len(extra), dtype=np.int64)])
            self.next_tid += len(extra)
            base = pts.copy()                                 # rebase on this keyframe (slam2.py:673-674)
            if self.ba_info is not None:                      # slam2.py:681-687: odometry base keyframe -> this frame
                P1 = np.vstack([self._P(rvec, tvec), [0, 0, 0, 1.0]])
                P0 = np.vstack([self._P(self.rvec_keyfr, self.tvec_keyfr), [0, 0, 0, 1.0]])
                self.ba_info.add_odometry(P1 @ np.linalg.inv(P0), self.history[0][0], frame_idx)     # trfm.delta_P(P1, P0)
                self.history = [(frame_idx, tid.copy(), pts.copy())]
            self.rvec_keyfr, self.tvec_keyfr = rvec, tvec
            self.keyframes.append(len(self.poses))
        self.pts, self.base_pts, self.lm, self.tid = pts, base, lm, tid
        self.prev_img = img
        self.poses.append((rvec, tvec))
        return result

    def projection_matrices(self):
        """Per frame the 3x4 world->camera matrix, None for rejected frames (the `Ps` of slam2.py:703-708)."""
        return [None if p is None else self._P(*p) for p in self.poses]

    def trajectory(self):
        """Camera centres (F, 3), NaN for rejected frames."""
        out = np.full((len(self.poses), 3), np.nan)
        for i, p in enumerate(self.poses):
            if p is not None:
                R = pnp.Rodrigues(p[0])
                out[i] = (-R.T @ np.asarray(p[1]).reshape(3)).ravel()
        return out
