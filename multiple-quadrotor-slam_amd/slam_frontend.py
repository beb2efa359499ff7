"""
The detect / track / match front-end of the reference's first SLAM prototype on the gfx950 kernels -- counterpart of
`main_loop` in Work/SLAM/application/own/slam.py:57-226 (the "OpenCV feature-detect / BF-match front-end" of the hot path):

    right key points    cv2.FastFeatureDetector().detect(right_img)              slam.py:61-63  -> features.FastFeatureDetector
    optical flow        cv2.calcOpticalFlowPyrLK(left_gray, right_gray, left_points)   :75-79   -> features.calcOpticalFlowPyrLK
    match_OF_based      keep flow points with status, err < max_OF_error (and in the mask) :84-90;
                        matcher.radiusMatch(flow points, FAST points, radius) :101-104; ratio test, relink to the left
                        indices, one match per FAST point preferring the lower flow error :106-127
                                                                                  -> matching.match_radius_ratio_unique (one call)
    bookkeeping         mean flow of the kept matches :160-165, partition into already-triangulated / new :167-177

Not covered: the chessboard branch's cv2.cornerSubPix refinement (:140-156, initialisation aid) -- matching with the
chessboard radius / ratio through `left_point_idxs` works, the sub-pixel refinement is not applied.
Every numeric step runs on the GPU library; nothing falls back.
"""
import numpy as np

from . import features
from . import matching

# slam.py:19-31
MAX_OF_ERROR = 12.0
MAX_RADIUS_OF_TO_FAST = {"chessboard": 4.0, "FAST": 2.0}
MAX_DIST_RATIO = {"chessboard": 1.0, "FAST": 0.7}


def match_OF_based(right_OF_points, right_FAST_points, err_OF, status_OF, max_radius_OF_to_FAST, max_dist_ratio,
                   left_point_idxs=None, max_OF_error=MAX_OF_ERROR):
    """slam.py:81-127.  Returns {trainIdx: DMatch(queryIdx = index into the LEFT points, trainIdx, distance)}."""
    pts = np.asarray(right_OF_points, dtype=np.float32).reshape(-1, 2)
    err = np.asarray(err_OF, dtype=np.float32).reshape(-1)
    keep = (np.asarray(status_OF).reshape(-1) != 0) & (err < max_OF_error)
    if left_point_idxs is not None:
        m = np.zeros(len(pts), dtype=bool)
        m[np.fromiter(left_point_idxs, dtype=np.int64)] = True
        keep &= m
    to_left = np.nonzero(keep)[0]
    if len(to_left) == 0:
        return {}
    best = matching.match_radius_ratio_unique(pts[to_left], np.asarray(right_FAST_points, dtype=np.float32).reshape(-1, 2),
                                              max_radius_OF_to_FAST, max_dist_ratio, priority=err[to_left])
    return {t: matching.DMatch(int(to_left[m.queryIdx]), m.trainIdx, m.distance) for t, m in best.items()}


def main_loop(left_points, left_gray, right_gray, triangl_idxs, chessboard_idxs=None, fast=None):
    """
    One frame of slam.py's main_loop (:57-226): returns
      right_FAST_points (n, 2) float32, matches_by_trainIdx, (matches of already triangulated left points, matches of new
      ones), mean_OF_vector, new triangl_idxs (= the matched FAST points, :222), new chessboard_idxs.
    """
    fast = fast if fast is not None else features.FastFeatureDetector()
    right_FAST_points, _ = fast.detect_arrays(right_gray)
    left_points = np.asarray(left_points, dtype=np.float32).reshape(-1, 2)
    right_OF_points, status_OF, err_OF = features.calcOpticalFlowPyrLK(left_gray, right_gray, left_points)
    err_OF = err_OF.reshape(-1)
    matches = match_OF_based(right_OF_points, right_FAST_points, err_OF, status_OF,
                             MAX_RADIUS_OF_TO_FAST["FAST"], MAX_DIST_RATIO["FAST"])
    if chessboard_idxs:
        cb = match_OF_based(right_OF_points, right_FAST_points, err_OF, status_OF,
                            MAX_RADIUS_OF_TO_FAST["chessboard"], MAX_DIST_RATIO["chessboard"], chessboard_idxs)
        matches.update(cb)                                    # chessboard matches overwrite FAST matches (:138)
        chessboard_idxs = set(cb)
    train = list(matches)
    query = [matches[t].queryIdx for t in train]
    mean_OF_vector = (right_FAST_points[train] - left_points[query]).mean(axis=0) if train else np.zeros(2, np.float32)
    tri = [matches[t] for t in matches if matches[t].queryIdx in triangl_idxs]
    non = [matches[t] for t in matches if matches[t].queryIdx not in triangl_idxs]
    return right_FAST_points, matches, (tri, non), mean_OF_vector, set(matches), chessboard_idxs
