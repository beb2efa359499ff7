"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's
brute-force matcher:  cv2.BFMatcher() (NORM_L2, crossCheck=False) as wrapped by
/root/reference/Work/python_libs/cv2_helpers.py:278-345 (radiusMatch with k = 2 built on
cv2.batchDistance, :296-339) and consumed by Work/SLAM/application/own/slam.py:101-125.

OpenCV 2.4 (not vendored; absent here) is NOT available: parity of this path is UNPINNED.  The
published behaviour restated: squared L2 distance accumulated in float32 in dimension order
(the scalar tail loop of normL2Sqr_, which is the whole loop for D < 4 -- the reference's D = 2),
sqrt, ascending order, ties towards the lower train index (batchDistance inserts on strict `<`).
"""
from collections import namedtuple
import numpy as np

DMatch = namedtuple("DMatch", ["queryIdx", "trainIdx", "distance"])


def sqdist_f32(query, train):
    """(Nq, Nt) float32 squared distances, sequential float32 accumulation, no FMA."""
    q = np.asarray(query, dtype=np.float32)
    t = np.asarray(train, dtype=np.float32)
    acc = np.zeros((q.shape[0], t.shape[0]), dtype=np.float32)
    for k in range(q.shape[1]):
        d = (q[:, k, None] - t[None, :, k]).astype(np.float32)
        acc = (acc + (d * d).astype(np.float32)).astype(np.float32)
    return acc


def knn2(query, train, block=2048):
    """idx (Nq,2) int32 (-1 where Nt < k), dist (Nq,2) float32 (+inf where idx == -1)."""
    q = np.asarray(query, dtype=np.float32)
    t = np.asarray(train, dtype=np.float32)
    Nq, Nt = q.shape[0], t.shape[0]
    idx = np.full((Nq, 2), -1, dtype=np.int32)
    dist = np.full((Nq, 2), np.inf, dtype=np.float32)
    for a in range(0, Nq, block):
        d2 = sqdist_f32(q[a:a + block], t)
        if Nt == 0:
            continue
        # stable argsort == lowest index first among ties
        order = np.argsort(d2, axis=1, kind="stable")[:, :2]
        rows = np.arange(d2.shape[0])
        for k in range(min(2, Nt)):
            idx[a:a + block, k] = order[:, k]
            dist[a:a + block, k] = np.sqrt(d2[rows, order[:, k]])
    return idx, dist


def knn2_hamming_bits(qbits, tbits, block=1024):
    """Binary descriptors as {0,1} arrays (N, D): exact integer squared distance = Hamming."""
    q = np.asarray(qbits, dtype=np.float32)
    t = np.asarray(tbits, dtype=np.float32)
    Nq, Nt = q.shape[0], t.shape[0]
    idx = np.full((Nq, 2), -1, dtype=np.int32)
    dist = np.full((Nq, 2), np.inf, dtype=np.float32)
    tn = (t * t).sum(1)
    for a in range(0, Nq, block):
        qa = q[a:a + block]
        d2 = (qa * qa).sum(1)[:, None] + tn[None, :] - 2.0 * (qa @ t.T)      # exact small integers
        order = np.argsort(d2, axis=1, kind="stable")[:, :2]
        rows = np.arange(d2.shape[0])
        for k in range(min(2, Nt)):
            idx[a:a + block, k] = order[:, k]
            dist[a:a + block, k] = np.sqrt(d2[rows, order[:, k]].astype(np.float32))
    return idx, dist


def radius_match(query, train, max_radius):
    """cv2_helpers.py:296-339: per query the <= 2 nearest train points with dist <= max_radius."""
    idx, dist = knn2(query, train)
    out = []
    for qi in range(len(idx)):
        ms = []
        for k in range(2):
            if idx[qi, k] >= 0 and dist[qi, k] <= max_radius:
                ms.append(DMatch(qi, int(idx[qi, k]), float(dist[qi, k])))
        out.append(ms)
    return out


def ratio_test_and_dedupe(matches_twoNN, err, max_dist_ratio=0.7):
    """slam.py:108-125: Lowe ratio (or singleton), then one match per trainIdx preferring lower err."""
    best = {}
    for ms in matches_twoNN:
        if not (len(ms) == 1 or (len(ms) > 1 and ms[0].distance / ms[1].distance < max_dist_ratio)):
            continue
        m = ms[0]
        if m.trainIdx not in best or err[m.queryIdx] < err[best[m.trainIdx].queryIdx]:
            best[m.trainIdx] = m
    return best


def match_OF_based(right_OF_points, right_FAST_points, err_OF, status_OF, max_radius_OF_to_FAST, max_dist_ratio,
                   left_point_idxs=None, max_OF_error=12.0):
    """slam.py:81-127, statement by statement (the filter :84-90, radiusMatch :101-104, the loop :106-125)."""
    kept = [(p, i) for i, p in enumerate(right_OF_points)
            if status_OF[i] and err_OF[i] < max_OF_error and (left_point_idxs is None or i in left_point_idxs)]
    if not kept:
        return {}
    pts, to_left = zip(*kept)
    matches_twoNN = radius_match(np.array(pts), right_FAST_points, max_radius_OF_to_FAST)
    best = {}
    for query_matches in matches_twoNN:
        if len(query_matches) > 1 and query_matches[1].distance == 0.0:
            continue                                          # the reference divides by zero here (ZeroDivisionError)
        if not (len(query_matches) == 1 or
                (len(query_matches) > 1 and query_matches[0].distance / query_matches[1].distance < max_dist_ratio)):
            continue
        match = DMatch(to_left[query_matches[0].queryIdx], query_matches[0].trainIdx, query_matches[0].distance)
        if match.trainIdx not in best or err_OF[match.queryIdx] < err_OF[best[match.trainIdx].queryIdx]:
            best[match.trainIdx] = match
    return best
