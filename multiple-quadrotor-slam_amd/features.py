"""
Image front-end of the per-frame loop on the gfx950 kernels of csrc/features.hip, with OpenCV's call surface:

  goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance[, corners, mask])   -> (n, 2) float32
      = Work/python_libs/cv2_helpers.py:34-37 (cv2.goodFeaturesToTrack reshaped to (-1, 2); to_add == 0 -> empty),
        called at Work/SLAM/application/own/slam2.py:665 and :1174
  calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, ...)   -> (nextPts, status, err)
      = cv2.calcOpticalFlowPyrLK as called at slam2.py:381 (defaults 21 x 21, maxLevel 3, 30 iterations / eps 0.01)

  FastFeatureDetector(threshold=10, nonmaxSuppression=True).detect(img)   -> list of KeyPoint(pt, size, response)
      = cv2.FastFeatureDetector() as used at Work/SLAM/application/own/slam.py:34, 62

Images: 2-D uint8 arrays (FastFeatureDetector also takes H x W x 3 BGR, like the reference passes it).  No CPU fallback: the library must be loadable.  Parity with OpenCV itself is unpinned
(oracle/features_np.py restates its published method; the reference holds no images).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import c_f32p, c_i32p, c_u8p


def _image(img, name):
    a = np.asarray(img)
    if a.ndim != 2 or a.dtype != np.uint8:
        raise ValueError("%s must be a 2-D uint8 image" % name)
    if a.shape[0] < 3 or a.shape[1] < 3:
        raise ValueError("%s must be at least 3 x 3" % name)
    return np.ascontiguousarray(a)


def goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance, corners=None, mask=None, capacity=None):
    """`corners` is OpenCV's unused output placeholder (slam2.py passes None)."""
    del corners
    maxCorners = int(maxCorners)
    if maxCorners == 0 and capacity is None:
        # cv2_helpers.goodFeaturesToTrack: to_add == 0 means "add nothing" (works around OpenCV returning everything)
        return np.zeros((0, 2), dtype=np.float32)
    im = _image(img, "img")
    H, W = im.shape
    m = None
    if mask is not None:
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if m.shape != im.shape:
            raise ValueError("mask must have the image's shape")
    cap = int(capacity) if capacity is not None else max(1, maxCorners)
    out = np.zeros((cap, 2), dtype=np.float32)
    n = np.zeros(1, dtype=np.int32)
    _lib.check(_lib.lib().mqs_good_features_to_track(
        _lib.default_context().handle, im.ctypes.data_as(c_u8p), W, H, maxCorners, ctypes.c_double(qualityLevel),
        ctypes.c_double(minDistance), None if m is None else m.ctypes.data_as(c_u8p), out.ctypes.data_as(c_f32p), cap,
        n.ctypes.data_as(c_i32p)))
    return out[:int(n[0])].copy()


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3,
                         criteria=(3, 30, 0.01), flags=0, minEigThreshold=1e-4):
    """criteria = (type, max iterations, epsilon) like cv2's TermCriteria tuple.  Returns nextPts shaped like prevPts
    (float32), status (n, 1) uint8, err (n, 1) float32."""
    if flags != 0 or nextPts is not None:
        raise NotImplementedError("OPTFLOW_USE_INITIAL_FLOW / OPTFLOW_LK_GET_MIN_EIGENVALS are not supported")
    a, b = _image(prevImg, "prevImg"), _image(nextImg, "nextImg")
    if a.shape != b.shape:
        raise ValueError("prevImg and nextImg must have the same size")
    p = np.asarray(prevPts)
    shape = p.shape
    pts = np.ascontiguousarray(p.reshape(-1, 2), dtype=np.float32)
    n = len(pts)
    H, W = a.shape
    out = np.zeros((n, 2), dtype=np.float32)
    status = np.zeros(n, dtype=np.uint8)
    err = np.zeros(n, dtype=np.float32)
    if n:
        _lib.check(_lib.lib().mqs_calc_optical_flow_pyr_lk(
            _lib.default_context().handle, a.ctypes.data_as(c_u8p), b.ctypes.data_as(c_u8p), W, H, pts.ctypes.data_as(c_f32p),
            n, int(winSize[0]), int(winSize[1]), int(maxLevel), int(criteria[1]), ctypes.c_double(criteria[2]),
            ctypes.c_double(minEigThreshold), out.ctypes.data_as(c_f32p), status.ctypes.data_as(c_u8p),
            err.ctypes.data_as(c_f32p)))
    return out.reshape(shape), status.reshape(-1, 1), err.reshape(-1, 1)


class KeyPoint(tuple):
    """(pt = (x, y), size, response): the fields of cv2.KeyPoint the reference reads (`kp.pt`, slam.py:63)."""
    __slots__ = ()
    pt = property(lambda self: self[0])
    size = property(lambda self: self[1])
    response = property(lambda self: self[2])


def bgr_to_gray(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for 8-bit images (fixed point, like OpenCV)."""
    a = np.asarray(img)
    b, g, r = (a[..., k].astype(np.int32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


class FastFeatureDetector:
    def __init__(self, threshold=10, nonmaxSuppression=True):
        self.threshold = int(threshold)
        self.nonmax = bool(nonmaxSuppression)

    def detect_arrays(self, img):
        """Returns (points (n, 2) float32 in row-major scan order, scores (n,) int32)."""
        a = np.asarray(img)
        if a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8:
            a = bgr_to_gray(a)
        im = np.ascontiguousarray(a)
        if im.ndim != 2 or im.dtype != np.uint8:
            raise ValueError("img must be a uint8 grey or BGR image")
        H, W = im.shape
        cap = 4096
        while True:
            xy = np.zeros((cap, 2), dtype=np.float32)
            sc = np.zeros(cap, dtype=np.int32)
            n = np.zeros(1, dtype=np.int32)
            _lib.check(_lib.lib().mqs_fast_detect(_lib.default_context().handle, im.ctypes.data_as(c_u8p), W, H, self.threshold,
                                                  int(self.nonmax), xy.ctypes.data_as(c_f32p), sc.ctypes.data_as(c_i32p), cap,
                                                  n.ctypes.data_as(c_i32p)))
            if int(n[0]) <= cap:
                return xy[:int(n[0])].copy(), sc[:int(n[0])].copy()
            cap = int(n[0])

    def detect(self, img, mask=None):
        if mask is not None:
            raise NotImplementedError("FastFeatureDetector.detect with a mask")
        xy, sc = self.detect_arrays(img)
        return [KeyPoint(((float(x), float(y)), 7.0, float(s))) for (x, y), s in zip(xy, sc)]
