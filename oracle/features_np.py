"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the image front-end the reference's
per-frame loop calls through OpenCV 2.4 (an external dependency, not vendored, not installable here):

    goodFeaturesToTrack   Work/python_libs/cv2_helpers.py:34-37, called at Work/SLAM/application/own/slam2.py:665, 1174
                          (Shi-Tomasi minimum-eigenvalue corners, blockSize 3, Sobel aperture 3, coverage mask)
    calcOpticalFlowPyrLK  slam2.py:381 (defaults: 21 x 21 window, maxLevel 3, 30 iterations / eps 0.01,
                          minEigThreshold 1e-4)
    FastFeatureDetector   Work/SLAM/application/own/slam.py:34 (FAST-9/16, threshold 10, non-max suppression)

PARITY: pinned loosely on reference-held data, not call by call.  The reference holds no recorded OUTPUT of these three calls
(no corner lists, no tracked positions) and OpenCV cannot be run here; what is restated is the published method of OpenCV 2.4
(cornerMinEigenVal / goodFeaturesToTrack, lkpyramid.cpp, fast.cpp) with float32 arithmetic in a fixed, documented order.  Where
OpenCV works in fixed point (bilinear weights in 14 bits, rounding descale) this restatement uses float32; results agree with
OpenCV's to its quantisation, not bit for bit.  What the reference DOES hold (found in round 4) is its own example run on real
images: 200 frames of the ICL-NUIM living-room sequence, the initial pose and 23 known 3-D corners, the renderer's exact
trajectory, and the trajectory slam2.py wrote for these frames with the real OpenCV calls inside (slam2.py:924-933;
Work/SLAM/datasets/ICL_NUIM/living_room_traj3n_frei_png).  Against that (tests/test_icl_nuim.py, fixture by
tests/golden/make_icl_nuim.py):
  * calc_optical_flow_pyr_lk tracks the 23 known corners through the real, noisy frames to where the exact trajectory projects
    them: median 0.03-0.05 px, 90th percentile 0.09-0.16 px over five frames;
  * the loop built on the GPU twins of these functions reproduces the reference's committed trajectory frame by frame within
    about what either keeps from the exact one (4.3-10.9 mm rmse over the 80 frames of the fixture across four RANSAC seeds;
    the reference's own 4.4 mm);
  * good_features_to_track fills slam2.py's quota on frame 0 (277 corners beside the 23 initial points: the reference's record of
    a run on this sequence shows 296 tracked points there, i.e. at least 273 corners) -- which it did not before its threshold was taken from the maximum under the mask.
FAST stays unpinned.  The GPU kernels are tested against THIS file (synthetic frames and the
real ones) and against analytic properties of synthetic frames (known shifts, known corner positions).
"""
import numpy as np

F = np.float32


def _pad101(a, r):
    return np.pad(a, r, mode="reflect")                      # BORDER_REFLECT_101 (OpenCV's BORDER_DEFAULT)


# ---------------------------------------------------------------------------------------------------
# goodFeaturesToTrack
# ---------------------------------------------------------------------------------------------------
def corner_min_eigenval(img):
    """cornerMinEigenVal(img, blockSize=3, ksize=3) for 8-bit input: Sobel derivatives scaled by
    1 / (2^(ksize-1) * blockSize * 255), products, un-normalised 3x3 box sums, then
    (a + c) - sqrt((a - c)^2 + b^2) with a = sxx / 2, b = sxy, c = syy / 2.  All float32, fixed order."""
    scale = F(1.0) / (F(4.0) * F(3.0) * F(255.0))
    p = _pad101(img.astype(F), 1)
    H, W = img.shape
    s = lambda dy, dx: p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    dx = ((s(-1, 1) - s(-1, -1)) + F(2) * (s(0, 1) - s(0, -1))) + (s(1, 1) - s(1, -1))
    dy = ((s(1, -1) - s(-1, -1)) + F(2) * (s(1, 0) - s(-1, 0))) + (s(1, 1) - s(-1, 1))
    dx = dx * scale
    dy = dy * scale
    xx, xy, yy = dx * dx, dx * dy, dy * dy

    def box(a):
        q = _pad101(a, 1)
        t = lambda oy, ox: q[1 + oy:1 + oy + H, 1 + ox:1 + ox + W]
        rows = [(t(oy, -1) + t(oy, 0)) + t(oy, 1) for oy in (-1, 0, 1)]
        return (rows[0] + rows[1]) + rows[2]

    a = box(xx) * F(0.5)
    b = box(xy)
    c = box(yy) * F(0.5)
    d = a - c
    return ((a + c) - np.sqrt(d * d + b * b)).astype(F)


def good_features_to_track(img, max_corners, quality_level, min_distance, mask=None):
    """Returns (n, 2) float32 (x, y).  Candidates: response > quality * max (the max over the unmasked pixels), equal to the 3x3 dilation, not on
    the 1-pixel border, mask != 0; ordered by response (descending; ties: row-major position -- OpenCV's
    std::sort leaves ties unspecified); greedy minimum-distance selection, at most max_corners (0: no limit)."""
    eig = corner_min_eigenval(img)
    H, W = eig.shape
    # the maximum under the mask: featureselect.cpp (2.4) `minMaxLoc(eig, 0, &maxVal, 0, 0, mask)` -- found from the reference's own
    # example run (round 4): with the whole image's maximum frame 0 of the ICL-NUIM sequence yields 216 corners where slam2.py's
    # record shows at least 273 (tests/test_icl_nuim.py)
    thr = F(eig.max() if mask is None else (eig[mask != 0].max() if np.any(mask != 0) else 0.0)) * F(quality_level)
    e = np.where(eig > thr, eig, F(0))                       # threshold(..., THRESH_TOZERO)
    q = np.pad(e, 1, mode="constant", constant_values=-np.inf)   # dilate ignores pixels outside the image
    dil = np.max(np.stack([q[1 + oy:1 + oy + H, 1 + ox:1 + ox + W] for oy in (-1, 0, 1) for ox in (-1, 0, 1)]), axis=0)
    cand = (e != 0) & (e == dil)
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    if mask is not None:
        cand &= mask != 0
    ys, xs = np.nonzero(cand)
    order = np.lexsort((ys * W + xs, -e[ys, xs].astype(np.float64)))
    ys, xs = ys[order], xs[order]
    out = []
    if min_distance >= 1:
        cell = int(round(min_distance))                       # cvRound(minDistance)
        gw, gh = (W + cell - 1) // cell, (H + cell - 1) // cell
        grid = [[] for _ in range(gw * gh)]
        md2 = min_distance * min_distance
        for y, x in zip(ys, xs):
            cx, cy = x // cell, y // cell
            good = True
            for yy in range(max(0, cy - 1), min(gh - 1, cy + 1) + 1):
                for xx in range(max(0, cx - 1), min(gw - 1, cx + 1) + 1):
                    for (px, py) in grid[yy * gw + xx]:
                        if (x - px) ** 2 + (y - py) ** 2 < md2:
                            good = False
                            break
                    if not good:
                        break
                if not good:
                    break
            if good:
                grid[cy * gw + cx].append((x, y))
                out.append((x, y))
                if 0 < max_corners <= len(out):
                    break
    else:
        for y, x in zip(ys, xs):
            out.append((x, y))
            if 0 < max_corners <= len(out):
                break
    return np.array(out, dtype=F).reshape(-1, 2)


# ---------------------------------------------------------------------------------------------------
# calcOpticalFlowPyrLK
# ---------------------------------------------------------------------------------------------------
def pyr_down(img):
    """pyrDown for 8-bit images: separable [1 4 6 4 1] (sum 256 over both passes), rounding (+128) >> 8,
    even pixels kept, size ((W+1)//2, (H+1)//2), BORDER_REFLECT_101."""
    H, W = img.shape
    p = _pad101(img.astype(np.int32), 2)
    k = (1, 4, 6, 4, 1)
    h = sum(k[i] * p[:, i:i + W] for i in range(5))          # horizontal pass, all padded rows
    v = sum(k[i] * h[i:i + H, :] for i in range(5))
    out = (v + 128) >> 8
    return out[0::2, 0::2].astype(np.uint8)


def build_pyramid(img, max_level):
    pyr = [np.ascontiguousarray(img, dtype=np.uint8)]
    for _ in range(max_level):
        if min(pyr[-1].shape) <= 2:
            break
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def scharr_deriv(img):
    """Un-normalised Scharr derivatives (3, 10, 3) x (-1, 0, 1), int16, REFLECT_101 -- calcSharrDeriv."""
    H, W = img.shape
    p = _pad101(img.astype(np.int32), 1)
    s = lambda dy, dx: p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    dx = 3 * (s(-1, 1) - s(-1, -1)) + 10 * (s(0, 1) - s(0, -1)) + 3 * (s(1, 1) - s(1, -1))
    dy = 3 * (s(1, -1) - s(-1, -1)) + 10 * (s(1, 0) - s(-1, 0)) + 3 * (s(1, 1) - s(-1, 1))
    return dx.astype(np.int16), dy.astype(np.int16)


def _bilinear_window(a, ix, iy, fx, fy, w, h):
    """Window [iy, iy + h) x [ix, ix + w) of `a` sampled at offset (fx, fy) in [0,1): float32 weights in the
    order w00 v00 + w01 v01 + w10 v10 + w11 v11 (left to right)."""
    v = a[iy:iy + h + 1, ix:ix + w + 1].astype(F)
    w00, w01 = F((1 - fx) * (1 - fy)), F(fx * (1 - fy))
    w10, w11 = F((1 - fx) * fy), F(fx * fy)
    return ((v[:-1, :-1] * w00 + v[:-1, 1:] * w01) + v[1:, :-1] * w10) + v[1:, 1:] * w11


def _refl101(idx, n):
    """BORDER_REFLECT_101 index (one reflection, then clamped: a window reaches at most win + 1 pixels beyond the image)."""
    idx = np.abs(idx)
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    return np.clip(idx, 0, n - 1)


def _bilinear_window_border(a, ix, iy, fx, fy, w, h, zero_outside):
    """_bilinear_window for a window that leaves the image, read as OpenCV 2.4's tracker reads it from the border-extended
    pyramid (lkpyramid.cpp: buildOpticalFlowPyramid pads every level by winSize with BORDER_REFLECT_101; the derivative image is
    padded with BORDER_CONSTANT zeros): zero_outside=False reflects, True reads 0 outside the image."""
    H, W = a.shape
    xs, ys = ix + np.arange(w + 1), iy + np.arange(h + 1)
    v = a[_refl101(ys, H)][:, _refl101(xs, W)].astype(F)
    if zero_outside:
        v = v * (((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]).astype(F)
    w00, w01 = F((1 - fx) * (1 - fy)), F(fx * (1 - fy))
    w10, w11 = F((1 - fx) * fy), F(fx * fy)
    return ((v[:-1, :-1] * w00 + v[:-1, 1:] * w01) + v[1:, :-1] * w10) + v[1:, 1:] * w11


def _window(a, ix, iy, fx, fy, w, h, zero_outside=False):
    H, W = a.shape
    if ix >= 0 and iy >= 0 and ix + w + 1 <= W and iy + h + 1 <= H:
        return _bilinear_window(a, ix, iy, fx, fy, w, h)
    return _bilinear_window_border(a, ix, iy, fx, fy, w, h, zero_outside)


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, win_size=(21, 21), max_level=3, max_iter=30, eps=0.01,
                             min_eig_threshold=1e-4):
    """Returns next_pts (n, 2) float32, status (n,) uint8, err (n,) float32 (mean absolute intensity difference
    over the window at level 0, like OpenCV without OPTFLOW_LK_GET_MIN_EIGENVALS)."""
    prev_pts = np.asarray(prev_pts, dtype=F).reshape(-1, 2)
    n = len(prev_pts)
    ww, wh = win_size
    half = np.array([(ww - 1) * 0.5, (wh - 1) * 0.5], dtype=F)
    pI = build_pyramid(prev_img, max_level)
    pJ = build_pyramid(next_img, max_level)
    levels = min(len(pI), len(pJ)) - 1
    derivs = [scharr_deriv(im) for im in pI]
    next_pts = np.zeros((n, 2), dtype=F)
    status = np.ones(n, dtype=np.uint8)
    err = np.zeros(n, dtype=F)
    FLT_SCALE = F(1.0 / (1 << 20))
    for level in range(levels, -1, -1):
        I, J = pI[level], pJ[level]
        dIx, dIy = derivs[level]
        H, W = I.shape
        for k in range(n):
            prev = prev_pts[k] * F(1.0 / (1 << level))
            nxt = prev.copy() if level == levels else next_pts[k] * F(2.0)
            next_pts[k] = nxt
            prev = prev - half
            ipx, ipy = int(np.floor(prev[0])), int(np.floor(prev[1]))
            if ipx < -ww or ipx >= W or ipy < -wh or ipy >= H:
                if level == 0:
                    status[k] = 0
                    err[k] = 0
                continue
            # a window may leave the image by up to its own size (the test above is OpenCV's): it is read through the
            # border-extended pyramid -- intensities reflected (BORDER_REFLECT_101), derivatives zero outside the image
            # (BORDER_CONSTANT).  (Until round 4 this restatement required the window inside the image and lost every corner
            # within half a window of the border: 28 of 277 on frame 0 of the reference's example sequence, where the reference's
            # own record shows next to none lost.)
            a, b = F(prev[0] - ipx), F(prev[1] - ipy)
            Iw = _window(I, ipx, ipy, a, b, ww, wh) * F(32.0)
            Ixw = _window(dIx, ipx, ipy, a, b, ww, wh, zero_outside=True)
            Iyw = _window(dIy, ipx, ipy, a, b, ww, wh, zero_outside=True)
            A11 = F(np.sum(Ixw.astype(np.float64) * Ixw)) * FLT_SCALE
            A12 = F(np.sum(Ixw.astype(np.float64) * Iyw)) * FLT_SCALE
            A22 = F(np.sum(Iyw.astype(np.float64) * Iyw)) * FLT_SCALE
            D = F(A11 * A22 - A12 * A12)
            min_eig = F((A22 + A11 - np.sqrt(F((A11 - A22) * (A11 - A22) + F(4.0) * A12 * A12))) / F(2 * ww * wh))
            if min_eig < min_eig_threshold or D < np.finfo(F).eps:
                if level == 0:
                    status[k] = 0
                continue
            D = F(1.0) / D
            nxt = nxt - half
            prev_delta = np.zeros(2, dtype=F)
            for j in range(max_iter):
                inx, iny = int(np.floor(nxt[0])), int(np.floor(nxt[1]))
                if inx < -ww or inx >= W or iny < -wh or iny >= H:
                    if level == 0:
                        status[k] = 0
                    break
                a, b = F(nxt[0] - inx), F(nxt[1] - iny)
                Jw = _window(J, inx, iny, a, b, ww, wh) * F(32.0)
                diff = Jw - Iw
                b1 = F(np.sum(diff.astype(np.float64) * Ixw)) * FLT_SCALE
                b2 = F(np.sum(diff.astype(np.float64) * Iyw)) * FLT_SCALE
                delta = np.array([(A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D], dtype=F)
                nxt = nxt + delta
                if delta[0] * delta[0] + delta[1] * delta[1] <= eps * eps:
                    break
                if j > 0 and abs(delta[0] + prev_delta[0]) < 0.01 and abs(delta[1] + prev_delta[1]) < 0.01:
                    nxt = nxt - delta * F(0.5)
                    break
                prev_delta = delta
            next_pts[k] = nxt + half
            if status[k] and level == 0:
                p = next_pts[k] - half
                inx, iny = int(np.floor(p[0])), int(np.floor(p[1]))
                if inx < -ww or inx >= W or iny < -wh or iny >= H:
                    status[k] = 0
                else:
                    a, b = F(p[0] - inx), F(p[1] - iny)
                    Jw = _window(J, inx, iny, a, b, ww, wh) * F(32.0)
                    err[k] = F(np.sum(np.abs((Jw - Iw).astype(np.float64))) / (32.0 * ww * wh))
    return next_pts, status, err


# ---------------------------------------------------------------------------------------------------
# FAST-9/16 (FastFeatureDetector(threshold=10, nonmaxSuppression=True), slam.py:34, 62)
# ---------------------------------------------------------------------------------------------------
FAST_CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
               (-3, 0), (-3, 1), (-2, 2), (-1, 3)]          # (dx, dy), OpenCV's makeOffsets order for patternSize 16


def bgr_to_gray(img):
    """cv2.cvtColor(BGR2GRAY) for 8-bit images: (B*1868 + G*9617 + R*4899 + 8192) >> 14."""
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def fast_score_image(gray, threshold):
    """Per pixel: 0 when not a FAST-9 corner at `threshold` (or within 3 pixels of the border), else the corner
    score of OpenCV 2.4's cornerScore<16>: the largest t for which the pixel is still a corner, i.e.
    max over the 16 arcs of 9 contiguous circle pixels of min |difference| over the arc (sign-consistent), minus 1."""
    H, W = gray.shape
    g = gray.astype(np.int32)
    score = np.zeros((H, W), dtype=np.int32)
    if H < 7 or W < 7:
        return score
    c = g[3:H - 3, 3:W - 3]
    d = np.stack([c - g[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for (dx, dy) in FAST_CIRCLE])    # centre - ring
    best_dark = np.full(c.shape, -10 ** 9)                   # ring darker than centre: d > 0
    best_bright = np.full(c.shape, -10 ** 9)                 # ring brighter: -d > 0
    for s in range(16):
        arc = np.stack([d[(s + k) % 16] for k in range(9)])
        best_dark = np.maximum(best_dark, arc.min(axis=0))
        best_bright = np.maximum(best_bright, (-arc).min(axis=0))
    m = np.maximum(best_dark, best_bright)
    score[3:H - 3, 3:W - 3] = np.where(m > threshold, m - 1, 0)
    return score


def fast_detect(gray, threshold=10, nonmax=True):
    """Returns (n, 2) float32 (x, y) in row-major scan order and their scores (int32)."""
    s = fast_score_image(gray, threshold)
    H, W = s.shape
    keep = s > 0
    if nonmax:
        p = np.pad(s, 1)
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                if oy or ox:
                    keep &= s > p[1 + oy:1 + oy + H, 1 + ox:1 + ox + W]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys], axis=1).astype(F), s[ys, xs]
