"""
Image front-end (goodFeaturesToTrack at slam2.py:665, calcOpticalFlowPyrLK at slam2.py:381).
Parity with OpenCV is UNPINNED (no images / golden output in the reference, no cv2 here): the GPU kernels are
checked against the numpy restatement of OpenCV 2.4's published method (oracle/features_np.py) and the oracle
itself against analytic properties of synthetic frames (known sub-pixel shifts, known corner positions).
"""
import numpy as np
import pytest

from oracle import features_np as Fn


def texture(H, W, shift=(0.0, 0.0), seed=0, blobs=400):
    """Sum of Gaussian blobs: analytic, so a sub-pixel shifted copy is rendered exactly."""
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(0, W, blobs), rng.uniform(0, H, blobs)
    s, a = rng.uniform(1.5, 4.0, blobs), rng.uniform(-1, 1, blobs)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    x, y = x - shift[0], y - shift[1]
    img = np.zeros((H, W))
    for k in range(blobs):
        img += a[k] * np.exp(-((x - cx[k]) ** 2 + (y - cy[k]) ** 2) / (2 * s[k] ** 2))
    lo, hi = -6.0, 6.0                                     # fixed range: the shifted frame uses the same mapping
    return np.clip(np.rint((img - lo) / (hi - lo) * 255), 0, 255).astype(np.uint8)


def checkerboard(H, W, cell):
    y, x = np.mgrid[0:H, 0:W]
    return (((x // cell + y // cell) % 2) * 200 + 20).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------
# CPU: the oracle against analytic truth
# ---------------------------------------------------------------------------------------------------
def test_oracle_corners_sit_on_checkerboard_crossings():
    img = checkerboard(96, 128, 16)
    pts = Fn.good_features_to_track(img, 0, 0.05, 5.0)
    assert len(pts) == 5 * 7                                # interior crossings of a 6 x 8 board
    # every corner within one pixel of a crossing (multiples of 16)
    d = np.abs(pts - np.rint(pts / 16.0) * 16.0)
    assert d.max() <= 1.0
    # minimum distance honoured, order = decreasing response
    e = Fn.corner_min_eigenval(img)
    r = e[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    assert np.all(np.diff(r) <= 0)
    dd = np.linalg.norm(pts[:, None] - pts[None], axis=2) + 1e9 * np.eye(len(pts))
    assert dd.min() >= 5.0


def test_oracle_mask_and_limits():
    img = texture(120, 160, seed=3)
    allp = Fn.good_features_to_track(img, 0, 0.01, 6.0)
    assert len(allp) > 40
    top = Fn.good_features_to_track(img, 10, 0.01, 6.0)
    np.testing.assert_array_equal(top, allp[:10])
    mask = np.ones(img.shape, np.uint8)
    mask[:, :80] = 0
    right = Fn.good_features_to_track(img, 0, 0.01, 6.0, mask)
    assert len(right) > 10 and right[:, 0].min() >= 80


def test_oracle_lk_recovers_known_shift():
    I = texture(240, 320, seed=0)
    for shift in [(3.3, -1.7), (-7.25, 4.5), (0.4, 0.3)]:
        J = texture(240, 320, shift=shift, seed=0)
        pts = Fn.good_features_to_track(I, 80, 0.01, 8.0)
        nxt, st, err = Fn.calc_optical_flow_pyr_lk(I, J, pts)
        ok = st == 1
        assert ok.sum() >= 0.6 * len(pts)
        flow = (nxt - pts)[ok]
        assert np.abs(np.median(flow, axis=0) - shift).max() < 0.03
        assert np.percentile(np.abs(flow - shift).max(axis=1), 90) < 0.15
        assert np.median(err[ok]) < 2.0
    # a feature whose window leaves the image is reported lost
    nxt, st, err = Fn.calc_optical_flow_pyr_lk(I, I, np.array([[3.0, 3.0], [160.0, 120.0]], np.float32))
    assert st.tolist() == [0, 1] and np.abs(nxt[1] - [160, 120]).max() < 1e-3


def test_oracle_pyramid_and_derivatives():
    img = texture(61, 83, seed=5)
    p = Fn.pyr_down(img)
    assert p.shape == (31, 42)
    flat = np.full((40, 40), 77, np.uint8)
    assert (Fn.pyr_down(flat) == 77).all()                  # kernel sums to 256, rounding
    dx, dy = Fn.scharr_deriv(np.tile(np.arange(40, dtype=np.uint8) * 3, (40, 1)))
    assert (dx[:, 1:-1] == 16 * 2 * 3).all() and (dy == 0).all()


def test_facade_argument_errors(mqs):
    with pytest.raises(ValueError):
        mqs.features.goodFeaturesToTrack(np.zeros((10, 10), np.float32), 5, 0.01, 3)
    with pytest.raises(ValueError):
        mqs.features.calcOpticalFlowPyrLK(np.zeros((10, 10), np.uint8), np.zeros((10, 12), np.uint8), np.zeros((1, 2)))
    assert mqs.features.goodFeaturesToTrack(np.zeros((10, 10), np.uint8), 0, 0.01, 3).shape == (0, 2)   # cv2_helpers.py:35-36


# ---------------------------------------------------------------------------------------------------
# GPU against the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,maxc,q,md,masked", [((120, 160), 50, 0.01, 6.0, False), ((240, 320), 300, 0.01, 7.0, True),
                                                     ((97, 131), 0, 0.02, 1.0, False), ((480, 640), 300, 0.01, 7.0, True),
                                                     ((64, 64), 20, 0.05, 0.0, False), ((200, 300), 1000, 0.001, 2.5, False),
                                                     # wide minimum distances: every candidate has many stronger ones nearby
                                                     # (more than the selection's per-candidate neighbour list holds)
                                                     ((240, 320), 0, 0.001, 25.0, False), ((480, 640), 40, 0.0005, 60.0, True),
                                                     ((480, 640), 0, 0.001, 1.0, False)])
def test_gpu_good_features_equal_oracle(shape, maxc, q, md, masked, gpu):
    img = texture(shape[0], shape[1], seed=shape[0])
    mask = None
    if masked:                                              # slam2.py:29-40 keypoint_mask: discs around existing points cleared
        mask = np.ones(shape, np.uint8)
        rng = np.random.default_rng(1)
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        for cx, cy in zip(rng.uniform(0, shape[1], 25), rng.uniform(0, shape[0], 25)):
            mask[(xx - cx) ** 2 + (yy - cy) ** 2 <= 49] = 0
    ref = Fn.good_features_to_track(img, maxc, q, md, mask)
    got = gpu.features.goodFeaturesToTrack(img, maxc, q, md, None, mask, capacity=max(len(ref) + 8, 1))
    assert got.dtype == np.float32 and got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)                 # same corners in the same order: exact


@pytest.mark.gpu
@pytest.mark.parametrize("md", [0.0, 3.0])
def test_gpu_good_features_more_candidates_than_one_sort_chunk(md, gpu):
    """A frame of noise: 20 342 local maxima pass the quality level, more than the 16 384 keys the selection kernel sorts in
    LDS at a time, so its chunks are merged.  Same corners in the same order as the oracle."""
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    ref = Fn.good_features_to_track(img, 0, 1e-4, md)
    assert len(ref) > 16384 if md == 0.0 else len(ref) > 5000
    got = gpu.features.goodFeaturesToTrack(img, 0, 1e-4, md, capacity=len(ref) + 8)
    np.testing.assert_array_equal(got, ref)
    top = gpu.features.goodFeaturesToTrack(img, 300, 1e-4, md)
    np.testing.assert_array_equal(top, ref[:300])


@pytest.mark.gpu
@pytest.mark.parametrize("min_distance", [1.0, 2.0, 5.0, 12.0, 30.0, 64.0])
def test_gpu_good_features_selection_walks_the_list_in_groups_like_the_greedy_rule(min_distance, gpu):
    """The minimum-distance selection goes through the sorted candidates 64 at a time (round 6): against the corners accepted in earlier
    groups through the grid, inside a group on wave-wide masks settled by repeated looks.  A frame of noise makes the dependency chains
    inside a group long at large distances (most of a group lies within 30 or 64 pixels of each other) and the groups many at small
    ones; the limit cuts a group in the middle.  The same corners in the same order as the oracle's sequential walk."""
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    ref = Fn.good_features_to_track(img, 0, 1e-4, min_distance)
    got = gpu.features.goodFeaturesToTrack(img, 0, 1e-4, min_distance, capacity=len(ref) + 8)
    np.testing.assert_array_equal(got, ref)
    for limit in (1, 37, 64, 65, 300):
        np.testing.assert_array_equal(gpu.features.goodFeaturesToTrack(img, limit, 1e-4, min_distance), ref[:limit])
    tex = texture(240, 320, seed=240)
    np.testing.assert_array_equal(gpu.features.goodFeaturesToTrack(tex, 300, 0.01, min_distance), Fn.good_features_to_track(tex, 300, 0.01, min_distance))


@pytest.mark.gpu
def test_gpu_good_features_checkerboard_and_flat(gpu):
    pts = gpu.features.goodFeaturesToTrack(checkerboard(96, 128, 16), 0, 0.05, 5.0, capacity=100)
    np.testing.assert_array_equal(pts, Fn.good_features_to_track(checkerboard(96, 128, 16), 0, 0.05, 5.0))
    assert len(gpu.features.goodFeaturesToTrack(np.full((50, 60), 9, np.uint8), 10, 0.01, 3.0)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,shift", [((240, 320), (3.3, -1.7)), ((480, 640), (-9.5, 6.25)), ((120, 160), (0.4, 0.3))])
def test_gpu_lk_equals_oracle_and_truth(shape, shift, gpu):
    I = texture(shape[0], shape[1], seed=7, blobs=300 * shape[0] * shape[1] // (240 * 320))
    J = texture(shape[0], shape[1], shift=shift, seed=7, blobs=300 * shape[0] * shape[1] // (240 * 320))
    pts = Fn.good_features_to_track(I, 120, 0.01, 8.0)
    pts = np.concatenate([pts, np.array([[2.0, 2.0], [shape[1] - 3.0, 5.0]], np.float32)])     # two lost at the border
    rn, rs, re = Fn.calc_optical_flow_pyr_lk(I, J, pts)
    gn, gs, ge = gpu.features.calcOpticalFlowPyrLK(I, J, pts)
    assert gn.shape == pts.shape and gs.shape == (len(pts), 1) and ge.shape == (len(pts), 1)
    np.testing.assert_array_equal(gs.ravel(), rs)            # same features tracked / lost
    ok = rs == 1
    assert np.abs(gn[ok] - rn[ok]).max() < 2e-3              # float32 window sums in a different order
    np.testing.assert_allclose(ge.ravel()[ok], re[ok], rtol=1e-3, atol=1e-3)
    flow = (gn - pts)[ok]
    assert np.abs(np.median(flow, axis=0) - shift).max() < 0.05
    # OpenCV-shaped input (n, 1, 2)
    gn2, _, _ = gpu.features.calcOpticalFlowPyrLK(I, J, pts.reshape(-1, 1, 2))
    assert gn2.shape == (len(pts), 1, 2)
    np.testing.assert_array_equal(gn2.reshape(-1, 2), gn)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,max_level", [((480, 640), 3), ((481, 643), 3), ((531, 777), 3), ((300, 404), 2), ((135, 150), 1)])
def test_gpu_lk_pyramid_in_one_launch_equals_the_per_level_launches(shape, max_level, gpu, monkeypatch):
    """The pyramid of both images, the derivative images and the border-extended copies the tracker reads come out of ONE launch where
    the top level is at least a border (32 pixels) wide and high; MQS_LK_PYRAMID_PER_LEVEL=1 keeps the launch per level + the padding
    launch.  Integer images: the tracker must return the same bits from both -- features at the image border included (windows that
    reach into the reflected border), odd sizes at every level."""
    H, W = shape
    I = texture(H, W, seed=5, blobs=300 * H * W // (240 * 320))
    J = texture(H, W, shift=(2.4, -1.3), seed=5, blobs=300 * H * W // (240 * 320))
    pts = Fn.good_features_to_track(I, 200, 0.01, 8.0)
    rim = np.array([[x, y] for x in (1.0, 9.5, W / 2, W - 11.0, W - 2.0) for y in (1.0, 8.0, H / 2, H - 9.5, H - 2.0)], np.float32)
    pts = np.concatenate([pts, rim])
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MQS_LK_PYRAMID_PER_LEVEL", mode)
        out[mode] = gpu.features.calcOpticalFlowPyrLK(I, J, pts, maxLevel=max_level)
    for a, b in zip(out["0"], out["1"]):
        np.testing.assert_array_equal(a, b)
    assert out["0"][1].sum() > 50


@pytest.mark.gpu
def test_gpu_detect_then_track_loop(gpu):
    """slam2.py's front-end cycle on rendered frames: detect, track over 5 frames of accumulating motion, top up under
    the coverage mask."""
    I0 = texture(240, 320, seed=11)
    pts = gpu.features.goodFeaturesToTrack(I0, 150, 0.01, 7.0)
    assert len(pts) >= 100
    cur, prev_img, total = pts.copy(), I0, np.zeros(2)
    alive = np.ones(len(pts), bool)
    for k in range(1, 6):
        total += (1.7, -0.9)
        img = texture(240, 320, shift=tuple(total), seed=11)
        nxt, st, err = gpu.features.calcOpticalFlowPyrLK(prev_img, img, cur)
        good = (st.ravel() == 1) & (err.ravel() < 7.0)      # slam2.py:382 max_OF_error
        alive &= good
        cur, prev_img = nxt, img
    assert alive.mean() > 0.7
    assert np.abs(np.median((cur - pts)[alive], axis=0) - total).max() < 0.1
    mask = np.ones(I0.shape, np.uint8)
    yy, xx = np.mgrid[0:240, 0:320]
    for x, y in cur[alive]:
        mask[(xx - x) ** 2 + (yy - y) ** 2 <= 49] = 0
    extra = gpu.features.goodFeaturesToTrack(prev_img, 50, 0.01, 7.0, None, mask)
    if len(extra):
        d = np.linalg.norm(extra[:, None] - cur[alive][None], axis=2)
        assert d.min() > 7.0


# ---------------------------------------------------------------------------------------------------
# FAST-9/16 (slam.py:34, 62)
# ---------------------------------------------------------------------------------------------------
def rectangles(H, W, seed=0, n=60):
    """Random axis-aligned bright / dark rectangles on grey: plenty of L-corners (what FAST fires on)."""
    rng = np.random.default_rng(seed)
    img = np.full((H, W), 120, np.int32)
    for _ in range(n):
        x0, y0 = rng.integers(0, W - 12), rng.integers(0, H - 12)
        w, h = rng.integers(8, 40), rng.integers(8, 40)
        img[y0:y0 + h, x0:x0 + w] = rng.integers(0, 255)
    img += rng.integers(-3, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_oracle_fast_fires_on_rectangle_corners_not_edges():
    img = np.full((60, 80), 50, np.int32)
    img[20:40, 30:60] = 200
    img += np.random.default_rng(0).integers(-4, 5, img.shape)       # break the exact score ties of a synthetic image
    img = img.astype(np.uint8)
    corners = np.array([[30, 20], [59, 20], [30, 39], [59, 39]])
    raw, _ = Fn.fast_detect(img, 10, nonmax=False)
    assert len(raw) >= 4 and all(np.abs(corners - p).max(axis=1).min() <= 2 for p in raw)     # only near the 4 corners
    pts, sc = Fn.fast_detect(img, 10)
    assert 4 <= len(pts) <= len(raw)
    assert {int(np.abs(corners - p).max(axis=1).argmin()) for p in pts} == {0, 1, 2, 3}       # each corner found
    assert (sc >= 10).all()
    assert len(Fn.fast_detect(np.tile(np.arange(80, dtype=np.uint8) * 3, (60, 1)), 10)[0]) == 0     # a ramp has no corners


@pytest.mark.gpu
@pytest.mark.parametrize("shape,thr,nonmax", [((120, 160), 10, True), ((480, 640), 10, True), ((97, 131), 25, True),
                                               ((240, 320), 5, False), ((7, 7), 10, True), ((5, 40), 10, True)])
def test_gpu_fast_equals_oracle(shape, thr, nonmax, gpu):
    img = rectangles(shape[0], shape[1], seed=shape[1]) if min(shape) > 20 else np.full(shape, 9, np.uint8)
    ref_xy, ref_s = Fn.fast_detect(img, thr, nonmax)
    det = gpu.features.FastFeatureDetector(thr, nonmax)
    xy, sc = det.detect_arrays(img)
    np.testing.assert_array_equal(xy, ref_xy)               # same corners, same (row-major) order
    np.testing.assert_array_equal(sc, ref_s)
    kps = det.detect(img)
    assert len(kps) == len(ref_xy)
    if len(kps):
        assert kps[0].pt == (float(ref_xy[0, 0]), float(ref_xy[0, 1])) and kps[0].size == 7.0
    if min(shape) > 20:
        assert len(ref_xy) > 20


@pytest.mark.gpu
def test_gpu_fast_bgr_and_reference_pipeline(gpu):
    """slam.py:62-104: FAST corners of the right image, LK from the left image, 2-NN radius match OF point -> FAST corner."""
    left = rectangles(240, 320, seed=4)
    right = np.roll(left, (2, 3), axis=(0, 1))              # integer shift: right(x + 3, y + 2) = left(x, y)
    bgr = np.stack([right, right, right], axis=2)
    det = gpu.features.FastFeatureDetector()
    kp_bgr = det.detect(bgr)
    xy_r, _ = det.detect_arrays(right)
    assert np.array_equal(np.array([k.pt for k in kp_bgr], np.float32).reshape(-1, 2), xy_r)     # grey of equal channels = itself
    xy_l, _ = det.detect_arrays(left)
    inner = xy_l[(xy_l[:, 0] > 20) & (xy_l[:, 0] < 290) & (xy_l[:, 1] > 20) & (xy_l[:, 1] < 210)]
    of, st, err = gpu.features.calcOpticalFlowPyrLK(left, right, inner)
    good = (st.ravel() == 1) & (err.ravel() < 7.0)
    assert good.mean() > 0.6
    m = gpu.matching.BFMatcher().radiusMatch(of[good], xy_r, 2.0)        # slam.py:101-104
    hit = [ms[0] for ms in m if ms]
    assert len(hit) > 0.7 * good.sum()
    d = np.array([xy_r[h.trainIdx] - (inner[good][h.queryIdx] + [3, 2]) for h in hit])
    assert np.median(np.abs(d)) < 0.5                       # the matched FAST corner is the shifted left corner
