"""
End-to-end monocular loop (slam_loop.MonoSlam = slam2.py's handle_new_frame state machine) on a rendered sequence with
known camera motion and known scene geometry (a textured plane) -- BASELINE configs[4].  No reference output exists for
images (parity unpinned); what is asserted is the accuracy against ground truth and the reference's own gates.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_keypoint_mask_and_homography_helpers(mqs):
    L = mqs.slam_loop
    mk = L.keypoint_mask((480, 640), np.array([[5.0, 5.0], [320.2, 240.7], [639, 479]]))
    assert mk[240, 320] == 0 and mk[240, 333] == 1 and mk[0, 0] == 0 and mk[253, 320] == 0 and mk[254, 320] == 1
    assert L.keypoint_mask((10, 10), np.zeros((0, 2))).all()
    rng = np.random.default_rng(0)
    p1 = rng.uniform(-1, 1, (50, 2))
    Ht = np.array([[1.02, 0.01, 0.03], [-0.02, 0.98, 0.01], [0.01, 0.02, 1.0]])
    q = np.c_[p1, np.ones(50)] @ Ht.T
    assert np.abs(L.homography_dlt(p1, q[:, :2] / q[:, 2:]) - Ht).max() < 1e-12


def test_rendered_sequence_is_consistent(mqs):
    seq = mqs.synthetic.PlaneSequence(frames=5)
    img = seq.render(0)
    assert img.shape == (480, 640) and img.dtype == np.uint8 and img.std() > 20
    # a plane point projects where the renderer drew it: the texture value at the projection matches the texture
    P = np.array([[-2.0, 0.5, 0.0], [0.0, -1.0, 0.0]])
    uv = seq.project(0, P)
    n = seq.tex.shape[0]
    for (u, v), X in zip(uv, P):
        t = seq.tex[int(round((X[1] + seq.extent) * (n - 1) / (2 * seq.extent))), int(round((X[0] + seq.extent) * (n - 1) / (2 * seq.extent)))]
        assert abs(float(img[int(round(v)), int(round(u))]) - t) < 12


@pytest.mark.gpu
def test_end_to_end_loop_on_rendered_sequence(gpu):
    import run_slam_loop
    out = run_slam_loop.run(40)
    assert out["accepted"] == 40 and out["keyframes"] >= 8
    assert out["landmarks_triangulated"] >= 200
    assert out["trajectory_rmse"] < 0.01 * out["path_length"]                 # < 1 % of the path (measured 0.3 %)
    assert out["map_plane_median_abs_z"] < 0.15                                # landmarks lie on the plane z = 0 (depth ~ 9)
