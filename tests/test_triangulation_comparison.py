"""
The product's own experiment harness (triangulation_comparison.py: all trials of a cell in one fused launch per method)
against the reference's committed known-answer files -- all six statistics it stores, including the 2-D reprojection
errors that the oracle harness does not restate -- and the host-side pieces on the CPU."""
import numpy as np
import pytest


def test_points_and_trajectories(mqs, golden3):
    tc = mqs.triangulation_comparison
    np.testing.assert_array_equal(tc.finite_3D_points(4), golden3["points_3D"])             # the reference's own array
    tr = tc.trajectories()
    assert len(tr) == 5 and all(len(t["angle_values"]) == 40 for t in tr)
    assert tr[0]["sideways_values"][-1] == 12 and tr[1]["towards_values"][-1] == 12 and tr[2]["sideways_values"][0] == 12
    assert tr[3]["angle_values"][0] == 0 and tr[4]["angle_values"][-1] == pytest.approx(np.pi / 2)
    assert tr[4]["sideways_values"][-1] == pytest.approx(40.0) and tr[4]["towards_values"][-1] == pytest.approx(40.0)
    e = np.array([[3.0, 4.0, 0.0], [0.0, 0.0, 0.5]])
    rms, med, sq = tc.error_rms(e)
    assert sq.tolist() == [25.0, 0.25] and rms == pytest.approx(np.sqrt(12.625)) and med == pytest.approx(np.sqrt(12.625))
    fp, fn = tc.robustness_stat(np.array([0.5, 2.0, 0.5, 2.0]), np.array([1, 1, 0, -1]))
    assert (fp, fn) == (0.25, 0.25)


@pytest.mark.gpu
def test_test3_cells_equal_the_reference_file(gpu, golden3):
    tc = gpu.triangulation_comparison
    out = tc.test_3()                              # the WHOLE of Test 3: 600 cells x 100 trials x 257 points x 3 methods
    assert out["is_inside_view"]
    np.testing.assert_allclose(out["noise_sigma_values"], golden3["noise_sigma_values"], rtol=1e-15)
    for traj in (0, 2, 3, 4):                      # trajectory 1 ("towards"): rank-deficient cells, medians only (below)
        for ni in range(40):
            for m in (0, 1, 2):
                for key in ("err3D_mean_summary", "err3D_median_summary", "err2D_mean_summary", "err2D_median_summary"):
                    for nty in (0, 1, 2):
                        ref = golden3[key][traj, nty, ni, m]
                        tol = 1e-8 if nty < 2 else 1e-6
                        assert out[key][traj, nty, ni, m] == pytest.approx(ref, rel=tol, abs=1e-12), (key, traj, nty, ni, m)
                for key in ("false_pos_summary", "false_neg_summary"):
                    for nty in (0, 1, 2):
                        assert abs(out[key][traj, nty, ni, m] - golden3[key][traj, nty, ni, m]) <= 3.5 / 25700
    # trajectory 1 ("towards"): the points on the optical axis are rank-deficient -- medians agree, the means are dominated by
    # those points (the reference's own file holds NaN for linear_eigen there)
    for key in ("err3D_median_summary", "err2D_median_summary"):
        for m in (0, 1, 2):
            for nty in (0, 1, 2):
                assert out[key][1, nty, 20, m] == pytest.approx(golden3[key][1, nty, 20, m], rel=0.02), (key, nty, m)
    assert np.isnan(out["err3D_mean_summary"][..., 3]).all()                                  # polynomial: not on this path


@pytest.mark.gpu
def test_test1and2_cells_equal_the_reference_file(gpu):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_1and2_golden.npz"))
    tc = gpu.triangulation_comparison
    out = tc.test_1and2()                          # all 5 x 40 poses
    for traj in (0, 2, 3, 4):
        for pi in range(1 if traj in (0, 3) else 0, 40):     # first pose of trajectories 0 and 3: the cameras coincide
            for m in (0, 1, 2):
                for key in ("err3D_mean_summary", "err3D_median_summary", "err2D_mean_summary", "err2D_median_summary"):
                    if "mean" in key and not g[key][traj, pi, m] < 100.0:
                        continue               # tiny baselines: the mean is a handful of near-singular points (1e46 in the file)
                    assert out[key][traj, pi, m] == pytest.approx(g[key][traj, pi, m], rel=1e-6), (key, traj, pi, m)
                # status flips: <= 6 of 25 700 (k1 = 0.3 tier: the undistortion is pinned to 1e-6) where the geometry is sound; near-singular poses (baseline << depth, almost every
                # point a false positive) put many points next to the error threshold
                tol = 6.5 / 25700 if g["err3D_median_summary"][traj, pi, m] < 2.0 else 2e-3
                for key in ("false_pos_summary", "false_neg_summary"):
                    assert abs(out[key][traj, pi, m] - g[key][traj, pi, m]) <= tol, (key, traj, pi, m)
