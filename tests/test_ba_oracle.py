"""Self-consistency of the BA oracle (oracle/ba_np.py): analytic Jacobians vs finite
differences, Schur form vs the dense Gauss-Newton system, and convergence."""
import numpy as np
import pytest

from oracle import ba_np
from ba_util import make_scene


@pytest.mark.parametrize("distortion", [False, True])
def test_jacobians_match_finite_differences(distortion):
    sc = make_scene(6, 3, seed=3, distortion=distortion)
    h = 1e-6
    for c in range(3):
        for i in range(6):
            pose, K, s, p, uv = sc["poses"][c], sc["calib"][c], sc["sigma"][c], sc["points"][i], sc["obs"][c, i]
            e0, Jp, Jl, ok = ba_np.factor(pose, K, s, p, uv)
            assert ok
            for k in range(6):
                d = np.zeros(6); d[k] = h
                ep = ba_np.factor(ba_np.retract_pose(pose, d), K, s, p, uv)[0]
                em = ba_np.factor(ba_np.retract_pose(pose, -d), K, s, p, uv)[0]
                np.testing.assert_allclose((ep - em) / (2 * h), Jp[:, k], rtol=1e-5, atol=1e-5)
            for k in range(3):
                d = np.zeros(3); d[k] = h
                ep = ba_np.factor(pose, K, s, p + d, uv)[0]
                em = ba_np.factor(pose, K, s, p - d, uv)[0]
                np.testing.assert_allclose((ep - em) / (2 * h), Jl[:, k], rtol=1e-5, atol=1e-5)


def test_schur_equals_dense_system():
    sc = make_scene(12, 3, seed=5, distortion=True)
    S, g, cost, nv, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"])
    H, b = ba_np.dense_reference_step(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"])
    H += 1e-9 * np.eye(len(H))                         # gauge freedom
    n6 = 18
    Hpp, Hpl, Hll = H[:n6, :n6], H[:n6, n6:], H[n6:, n6:]
    Sd = Hpp - Hpl @ np.linalg.solve(Hll, Hpl.T)
    gd = b[:n6] - Hpl @ np.linalg.solve(Hll, b[n6:])
    np.testing.assert_allclose(S, Sd, rtol=1e-6, atol=1e-6 * np.abs(S).max())
    np.testing.assert_allclose(g, gd, rtol=1e-6, atol=1e-6 * np.abs(g).max())
    assert nv == 36 and cost > 0


def test_gauss_newton_converges_with_gauge_priors():
    sc = make_scene(40, 4, seed=7)
    pp = (sc["poses_true"], np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (4, 1)), np.array([1, 0, 0, 0]))
    poses, points, hist = ba_np.gauss_newton(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                             prior_w=sc["prior_w"], prior_xyz=sc["prior_xyz"], pose_prior=pp, iters=8)
    assert hist[-1] < 0.05 * hist[0]
    assert abs(hist[-1] - hist[-2]) < 1e-6 * hist[-1]      # converged
    # residual level ~ pixel noise: chi2 per factor ~ 1
    assert hist[-1] / (40 * 4) < 1.5


def test_cheirality_convention():
    sc = make_scene(5, 2, seed=1, behind=2)
    e, Jp, Jl, ok = ba_np.factor(sc["poses"][0], sc["calib"][0], 2.0, sc["points"][0], sc["obs"][0, 0])
    assert not ok and np.all(Jp == 0) and np.all(Jl == 0)
    np.testing.assert_allclose(e, np.full(2, 2 * 480.0) / 2.0)


@pytest.mark.parametrize("kw", [dict(N=60, C=2), dict(N=80, C=4, distortion=True, masked_frac=0.3), dict(N=50, C=4, behind=5),
                                dict(N=40, C=6, distortion=True)])
def test_c_oracle_matches_numpy_oracle(kw, c_oracle):
    kw = dict(kw)
    N, C = kw.pop("N"), kw.pop("C")
    sc = make_scene(N, C, seed=N, **kw)
    for lam in (0.0, 1e-2):
        S, g, cost, nv, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                                 sc["mask"], sc["prior_w"], sc["prior_xyz"], lam)
        for omp in (False, True):
            Sc, gc, cc, nvc = c_oracle.ba_linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                                    sc["mask"], sc["prior_w"], sc["prior_xyz"], lam, use_omp=omp)
            assert np.abs(Sc - S).max() <= 1e-11 * np.abs(S).max()
            assert np.abs(gc - g).max() <= 1e-11 * np.abs(g).max()
            assert cc == pytest.approx(cost, rel=1e-12) and nvc == nv
        dpose = np.linalg.solve(S + 1e-3 * np.diag(np.diag(S)), g)
        po = c_oracle.ba_backsub(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"], dpose, sc["mask"],
                                 sc["prior_w"], sc["prior_xyz"], lam)
        np.testing.assert_allclose(po, sc["points"] + ba_np.backsub(pieces, dpose), atol=1e-9)


def test_between_factor_jacobians_are_those_of_between():
    """GTSAM's BetweenFactor uses the Jacobians of `between` (-Ad(h^-1), I); at zero error they are exact
    derivatives of the first-order local coordinates under this build's retraction (R Exp(w), t + R v)."""
    rng = np.random.default_rng(0)
    for _ in range(5):
        T1 = np.concatenate([ba_np.so3_exp(rng.normal(0, 0.5, 3)).reshape(-1), rng.normal(0, 2, 3)])
        T2 = np.concatenate([ba_np.so3_exp(rng.normal(0, 0.5, 3)).reshape(-1), rng.normal(0, 2, 3)])
        R1, R2 = T1[:9].reshape(3, 3), T2[:9].reshape(3, 3)
        Tm = np.concatenate([(R1.T @ R2).reshape(-1), R1.T @ (T2[9:] - T1[9:])])
        e, H1, H2 = ba_np.between_error(T1, T2, Tm)
        assert np.abs(e).max() < 1e-12
        h = 1e-6
        for k in range(6):
            d = np.zeros(6); d[k] = h
            j1 = (ba_np.between_error(ba_np.retract_pose(T1, d), T2, Tm)[0] - ba_np.between_error(ba_np.retract_pose(T1, -d), T2, Tm)[0]) / (2 * h)
            j2 = (ba_np.between_error(T1, ba_np.retract_pose(T2, d), Tm)[0] - ba_np.between_error(T1, ba_np.retract_pose(T2, -d), Tm)[0]) / (2 * h)
            np.testing.assert_allclose(j1, H1[:, k], atol=1e-8)
            np.testing.assert_allclose(j2, H2[:, k], atol=1e-8)
