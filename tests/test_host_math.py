"""
CPU check of the arithmetic the gfx950 kernels execute: csrc/tri_math.h compiled for the host
(tests/host_math.cpp) against the oracle and against the reference's golden cells.  This is
NOT a product path (nothing in the package can reach it); it lets the no-GPU suite catch
arithmetic regressions before GPU time is spent.
"""
import ctypes
import numpy as np
import pytest

from oracle import harness_np as H
from util import random_scene, rel_err, stable_mask, assert_excluded_explained

F64 = ctypes.POINTER(ctypes.c_double)


def host_tri(lib, kind, u, P, tol=3e-5, max_iter=10, max_coord=1e16):
    u = np.ascontiguousarray(u, dtype=np.float64)
    P = np.ascontiguousarray(np.asarray(P)[:, :3], dtype=np.float64)
    C, N, _ = u.shape
    x = np.empty((N, 3))
    st = np.zeros(N, np.int32)
    ok = np.zeros(N, np.uint8)
    rc = lib.host_tri(kind, u.ctypes.data_as(F64), P.ctypes.data_as(F64), C, ctypes.c_int64(N),
                      ctypes.c_double(tol), max_iter, ctypes.c_double(max_coord), x.ctypes.data_as(F64),
                      st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                      ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    assert rc == 0
    return x, st, ok.astype(bool)


@pytest.mark.parametrize("cell", [(0, 0, 8), (2, 1, 20), (3, 0, 8), (4, 0, 8), (4, 1, 39)])
def test_device_math_reproduces_golden_cells(cell, golden3, host_math):
    methods = [lambda u, P: host_tri(host_math, 2, u, P)[0::2],
               lambda u, P: (host_tri(host_math, 0, u, P)[0], np.ones(u.shape[1], bool)),
               lambda u, P: host_tri(host_math, 1, u, P)[0:2]]
    tr, nt, si = cell
    res = H.test_3_cell(tr, nt, golden3["noise_sigma_values"][si], methods, int(golden3["num_trials"]))
    n = 257 * int(golden3["num_trials"])
    for m in range(3):
        assert res[m][0] == pytest.approx(golden3["err3D_mean_summary"][tr, nt, si, m], rel=1e-8)
        assert res[m][1] == pytest.approx(golden3["err3D_median_summary"][tr, nt, si, m], rel=1e-8)
        assert abs(res[m][2] - golden3["false_pos_summary"][tr, nt, si, m]) <= 3 / n
        assert abs(res[m][3] - golden3["false_neg_summary"][tr, nt, si, m]) <= 3 / n


@pytest.mark.parametrize("C", [2, 3, 4, 5, 8])
def test_device_math_vs_oracle_nview(C, host_math, c_oracle):
    u, P, _ = random_scene(3000, C, seed=100 + C, behind_frac=0.1)
    for kind, fn in ((0, c_oracle.linear_LS_triangulation), (1, c_oracle.iterative_LS_triangulation),
                     (2, c_oracle.linear_eigen_triangulation)):
        xo, so = fn(u, P)
        xh, sh, okh = host_tri(host_math, kind, u, P)
        good = assert_excluded_explained(fn, u, P, xo, so if kind else None, xh, (sh if kind == 1 else okh) if kind else None)
        assert good.mean() > 0.99
        assert np.max(rel_err(xh[good], xo[good])) < 1e-5          # the parity bar (north_star)
        assert np.median(rel_err(xh[good], xo[good])) < 1e-11      # what fp64 actually delivers
        if kind == 1:
            np.testing.assert_array_equal(sh[good], so[good])
        if kind == 2:
            np.testing.assert_array_equal(okh[good], so[good])


def test_rank_deficient_minimum_norm(host_math, c_oracle):
    """Points on the common optical axis of two forward-displaced cameras: A has rank 2; both the
    oracle (SVD cut) and the device arithmetic (eigen pseudo-inverse) return the minimum-norm x."""
    P = np.stack([np.concatenate([np.eye(3), [[0], [0], [40.0]]], axis=1),
                  np.concatenate([np.eye(3), [[0], [0], [28.0]]], axis=1)])
    u = np.zeros((2, 5, 2))
    xo, _ = c_oracle.linear_LS_triangulation(u, P)
    xh, _, _ = host_tri(host_math, 0, u, P)
    np.testing.assert_allclose(xh, xo, atol=1e-9)
    assert np.all(np.isfinite(xh))


# ---------------------------------------------------------------------------------------
# BA arithmetic (csrc/ba_math.h) compiled for the host vs the numpy oracle
# ---------------------------------------------------------------------------------------
def _p(a, t=ctypes.c_double):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


@pytest.mark.parametrize("C,kw", [(2, {}), (3, dict(distortion=True)), (4, dict(behind=2, masked_frac=0.3)),
                                  (5, dict(distortion=True, masked_frac=0.2)), (6, {})])
def test_device_ba_math_vs_oracle(C, kw, host_math):
    from oracle import ba_np
    from ba_util import make_scene
    N, lam = 37, 1e-3
    sc = make_scene(N, C, seed=11 + C, **kw)
    S, g, cost, nv, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"], sc["mask"],
                                             sc["prior_w"], sc["prior_xyz"], lam)
    n6 = 6 * C
    out = np.zeros(n6 * n6 + n6 + 2)
    arrs = [np.ascontiguousarray(sc[k]) for k in ("poses", "calib", "sigma", "points", "obs")]
    mask = None if sc["mask"] is None else np.ascontiguousarray(sc["mask"], dtype=np.uint8)
    rc = host_math.host_ba_linearize(*[_p(a) for a in arrs[:3]], C, _p(arrs[3]), _p(arrs[4]), _p(mask, ctypes.c_uint8),
                                     _p(sc["prior_w"]), _p(sc["prior_xyz"]), ctypes.c_int64(N), ctypes.c_double(lam), _p(out))
    assert rc == 0
    Sh, gh = out[:n6 * n6].reshape(n6, n6), out[n6 * n6:n6 * n6 + n6]
    np.testing.assert_allclose(Sh, S, rtol=0, atol=1e-9 * np.abs(S).max())
    np.testing.assert_allclose(gh, g, rtol=0, atol=1e-9 * np.abs(g).max())
    np.testing.assert_allclose(out[-2], cost, rtol=1e-12)
    assert out[-1] == nv
    np.testing.assert_array_equal(Sh, Sh.T)
    dpose = 1e-3 * np.random.default_rng(C).standard_normal(n6)
    pts_new = np.empty((N, 3))
    rc = host_math.host_ba_backsub(*[_p(a) for a in arrs[:3]], C, _p(arrs[3]), _p(arrs[4]), _p(mask, ctypes.c_uint8),
                                   _p(sc["prior_w"]), _p(sc["prior_xyz"]), ctypes.c_int64(N), ctypes.c_double(lam),
                                   _p(dpose), _p(pts_new))
    assert rc == 0
    np.testing.assert_allclose(pts_new - sc["points"], ba_np.backsub(pieces, dpose), rtol=0, atol=1e-9)


def test_unstable_points_are_explained_not_just_counted(c_oracle):
    """The helper that guards the parity tests' exclusions, on a scene built to have unstable points: landmarks almost on the
    baseline of two cameras (rank-adjacent DLT systems).  A result equal to the perturbed oracle passes; a result that is
    wrong on a WELL-conditioned point, or by more than the oracle's own sensitivity on an unstable one, fails."""
    rng = np.random.default_rng(3)
    P = np.stack([np.concatenate([np.eye(3), [[0.0], [0.0], [40.0]]], axis=1),
                  np.concatenate([np.eye(3), [[-12.0], [0.0], [40.0]]], axis=1)])
    pts = rng.uniform(-3, 3, (400, 3))
    pts[:12] = np.array([[-200.0, 0.0, -40.0]]) + 1e-7 * rng.standard_normal((12, 3))      # on the baseline: depth ~ 0 in both views
    u = np.stack([(pts @ P[c, :, :3].T + P[c, :, 3])[:, :2] / (pts @ P[c, :, :3].T + P[c, :, 3])[:, 2:3] for c in range(2)])
    fn = c_oracle.linear_LS_triangulation
    xo, _ = fn(u, P)
    good, xp, _ = stable_mask(fn, u, P, xo, None, return_perturbed=True)
    assert 0 < (~good).sum() <= 12 and good[12:].all()
    assert_excluded_explained(fn, u, P, xo, None, xp, None, max_frac=0.05)                 # inside the oracle's own sensitivity
    wrong = xo.copy()
    wrong[np.flatnonzero(~good)[0]] += 1e6 * (1.0 + np.abs(xo[np.flatnonzero(~good)[0]]))
    with pytest.raises(AssertionError):
        assert_excluded_explained(fn, u, P, xo, None, wrong, None, max_frac=0.05)
    with pytest.raises(AssertionError):
        assert_excluded_explained(fn, u, P, xo, None, xp, None, max_frac=1e-4)             # and the count is still bounded
