"""
GPU parity tests proper: the HIP kernels, called through the C ABI (host-pointer entry points
via the reference-shaped Python facade, and device-pointer entry points via torch tensors),
against the oracle on the same seeded inputs; against the reference's golden cells; and at the
benchmark's full size through size-independent properties.
"""
import numpy as np
import pytest

from oracle import harness_np as H
from util import random_scene, rel_err, stable_mask, assert_excluded_explained

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north_star: 1e-5 relative on triangulated 3D points


@pytest.mark.parametrize("cell", [(0, 0, 8), (0, 1, 20), (2, 1, 39), (3, 0, 8), (4, 0, 8), (4, 1, 20)])
def test_gpu_reproduces_reference_golden_cells(cell, golden3, gpu):
    """The reference's own known-answer file, through the drop-in 2-view facade."""
    t = gpu.triangulation
    methods = [lambda u, P: t.linear_eigen_triangulation(u[0], P[0], u[1], P[1]),
               lambda u, P: t.linear_LS_triangulation(u[0], P[0], u[1], P[1]),
               lambda u, P: t.iterative_LS_triangulation(u[0], P[0], u[1], P[1])]
    tr, nt, si = cell
    res = H.test_3_cell(tr, nt, golden3["noise_sigma_values"][si], methods, int(golden3["num_trials"]))
    n = 257 * int(golden3["num_trials"])
    for m in range(3):
        assert res[m][0] == pytest.approx(golden3["err3D_mean_summary"][tr, nt, si, m], rel=1e-8)
        assert res[m][1] == pytest.approx(golden3["err3D_median_summary"][tr, nt, si, m], rel=1e-8)
        assert abs(res[m][2] - golden3["false_pos_summary"][tr, nt, si, m]) <= 3 / n
        assert abs(res[m][3] - golden3["false_neg_summary"][tr, nt, si, m]) <= 3 / n


def _facade_methods(gpu):
    t = gpu.triangulation
    return [lambda u, P: t.linear_eigen_triangulation(u[0], P[0], u[1], P[1]),
            lambda u, P: t.linear_LS_triangulation(u[0], P[0], u[1], P[1]),
            lambda u, P: t.iterative_LS_triangulation(u[0], P[0], u[1], P[1])]


@pytest.mark.parametrize("traj", [0, 2, 3, 4])
def test_gpu_golden_sweep_every_usable_cell_through_the_facade(traj, golden3, gpu):
    """ALL usable cells of the reference's known-answer file test_3.mat (4 trajectories x noise types 0, 1 x 40 sigmas x 3
    methods = 960 method-cells, 25 700 points each) through the drop-in 2-view facade -- one host-pointer call per trial and
    method, exactly as the reference's harness calls the extension (triangulation_comparison.py:590).  The k1 = 0.3 tier
    (noise type 2) runs in tests/test_camera.py and tests/test_triangulation_comparison.py."""
    methods = _facade_methods(gpu)
    nt_trials = int(golden3["num_trials"])
    n = 257 * nt_trials
    for nt in (0, 1):
        for si in range(40):
            res = H.test_3_cell(traj, nt, golden3["noise_sigma_values"][si], methods, nt_trials)
            for m in range(3):
                where = (traj, nt, si, m)
                assert res[m][0] == pytest.approx(golden3["err3D_mean_summary"][traj, nt, si, m], rel=1e-8), where
                assert res[m][1] == pytest.approx(golden3["err3D_median_summary"][traj, nt, si, m], rel=1e-8), where
                # <= 3 status flips of 25 700 (the .5 absorbs the rounding of the difference of two fractions)
                assert abs(res[m][2] - golden3["false_pos_summary"][traj, nt, si, m]) <= 3.5 / n, where
                assert abs(res[m][3] - golden3["false_neg_summary"][traj, nt, si, m]) <= 3.5 / n, where


def test_gpu_golden_sweep_rank_deficient_trajectory(golden3, gpu):
    """Trajectory 1 ("towards"): the landmarks on the common optical axis give rank-deficient systems whose answer depends on
    the SVD implementation (the reference's own file holds NaN means for linear_eigen there): medians of every cell, 2 %."""
    methods = _facade_methods(gpu)
    for nt in (0, 1):
        for si in range(0, 40, 3):
            res = H.test_3_cell(1, nt, golden3["noise_sigma_values"][si], methods, int(golden3["num_trials"]))
            for m in range(3):
                assert res[m][1] == pytest.approx(golden3["err3D_median_summary"][1, nt, si, m], rel=0.02), (nt, si, m)


@pytest.mark.parametrize("traj", [0, 2, 3, 4])
def test_gpu_golden_sweep_test_1and2_through_the_facade(traj, gpu):
    """Every pose with a baseline of the reference's test_1and2.mat (k1 = 0.3, sigma 0.8 px, discretised; 40 poses per
    trajectory) through the 2-view facade; the undistortion of the observations is the oracle harness's (pinned on the
    same file in tests/test_camera.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_1and2_golden.npz"))
    methods = _facade_methods(gpu)
    nt_trials = int(g["num_trials"])
    for pi in range(1 if traj in (0, 3) else 0, 40):         # first pose of trajectories 0 and 3: the cameras coincide
        res = H.test_1and2_cell(traj, pi, methods, nt_trials)
        for m in range(3):
            where = (traj, pi, m)
            sound = g["err3D_median_summary"][traj, pi, m] < 2.0
            if g["err3D_mean_summary"][traj, pi, m] < 100.0:   # tiny baselines: the mean is a handful of near-singular points
                assert res[m][0] == pytest.approx(g["err3D_mean_summary"][traj, pi, m], rel=1e-6), where
            assert res[m][1] == pytest.approx(g["err3D_median_summary"][traj, pi, m], rel=1e-6), where
            tol = 6.5 / (257 * nt_trials) if sound else 2e-3
            assert abs(res[m][2] - g["false_pos_summary"][traj, pi, m]) <= tol, where
            assert abs(res[m][3] - g["false_neg_summary"][traj, pi, m]) <= tol, where


@pytest.mark.parametrize("C", [2, 3, 4, 5, 6, 7, 8])
def test_nview_parity_host_abi(C, gpu, c_oracle):
    u, P, _ = random_scene(5000, C, seed=200 + C, behind_frac=0.1)
    tc = gpu.triangulation_c
    for name, fn_gpu, fn_or in (("ls", tc.linear_LS_triangulation_nview, c_oracle.linear_LS_triangulation),
                                ("iter", tc.iterative_LS_triangulation_nview, c_oracle.iterative_LS_triangulation),
                                ("eigen", tc.linear_eigen_triangulation_nview, c_oracle.linear_eigen_triangulation)):
        xo, so = fn_or(u, P)
        xg, sg = fn_gpu(u, P)
        good = assert_excluded_explained(fn_or, u, P, xo, None if name == "ls" else so, xg, None if name == "ls" else sg)
        assert good.mean() > 0.99, name
        assert np.max(rel_err(xg[good], xo[good])) < TOL, name
        assert np.median(rel_err(xg[good], xo[good])) < 1e-11, name
        np.testing.assert_array_equal(np.asarray(sg)[good], np.asarray(so)[good])
        assert xg.dtype == np.float64


@pytest.mark.parametrize("N", [0, 1, 2, 63, 64, 255, 256, 257, 511, 513, 1000])
def test_ragged_sizes(N, gpu, c_oracle):
    """Empty input, single point, and sizes straddling the 256-thread workgroup / LDS transpose."""
    u, P, _ = random_scene(max(N, 1), 2, seed=N, behind_frac=0.0)
    u = u[:, :N]
    t = gpu.triangulation
    x, s = t.iterative_LS_triangulation(u[0], P[0], u[1], P[1])
    assert x.shape == (N, 3) and s.shape == (N,) and s.dtype == np.int32
    x2, s2 = t.linear_LS_triangulation(u[0], P[0], u[1], P[1])
    assert x2.shape == (N, 3) and s2.dtype == bool and s2.all()
    x3, s3 = t.linear_eigen_triangulation(u[0], P[0], u[1], P[1])
    assert x3.shape == (N, 3) and s3.dtype == bool
    if N:
        xo, so = c_oracle.iterative_LS_triangulation(u, P)
        assert np.max(rel_err(x, xo)) < TOL
        np.testing.assert_array_equal(s, so)
        assert np.max(rel_err(x2, c_oracle.linear_LS_triangulation(u, P)[0])) < TOL
        assert np.max(rel_err(x3, c_oracle.linear_eigen_triangulation(u, P)[0])) < TOL


def test_reference_dtype_rules(gpu, c_oracle):
    """float32 u is cast to float64 (triangulation_c/__init__.py:32-33); 4x4 and 3x4 P accepted;
    non-contiguous u accepted; output dtype knob (slam2.py:19 sets float32)."""
    u, P, _ = random_scene(300, 2, seed=3, behind_frac=0.0)
    P4 = np.stack([np.vstack([P[c], [0, 0, 0, 1.0]]) for c in range(2)])
    t = gpu.triangulation
    u32 = u.astype(np.float32)
    x32, s32 = t.iterative_LS_triangulation(u32[0], P4[0], u32[1], P4[1])
    xo, so = c_oracle.iterative_LS_triangulation(u32.astype(np.float64), P)
    assert np.max(rel_err(x32, xo)) < TOL
    np.testing.assert_array_equal(s32, so)
    big = np.zeros((300, 4))
    big[:, ::2] = u[0]
    xs, _ = t.linear_LS_triangulation(big[:, ::2], P[0], u[1], P4[1])       # strided view
    assert np.max(rel_err(xs, c_oracle.linear_LS_triangulation(u, P)[0])) < TOL
    try:
        t.set_triangl_output_dtype(np.float32)
        xf, _ = t.iterative_LS_triangulation(u[0], P[0], u[1], P[1])
        assert xf.dtype == np.float32
    finally:
        t.set_triangl_output_dtype(float)


def test_status_codes_and_tolerance(gpu, c_oracle):
    u, P, _ = random_scene(4000, 2, seed=17, behind_frac=0.3)
    for tol in (3e-5, 1e-9, 1e-2):
        xo, so = c_oracle.iterative_LS_triangulation(u, P, tolerance=tol)
        xg, sg = gpu.triangulation.iterative_LS_triangulation(u[0], P[0], u[1], P[1], tolerance=tol)
        good = assert_excluded_explained(lambda a, b: c_oracle.iterative_LS_triangulation(a, b, tolerance=tol), u, P, xo, so, xg, sg)
        assert good.mean() > 0.99
        np.testing.assert_array_equal(sg[good], so[good])
        assert set(np.unique(sg)).issubset({1, 0, -1, -2, -3})
    assert (sg < 0).any() and (sg == 1).any()


def test_non_finite_inputs_flagged(gpu):
    u, P, _ = random_scene(64, 2, seed=1, behind_frac=0.0)
    u[0, 5] = np.nan
    x, ok = gpu.triangulation.linear_eigen_triangulation(u[0], P[0], u[1], P[1])
    assert not ok[5] and ok[:5].all()
    _, s = gpu.triangulation.iterative_LS_triangulation(u[0], P[0], u[1], P[1])
    assert s[5] == 0                      # NaN depths: not > 0 and not <= 0 (triangulation.c:154-159)


def test_device_pointer_abi_matches_host_abi(gpu):
    import torch
    u, P, _ = random_scene(10000, 4, seed=9)
    tc = gpu.triangulation_c
    ud = torch.from_numpy(u).cuda()
    Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    x1, s1 = tc.iterative_LS_triangulation_nview(u, P)
    x2, s2 = gpu.device.iterative_LS_triangulation(ud, Pd)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x2.cpu().numpy(), x1)          # same kernel, same bits
    np.testing.assert_array_equal(s2.cpu().numpy(), s1)
    np.testing.assert_array_equal(gpu.device.linear_LS_triangulation(ud, Pd).cpu().numpy(),
                                  tc.linear_LS_triangulation_nview(u, P)[0])
    x3, ok3 = gpu.device.linear_eigen_triangulation(ud, Pd)
    x4, ok4 = tc.linear_eigen_triangulation_nview(u, P)
    np.testing.assert_array_equal(x3.cpu().numpy(), x4)
    np.testing.assert_array_equal(ok3.cpu().numpy().astype(bool), ok4)


def test_full_size_properties_1e6x4(gpu, c_oracle):
    """BASELINE configs[1] (1e6 landmarks x 4 cameras) through size-independent properties:
    (a) a random 20k sample equals the oracle; (b) shard invariance: triangulating two halves
    separately gives the same bits as the whole (what the multi-GPU path relies on);
    (c) determinism; (d) the result is the least-squares minimiser: perturbing it increases the
    algebraic residual."""
    import torch
    N, C = 1_000_000, 4
    u, P, pts = gpu.synthetic.triangulation_problem(N, C)
    ud = torch.from_numpy(u).cuda()
    Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    dev = gpu.device
    x_it, s_it = dev.iterative_LS_triangulation(ud, Pd)
    x_ls = dev.linear_LS_triangulation(ud, Pd)
    x_eg, ok_eg = dev.linear_eigen_triangulation(ud, Pd)
    torch.cuda.synchronize()
    x_it_h, s_it_h, x_ls_h, x_eg_h = x_it.cpu().numpy(), s_it.cpu().numpy(), x_ls.cpu().numpy(), x_eg.cpu().numpy()
    assert np.isfinite(x_it_h).all() and np.isfinite(x_ls_h).all() and ok_eg.cpu().numpy().all()
    # (a)
    idx = np.random.default_rng(1).choice(N, 20000, replace=False)
    us = np.ascontiguousarray(u[:, idx])
    for fn, xg, sg in ((c_oracle.iterative_LS_triangulation, x_it_h[idx], s_it_h[idx]),
                       (c_oracle.linear_LS_triangulation, x_ls_h[idx], None),
                       (c_oracle.linear_eigen_triangulation, x_eg_h[idx], None)):
        xo, so = fn(us, P)
        good = assert_excluded_explained(fn, us, P, xo, so if sg is not None else None, xg, sg, max_frac=0.005)
        assert good.mean() > 0.995
        assert np.max(rel_err(xg[good], xo[good])) < TOL
        if sg is not None:
            np.testing.assert_array_equal(sg[good], so[good])
    # (a') the bench step's fused pass (tri_kernel<4, 3>: both least-squares methods in one launch) on the same sample:
    # x_it and the status codes are the stand-alone kernel's bits, x_ls meets the oracle at the parity bar
    f_ls, f_it, f_st = dev.linear_and_iterative_LS_triangulation(ud, Pd)
    torch.cuda.synchronize()
    assert torch.equal(f_it, x_it) and torch.equal(f_st, s_it)
    xo, _ = c_oracle.linear_LS_triangulation(us, P)
    good = assert_excluded_explained(c_oracle.linear_LS_triangulation, us, P, xo, None, f_ls.cpu().numpy()[idx], None, max_frac=0.005)
    assert good.mean() > 0.995 and np.max(rel_err(f_ls.cpu().numpy()[idx][good], xo[good])) < TOL
    assert np.median(rel_err(f_ls.cpu().numpy(), x_ls_h)) < 1e-13
    # (b) + (c)
    h = N // 2 + 37
    xa, sa = dev.iterative_LS_triangulation(ud[:, :h].clone(), Pd)
    xb, sb = dev.iterative_LS_triangulation(ud[:, h:].clone(), Pd)
    x_again, s_again = dev.iterative_LS_triangulation(ud, Pd)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([xa, xb]), x_it) and torch.equal(torch.cat([sa, sb]), s_it)
    assert torch.equal(x_again, x_it) and torch.equal(s_again, s_it)
    # (d) linear-LS minimises |A x - b|^2
    sl = slice(0, 50000)
    A = np.empty((50000, 2 * C, 3)); b = np.empty((50000, 2 * C))
    for c in range(C):
        for k in range(2):
            A[:, 2 * c + k] = u[c, sl, k:k + 1] * P[c, 2, :3] - P[c, k, :3]
            b[:, 2 * c + k] = -(u[c, sl, k] * P[c, 2, 3] - P[c, k, 3])
    r0 = np.sum((np.einsum("nij,nj->ni", A, x_ls_h[sl]) - b) ** 2, axis=1)
    rng = np.random.default_rng(2)
    for _ in range(3):
        xp = x_ls_h[sl] + 1e-4 * rng.standard_normal((50000, 3))
        assert np.all(np.sum((np.einsum("nij,nj->ni", A, xp) - b) ** 2, axis=1) >= r0 * (1 - 1e-12))
    # sanity: the triangulated cloud is close to the true one (noise-limited, not bug-limited)
    assert np.sqrt(np.mean(np.sum((x_ls_h - pts) ** 2, axis=1))) < 0.2


def test_baseline_config0_10k_two_cameras(gpu, c_oracle):
    """BASELINE configs[0]: the scaled synthetic generator at 10 000 landmarks x 2 cameras, all three methods through
    the reference-shaped 2-view facade (the call `method(u1, cam1.P, u2, cam2.P)` of triangulation_comparison.py:468)
    against the C oracle and, for the linear DLT, the numpy restatement of the reference's Python twin."""
    from oracle import triangulation_np as T
    u, P, pts = gpu.synthetic.triangulation_problem(10_000, 2)
    t = gpu.triangulation
    P4 = [np.vstack([P[c], [0, 0, 0, 1.0]]) for c in range(2)]                       # slam2.py:554-555 passes 4x4 matrices
    for name, fn, oracle_fn in (("linear_eigen", t.linear_eigen_triangulation, c_oracle.linear_eigen_triangulation),
                                ("linear_LS", t.linear_LS_triangulation, c_oracle.linear_LS_triangulation),
                                ("iterative_LS", t.iterative_LS_triangulation, c_oracle.iterative_LS_triangulation)):
        x, s = fn(u[0], P4[0], u[1], P4[1])
        xo, so = oracle_fn(u, P)
        assert x.shape == (10_000, 3) and np.max(rel_err(x, xo)) < TOL, name
        good = assert_excluded_explained(oracle_fn, u, P, xo, None if name == "linear_LS" else so, x,
                                         None if name == "linear_LS" else np.asarray(s).astype(np.asarray(so).dtype), max_frac=0.005)
        assert good.mean() > 0.995
        np.testing.assert_array_equal(np.asarray(s)[good], np.asarray(so)[good].astype(np.asarray(s).dtype))
        assert np.median(np.linalg.norm(x - pts[:, :3], axis=1)) < 0.5               # and it is the scene that was generated
    xn, _ = T.linear_eigen_triangulation(u[:, :500], P)
    assert np.max(rel_err(t.linear_eigen_triangulation(u[0, :500], P[0], u[1, :500], P[1])[0], xn)) < TOL
    xl = T.linear_LS_triangulation(u[:, :500], P)
    xl = xl[0] if isinstance(xl, tuple) else xl
    assert np.max(rel_err(t.linear_LS_triangulation(u[0, :500], P[0], u[1, :500], P[1])[0], xl)) < TOL


@pytest.mark.parametrize("C", [2, 4, 7])
def test_fused_linear_and_iterative_pass(C, gpu, c_oracle):
    """mqs_triangulate_ls_and_iterative_dev: one pass, both least-squares methods.  The iterative result and the status
    codes are the stand-alone kernel's bit for bit; linear-LS (the first solve of the iteration, unit weights) equals the
    stand-alone kernel to rounding and the oracle to the parity bar."""
    import torch
    u, P, _ = random_scene(3001, C, seed=77 + C, behind_frac=0.05)
    ud, Pd = torch.from_numpy(u).cuda(), torch.from_numpy(np.ascontiguousarray(P)).cuda()
    D = gpu.device
    x_ls, x_it, st = D.linear_and_iterative_LS_triangulation(ud, Pd)
    xi, si = D.iterative_LS_triangulation(ud, Pd)
    xl = D.linear_LS_triangulation(ud, Pd)
    torch.cuda.synchronize()
    assert torch.equal(x_it, xi) and torch.equal(st, si)
    xo, _ = c_oracle.linear_LS_triangulation(u, P)
    good = assert_excluded_explained(c_oracle.linear_LS_triangulation, u, P, xo, None, x_ls.cpu().numpy(), None)
    assert good.mean() > 0.99
    assert np.max(rel_err(x_ls.cpu().numpy()[good], xo[good])) < TOL
    assert np.median(rel_err(x_ls.cpu().numpy()[good], xl.cpu().numpy()[good])) < 1e-13
    with pytest.raises(RuntimeError):
        D.linear_and_iterative_LS_triangulation(ud, Pd, max_iter=0)


@pytest.mark.parametrize("C", [2, 4, 8])
def test_float32_observations_widened_on_load(C, gpu):
    """mqs_triangulate_f32_dev: float32 observations (what slam2.py passes) widened inside the kernel -- bit for bit what the
    float64 entry points give for the host-widened array (the reference's `u.astype(float64)`, __init__.py:32-33)."""
    import torch
    u, P, _ = random_scene(2049, C, seed=5 + C, behind_frac=0.05)
    u32 = torch.from_numpy(u.astype(np.float32)).cuda()
    Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    wide = u32.double()
    D = gpu.device
    for kind, ref in (("linear_ls", lambda: (D.linear_LS_triangulation(wide, Pd), None)),
                      ("iterative_ls", lambda: D.iterative_LS_triangulation(wide, Pd)),
                      ("linear_eigen", lambda: D.linear_eigen_triangulation(wide, Pd))):
        x, s = D.triangulate_f32(kind, u32, Pd)
        xr, sr = ref()
        torch.cuda.synchronize()
        assert torch.equal(x, xr), kind
        assert (s is None and sr is None) or torch.equal(s, sr), kind
    with pytest.raises(ValueError):
        D.triangulate_f32("linear_ls", wide, Pd)


@pytest.mark.skipif(__import__("os").environ.get("MQS_HUGE", "0") != "1",
                    reason="15 GB / ~200 s on the GPU box: opt-in with MQS_HUGE=1 (passed on MI355X when this test was added)")
def test_beyond_one_pass_of_the_grid(gpu):
    """2^28 + 999 landmarks x 2 cameras: more than the 2^20 workgroups one launch dispatches, so the kernels grid-stride and
    every index is 64-bit.  Slices at the start, across the stride boundary and at the ragged end equal the same slices
    triangulated on their own, bit for bit (device-generated observations: 8.6 GB in, 6.4 GB out)."""
    import torch
    N = (1 << 28) + 999
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.empty((2, N, 2), dtype=torch.float64, device="cuda")
    u.uniform_(-0.4, 0.4, generator=g)
    u[1] += 0.05                                                              # a little disparity between the two views
    P = np.stack([gpu.synthetic.camera_matrix(0.0, 0.0, 0.0), gpu.synthetic.camera_matrix(12.0, 0.0, 0.3)])
    Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    D = gpu.device
    x_it, st = D.iterative_LS_triangulation(u, Pd)
    x_ls = D.linear_LS_triangulation(u, Pd)
    torch.cuda.synchronize()
    for a in (0, (1 << 28) - 500, N - 1200):
        sl = slice(a, a + 1200 if a + 1200 <= N else N)
        us = u[:, sl].contiguous()
        xi, si = D.iterative_LS_triangulation(us, Pd)
        xl = D.linear_LS_triangulation(us, Pd)
        torch.cuda.synchronize()
        assert torch.equal(x_it[sl], xi) and torch.equal(st[sl], si) and torch.equal(x_ls[sl], xl), a
    assert int((st == 1).sum().item()) > 0                                   # random rays: few intersect in front of both
    del u, x_it, x_ls, st
    torch.cuda.empty_cache()
