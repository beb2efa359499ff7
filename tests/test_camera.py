"""
Camera-model tier: oracle pinned against the reference's k1 = 0.3 golden
cells (test_3.mat noise type 2 and test_1and2.mat), the device arithmetic checked on the host, and
GPU parity of the undistort / project / reprojection-error kernels incl. the golden cells run
end to end (pixels -> undistort kernel -> triangulation kernels).
"""
import ctypes
import numpy as np
import pytest

from oracle import harness_np as H

F64 = ctypes.POINTER(ctypes.c_double)
K480 = np.array([[480.0, 0, 320], [0, 480.0, 240], [0, 0, 1]])


def _check_cell(res, gold4, n, rel=1e-8):
    for m in range(3):
        assert res[m][0] == pytest.approx(gold4[0][m], rel=rel)
        assert res[m][1] == pytest.approx(gold4[1][m], rel=rel)
        assert abs(res[m][2] - gold4[2][m]) <= 3 / n and abs(res[m][3] - gold4[3][m]) <= 3 / n


@pytest.fixture(scope="session")
def golden12():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_1and2_golden.npz"))


KEYS = ("err3D_mean_summary", "err3D_median_summary", "false_pos_summary", "false_neg_summary")


@pytest.mark.parametrize("cell", [(0, 8), (2, 39), (3, 20), (4, 0)])
def test_oracle_reproduces_test3_distortion_tier(cell, golden3, c_oracle):
    tr, si = cell
    methods = [c_oracle.linear_eigen_triangulation, c_oracle.linear_LS_triangulation, c_oracle.iterative_LS_triangulation]
    res = H.test_3_cell(tr, 2, golden3["noise_sigma_values"][si], methods, int(golden3["num_trials"]))
    _check_cell(res, [golden3[k][tr, 2, si] for k in KEYS], 25700)


@pytest.mark.parametrize("cell", [(0, 5), (2, 10), (3, 20), (4, 39), (0, 39), (2, 25)])
def test_oracle_reproduces_test_1and2(cell, golden12, c_oracle):
    tr, pi = cell
    methods = [c_oracle.linear_eigen_triangulation, c_oracle.linear_LS_triangulation, c_oracle.iterative_LS_triangulation]
    res = H.test_1and2_cell(tr, pi, methods, int(golden12["num_trials"]))
    _check_cell(res, [golden12[k][tr, pi] for k in KEYS], 25700)


def _scene(n=2000, seed=0):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-4, 4, (n, 3))
    P = np.array([[0.9, 0.0, 0.4359, 1.0], [0.0, 1.0, 0.0, -0.5], [-0.4359, 0.0, 0.9, 38.0]])
    dist = np.array([0.3, -0.05, 0.002, -0.001, 0.01])
    return pts, P, dist


def test_device_camera_math_on_host(host_math):
    pts, P, dist = _scene()
    intr = np.concatenate([[480.0, 470.0, 320.0, 240.0], dist])
    uv = np.empty((len(pts), 2)); z = np.empty(len(pts))
    host_math.host_project(pts.ctypes.data_as(F64), np.ascontiguousarray(P).ctypes.data_as(F64), intr.ctypes.data_as(F64),
                           ctypes.c_int64(len(pts)), uv.ctypes.data_as(F64), z.ctypes.data_as(F64))
    q = pts @ P[:, :3].T + P[:, 3]
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + dist[0] * r2 + dist[1] * r2 ** 2 + dist[4] * r2 ** 3
    xd = x * g + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
    yd = y * g + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
    np.testing.assert_allclose(uv, np.stack([480 * xd + 320, 470 * yd + 240], 1), rtol=1e-13)
    np.testing.assert_allclose(z, q[:, 2], rtol=1e-14)
    # undistort inverts the (k1-only) model of the reference to the 5-iteration accuracy
    intr1 = np.array([480.0, 480.0, 320.0, 240.0, 0.3, 0, 0, 0, 0])
    xo, yo = H.distort_normalized(x, y, 0.3)
    pix = np.ascontiguousarray(np.stack([480 * xo + 320, 480 * yo + 240], 1))
    out = np.empty_like(pix)
    host_math.host_undistort(pix.ctypes.data_as(F64), intr1.ctypes.data_as(F64), ctypes.c_int64(len(pix)), out.ctypes.data_as(F64))
    xr, yr = H.undistort_normalized((pix[:, 0] - 320) / 480, (pix[:, 1] - 240) / 480, 0.3)
    np.testing.assert_allclose(out, np.stack([xr, yr], 1), rtol=1e-13, atol=1e-15)
    assert np.abs(out - np.stack([x, y], 1)).max() < 1e-4


def test_facade_validation(mqs):
    with pytest.raises(ValueError):
        mqs.camera.undistort_points(np.zeros((3, 3)), K480, [0.3, 0, 0, 0])
    with pytest.raises(ValueError):
        mqs.camera.undistort_points(np.zeros((3, 2)), K480, [0.3, 0, 0])
    np.testing.assert_allclose(mqs.camera.rodrigues([0, 0.3, 0]),
                               [[np.cos(.3), 0, np.sin(.3)], [0, 1, 0], [-np.sin(.3), 0, np.cos(.3)]], atol=1e-15)


@pytest.mark.gpu
def test_gpu_undistort_project_reprojection(gpu):
    pts, P, dist = _scene(5003, seed=3)
    K = np.array([[480.0, 0, 320], [0, 470.0, 240], [0, 0, 1]])
    uv, z, _ = gpu.camera.project_points(pts, K, dist, P)
    q = pts @ P[:, :3].T + P[:, 3]
    np.testing.assert_allclose(z, q[:, 2], rtol=1e-14)
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + dist[0] * r2 + dist[1] * r2 ** 2 + dist[4] * r2 ** 3
    xd = x * g + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
    yd = y * g + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
    np.testing.assert_allclose(uv, np.stack([480 * xd + 320, 470 * yd + 240], 1), rtol=1e-13)
    # reprojection_error with the reference's (rvec, tvec) call shape
    rvec = np.array([0.0, np.arcsin(0.4359), 0.0])
    imgp = uv + np.random.default_rng(0).normal(0, 0.7, uv.shape)
    rms, reproj = gpu.camera.reprojection_error(pts, imgp, K, dist, rvec, P[:, 3])
    Pm = np.concatenate([gpu.camera.rodrigues(rvec), P[:, 3:4]], 1)
    uv2, _, _ = gpu.camera.project_points(pts, K, dist, Pm)
    assert rms == pytest.approx(np.sqrt(((uv2 - imgp) ** 2).sum() / len(imgp)), rel=1e-12)
    np.testing.assert_array_equal(reproj, uv2)
    # undistort: float32 in -> float32 out like cv2; k1-only model against the oracle
    pix = uv.astype(np.float32)
    un32 = gpu.camera.undistort_points(pix, K, dist)
    assert un32.dtype == np.float32 and un32.shape == pix.shape
    K1 = np.array([[480.0, 0, 320], [0, 480.0, 240], [0, 0, 1]])
    xo, yo = H.distort_normalized(x, y, 0.3)
    pix1 = np.stack([480 * xo + 320, 480 * yo + 240], 1)
    un = gpu.camera.undistort_points(pix1, K1, [0.3, 0, 0, 0])
    xr, yr = H.undistort_normalized((pix1[:, 0] - 320) / 480, (pix1[:, 1] - 240) / 480, 0.3)
    np.testing.assert_allclose(un, np.stack([xr, yr], 1), rtol=1e-13, atol=1e-15)
    assert gpu.camera.undistort_points(np.zeros((0, 2)), K1, [0.3, 0, 0, 0]).shape == (0, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("which,cell", [("t3", (0, 8)), ("t3", (3, 20)), ("t12", (2, 10)), ("t12", (4, 39))])
def test_gpu_distortion_tier_golden_cells_end_to_end(which, cell, golden3, golden12, gpu, monkeypatch):
    """pixels -> undistort kernel -> triangulation kernels, against the reference's k1 = 0.3 goldens."""
    t = gpu.triangulation
    methods = [lambda u, P: t.linear_eigen_triangulation(u[0], P[0], u[1], P[1]),
               lambda u, P: t.linear_LS_triangulation(u[0], P[0], u[1], P[1]),
               lambda u, P: t.iterative_LS_triangulation(u[0], P[0], u[1], P[1])]

    def gpu_normalized(self):                               # triangulation_comparison.py:164-173 on the GPU
        return gpu.camera.undistort_points(self.points_2D, K480, [self.k1, 0.0, 0.0, 0.0])

    monkeypatch.setattr(H.Camera, "normalized_points", gpu_normalized)
    if which == "t3":
        tr, si = cell
        res = H.test_3_cell(tr, 2, golden3["noise_sigma_values"][si], methods, int(golden3["num_trials"]))
        _check_cell(res, [golden3[k][tr, 2, si] for k in KEYS], 25700)
    else:
        tr, pi = cell
        res = H.test_1and2_cell(tr, pi, methods, int(golden12["num_trials"]))
        _check_cell(res, [golden12[k][tr, pi] for k in KEYS], 25700)


@pytest.mark.gpu
def test_fused_pixels_triangulation_equals_two_step(gpu):
    """pixels -> (undistort on load) -> triangulate in ONE kernel == undistort kernel then triangulation kernel, bitwise."""
    import torch
    N, C = 20011, 4
    u, P, pts = gpu.synthetic.triangulation_problem(N, C)
    intr = np.tile(np.array([480.0, 470.0, 320.0, 240.0, 0.3, -0.05, 0.002, -0.001, 0.01]), (C, 1))
    intr[:, 0] += np.arange(C)
    xd = np.empty_like(u)
    for c in range(C):                                    # distort the normalised observations, then to pixels
        x, y = u[c, :, 0], u[c, :, 1]
        r2 = x * x + y * y
        g = 1 + intr[c, 4] * r2 + intr[c, 5] * r2 ** 2 + intr[c, 8] * r2 ** 3
        xd[c, :, 0] = (x * g + 2 * intr[c, 6] * x * y + intr[c, 7] * (r2 + 2 * x * x)) * intr[c, 0] + intr[c, 2]
        xd[c, :, 1] = (y * g + intr[c, 6] * (r2 + 2 * y * y) + 2 * intr[c, 7] * x * y) * intr[c, 1] + intr[c, 3]
    pix = torch.from_numpy(xd).cuda()
    Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    intr_d = torch.from_numpy(intr).cuda()
    un = torch.empty_like(pix)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for c in range(C):
        gpu._lib.check(gpu._lib.lib().mqs_undistort_points_dev(pix[c].data_ptr(), intr_d[c].data_ptr(), N, un[c].data_ptr(), sp))
    D = gpu.device
    for kind, two_step in (("linear_ls", lambda: (D.linear_LS_triangulation(un, Pd), None)),
                           ("iterative_ls", lambda: D.iterative_LS_triangulation(un, Pd)),
                           ("linear_eigen", lambda: D.linear_eigen_triangulation(un, Pd))):
        x2, s2 = two_step()
        x1, s1 = D.triangulate_pixels(kind, pix, intr_d, Pd)
        torch.cuda.synchronize()
        assert torch.equal(x1, x2)
        if s2 is not None:
            assert torch.equal(s1, s2)
    # and the fused result is the right answer: close to the scene (noise-limited)
    assert float((x1.cpu() - torch.from_numpy(pts)).norm(dim=1).median()) < 0.2
