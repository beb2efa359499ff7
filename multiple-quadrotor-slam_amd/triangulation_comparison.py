"""
The reference's synthetic triangulation experiment on the gfx950 path -- counterpart of
Work/triangulation_comparison/triangulation_comparison.py (the source of BASELINE configs[0] and of the committed
known-answer files test_1and2.mat / test_3.mat):

    finite_3D_points                       :21-34     the 257 integer points inside the radius-4 ball
    Camera (intrinsics, pose, projection,  :89-173    f = min(resolution), c = resolution / 2, dist = [k1, 0, 0, 0];
            noise, normalisation)                     centre (sideways, 0, -offset + towards), R = Rot_y(angle)
    error_rms, robustness_stat             :205-217, 242-260
    cam_trajectory + the five trajectories :323-401
    test_1and2, test_3                     :403-627   the arrays they store (`err3D_*`, `err2D_*`, `false_*_summary`)

What is different is the execution: the reference triangulates 257 points per call, `num_trials` x methods x cells times
(hours of Python); here ALL trials of a cell are one batch -- the noise of the 100 trials is drawn in the reference's order
from the same legacy generator stream (RandomState(rseed): cam1 then cam2, trial after trial), the 2 x 25 700 pixel
observations go to the device once, and each method is ONE fused launch (undistort + normalise + triangulate,
`mqs_triangulate_pixels_dev`); the exact projections and the reprojection of the computed points use the projection
kernel.  Methods: linear_eigen, linear_LS, iterative_LS (the reference's fourth, polynomial, is outside the hot path:
its column of the result arrays is NaN).  No CPU fallback: without the HIP library every call raises.
"""
from math import asin

import numpy as np

from . import camera
from . import device
from . import synthetic

RSEED = 123456789                       # :370
NUM_TRIALS = 100                        # :371
ROBUSTNESS_THRESH_MAX = 1.0 ** 2        # :372
ROBUSTNESS_THRESH_MIN = 1.0 ** 2        # :373
DEFAULT_PARAMS = {"cam_resolution": (640, 480), "cam_k1": 0.3, "cam_pose_offset": 40.0, "cam_noise_sigma": 0.8,
                  "cam_noise_discretized": True}                                     # :272-279
METHODS = ("linear_eigen_triangulation", "linear_LS_triangulation", "iterative_LS_triangulation", "polynomial_triangulation")
_KIND = {0: "linear_eigen", 1: "linear_ls", 2: "iterative_ls"}


def finite_3D_points(r=4):
    """:21-34 -- homogeneous integer grid points with x^2 + y^2 + z^2 <= r^2 (x outermost, z innermost)."""
    g = np.arange(-r, r + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    keep = (X * X + Y * Y + Z * Z) <= r * r
    return np.stack([X[keep], Y[keep], Z[keep], np.ones(keep.sum())], axis=1).astype(np.float64)


def cam_trajectory(sideways_values, towards_values, angle_values):
    return {"sideways_values": np.asarray(sideways_values, dtype=np.float64),
            "towards_values": np.asarray(towards_values, dtype=np.float64),
            "angle_values": np.asarray(angle_values, dtype=np.float64)}


def trajectories(num_poses=40, max_sideways=12.0, max_towards=12.0, offset=DEFAULT_PARAMS["cam_pose_offset"]):
    """:382-401 -- sideways, towards, both, and two arcs around the scene (angles up to asin(12/40) and pi/2)."""
    zeros = np.zeros(num_poses)
    arc = lambda a0, a1: np.linspace(a0, a1, num_poses)
    out = [cam_trajectory(np.linspace(0, max_sideways, num_poses), zeros, zeros),
           cam_trajectory(zeros, np.linspace(0, max_towards, num_poses), zeros),
           cam_trajectory(np.full(num_poses, max_sideways), np.linspace(0, max_towards, num_poses), zeros)]
    for a in (arc(asin(0.0 / offset), asin(max_sideways / offset)), arc(asin(max_sideways / offset), asin(offset / offset))):
        out.append(cam_trajectory(offset * np.sin(a), offset * (1 - np.cos(a)), a))
    return out


class Camera:
    """:89-173, state kept as arrays; projection through the GPU projection kernel."""

    def __init__(self, resolution=DEFAULT_PARAMS["cam_resolution"], k1=0.0):
        self.camera_intrinsics(resolution, k1)
        self.P = None

    def camera_intrinsics(self, resolution, k1=0.0):
        f, c = float(min(resolution)), np.asarray(resolution, dtype=np.float64) / 2.0
        self.K = np.array([[f, 0.0, c[0]], [0.0, f, c[1]], [0.0, 0.0, 1.0]])
        self.dist = np.array([k1, 0.0, 0.0, 0.0])
        self.intr = np.array([f, f, c[0], c[1], k1, 0.0, 0.0, 0.0, 0.0])
        return self

    def camera_pose(self, offset, sideways=0.0, towards=0.0, angle=0.0):
        self.P = synthetic.camera_matrix(sideways, towards, angle, offset)
        return self

    def project_points(self, points_3D, remember=True):
        uv, _, _ = camera.project_points(np.asarray(points_3D, dtype=np.float64)[:, 0:3], self.K, self.dist, self.P)
        if remember:
            self.points_2D_exact = uv
        return uv


def error_rms(error_vectors):
    """:205-217 -> (rms, root of the median squared error, squared errors)."""
    e = np.sum(np.asarray(error_vectors) ** 2, axis=1)
    return np.sqrt(np.mean(e)), np.sqrt(np.median(e)), e


def robustness_stat(errors, statuses):
    """:242-260 -> (false positive ratio, false negative ratio) of the methods' status values."""
    est = np.asarray(statuses) > 0
    return (float(np.mean(~(errors <= ROBUSTNESS_THRESH_MAX) & est)),
            float(np.mean((errors <= ROBUSTNESS_THRESH_MIN) & ~est)))


def run_cell(points_3D, cam1, cam2, noise_sigma, noise_discretized, num_trials=NUM_TRIALS, rseed=RSEED, methods=(0, 1, 2)):
    """
    One cell of the experiment (the body of :444-476 / :571-603): `num_trials` noisy observation sets, every method on
    all of them.  Returns {method index: (err3D_mean, err3D_median, err2D_mean, err2D_median, false_pos, false_neg)} and
    whether every noisy point of camera 2 stayed inside the image.
    """
    import torch
    N = len(points_3D)
    rng = np.random.RandomState(rseed)                                   # reset_random(): the legacy stream, restarted per cell
    exact = np.stack([cam1.points_2D_exact, cam2.points_2D_exact])       # (2, N, 2)
    if noise_sigma:
        noise = rng.normal(0, noise_sigma, (num_trials, 2, N, 2))        # trial-major, camera 1 before camera 2: the reference's order
        pix = exact[None] + noise
    else:
        pix = np.broadcast_to(exact[None], (num_trials, 2, N, 2)).copy()
    if noise_discretized:
        pix = np.rint(pix)
    res = DEFAULT_PARAMS["cam_resolution"]
    inside = bool((pix[:, 1, :, 0] >= 0).all() and (pix[:, 1, :, 0] < res[0]).all() and
                  (pix[:, 1, :, 1] >= 0).all() and (pix[:, 1, :, 1] < res[1]).all())
    dev = torch.device("cuda", 0)
    pixels = torch.from_numpy(np.ascontiguousarray(pix.transpose(1, 0, 2, 3).reshape(2, num_trials * N, 2))).to(dev)
    intr = torch.from_numpy(np.stack([cam1.intr, cam2.intr])).to(dev)
    P = torch.from_numpy(np.ascontiguousarray(np.stack([cam1.P[0:3], cam2.P[0:3]]))).to(dev)
    truth = np.tile(np.asarray(points_3D)[:, 0:3], (num_trials, 1))
    exact1, exact2 = np.tile(exact[0], (num_trials, 1)), np.tile(exact[1], (num_trials, 1))
    out = {}
    for m in methods:
        x, status = device.triangulate_pixels(_KIND[m], pixels, intr, P)
        x = x.cpu().numpy()
        status = np.ones(len(x), dtype=bool) if status is None else status.cpu().numpy()
        e3 = error_rms(x - truth)
        e2 = error_rms(np.concatenate([cam1.project_points(x, False) - exact1, cam2.project_points(x, False) - exact2]))
        out[m] = (e3[0], e3[1], e2[0], e2[1]) + robustness_stat(e3[2], status)
    return out, inside


_FIELDS = ("err3D_mean_summary", "err3D_median_summary", "err2D_mean_summary", "err2D_median_summary", "false_pos_summary",
           "false_neg_summary")


def test_3(trajs=None, max_noise_sigma=4.0, num_noise_tests=40, num_trials=NUM_TRIALS, noise_ids=None, methods=(0, 1, 2),
           filename=None):
    """:517-627 -- effect of the noise model at the last pose of every trajectory.  Result arrays are indexed
    [trajectory, noise type, noise sigma, method] like the reference's; `noise_ids` restricts the sigma indices."""
    trajs = trajectories() if trajs is None else trajs
    params = DEFAULT_PARAMS
    points_3D = finite_3D_points(4)
    sigmas = np.linspace(0, max_noise_sigma, num_noise_tests)
    arrays = {k: np.full((len(trajs), 3, num_noise_tests, len(METHODS)), np.nan) for k in _FIELDS}
    inside = True
    for ti, tr in enumerate(trajs):
        for nty in range(3):
            discretized, k1 = nty >= 1, (params["cam_k1"] if nty == 2 else 0.0)
            cam1 = Camera(params["cam_resolution"], k1).camera_pose(params["cam_pose_offset"])
            cam2 = Camera(params["cam_resolution"], k1).camera_pose(params["cam_pose_offset"], tr["sideways_values"][-1],
                                                                    tr["towards_values"][-1], tr["angle_values"][-1])
            cam1.project_points(points_3D)
            cam2.project_points(points_3D)
            for ni in (range(num_noise_tests) if noise_ids is None else noise_ids):
                cell, ok = run_cell(points_3D, cam1, cam2, sigmas[ni], discretized, num_trials, methods=methods)
                inside = inside and ok
                for m, vals in cell.items():
                    for k, v in zip(_FIELDS, vals):
                        arrays[k][ti, nty, ni, m] = v
    arrays.update(noise_sigma_values=sigmas, points_3D=points_3D, num_trials=num_trials, rseed=RSEED, is_inside_view=inside,
                  triangl_methods=np.array(METHODS))
    if filename:
        np.savez(filename, **arrays)
    return arrays


def test_1and2(trajs=None, num_trials=NUM_TRIALS, pose_ids=None, methods=(0, 1, 2), filename=None):
    """:403-515 (the per-pose summaries of Test 2) -- effect of the second camera's configuration: k1 = 0.3, sigma 0.8 px,
    discretised.  Arrays are indexed [trajectory, pose, method]."""
    trajs = trajectories() if trajs is None else trajs
    params = DEFAULT_PARAMS
    points_3D = finite_3D_points(4)
    num_poses = len(trajs[0]["sideways_values"])
    arrays = {k: np.full((len(trajs), num_poses, len(METHODS)), np.nan) for k in _FIELDS}
    cam1 = Camera(params["cam_resolution"], params["cam_k1"]).camera_pose(params["cam_pose_offset"])
    cam1.project_points(points_3D)
    inside = True
    for ti, tr in enumerate(trajs):
        for pi in (range(num_poses) if pose_ids is None else pose_ids):
            cam2 = Camera(params["cam_resolution"], params["cam_k1"]).camera_pose(
                params["cam_pose_offset"], tr["sideways_values"][pi], tr["towards_values"][pi], tr["angle_values"][pi])
            cam2.project_points(points_3D)
            cell, ok = run_cell(points_3D, cam1, cam2, params["cam_noise_sigma"], params["cam_noise_discretized"], num_trials,
                                methods=methods)
            inside = inside and ok
            for m, vals in cell.items():
                for k, v in zip(_FIELDS, vals):
                    arrays[k][ti, pi, m] = v
    arrays.update(points_3D=points_3D, num_trials=num_trials, rseed=RSEED, is_inside_view=inside, triangl_methods=np.array(METHODS))
    if filename:
        np.savez(filename, **arrays)
    return arrays
