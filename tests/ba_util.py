"""Small BA scenes for the parity tests (same geometry family as the benchmark scene)."""
import numpy as np
from oracle import ba_np


def world_to_cam_P_to_pose(P):
    """[R_wc | t_wc] (world->camera 3x4) -> camera-to-world pose12 (R row-major, t)."""
    Rwc, twc = P[:, :3], P[:, 3]
    R = Rwc.T
    t = -Rwc.T @ twc
    return np.concatenate([R.reshape(-1), t])


def make_scene(N, C, seed=0, pixel_sigma=1.0, distortion=False, behind=0, masked_frac=0.0,
               pose_noise=(0.02, 0.1), point_noise=0.05):
    import mqslam_amd
    syn = mqslam_amd.synthetic
    rng = np.random.default_rng(seed)
    P = syn.benchmark_cameras(C)
    pts = syn.ball_points(N, seed=seed + 1)
    calib = np.tile(np.array([480.0, 480.0, 0.0, 320.0, 240.0, 0, 0, 0, 0]), (C, 1))
    if distortion:
        calib[:, 2] = 0.7
        calib[:, 5:] = [0.08, -0.02, 0.001, -0.0015]
        calib[:, 0] = 470.0
    poses_true = np.stack([world_to_cam_P_to_pose(P[c]) for c in range(C)])
    obs = np.empty((C, N, 2))
    for c in range(C):
        for i in range(N):
            uv, _, _, ok = ba_np.project(poses_true[c], calib[c], pts[i])
            obs[c, i] = uv if ok else 0.0
    obs += pixel_sigma * rng.standard_normal(obs.shape)
    poses = np.stack([ba_np.retract_pose(poses_true[c], np.concatenate(
        [pose_noise[0] * rng.standard_normal(3), pose_noise[1] * rng.standard_normal(3)])) for c in range(C)])
    points = pts + point_noise * rng.standard_normal(pts.shape)
    if behind:
        points[:behind, 2] -= 60.0                      # behind cameras 0..2 at least
    mask = None
    if masked_frac > 0:
        mask = (rng.random((C, N)) >= masked_frac).astype(np.uint8)
        mask[:2] = 1                                    # keep every landmark constrained by >= 2 views
    sigma = np.full(C, float(pixel_sigma) if pixel_sigma > 0 else 1.0)
    prior_w = np.zeros(N)
    prior_w[:4] = 1.0 / 0.2 ** 2                         # GenerateData.hpp:123 point3D sigma, first landmarks
    prior_xyz = pts.copy()
    return dict(poses=poses, calib=calib, sigma=sigma, points=points, obs=obs, mask=mask,
                prior_w=prior_w, prior_xyz=prior_xyz, poses_true=poses_true, points_true=pts)
