"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's
synthetic triangulation harness, enough of it to reproduce cells of its committed
known-answer files test_3.mat (all three noise types) and test_1and2.mat (k1 = 0.3).

Follows Work/triangulation_comparison/triangulation_comparison.py (paths relative to
/root/reference):
  finite_3D_points            :21-34
  Camera.camera_intrinsics    :91-107
  Camera.camera_pose          :109-125   (cv2.Rodrigues((0,a,0)) == Rot_y(a), restated)
  Camera.project_points       :129-147   (cv2.projectPoints with zero rvec/tvec: pinhole map
                                          f*(x', y') + c after the published OpenCV 2.4 radial /
                                          tangential model -- OpenCV is not vendored)
  Camera.apply_noise          :149-162
  Camera.normalized_points    :164-173   (k1 == 0 shortcut; else cv2.undistortPoints = 5
                                          fixed-point iterations in OpenCV 2.4)
  error_vectors_3D, error_rms :179-188, 205-217
  robustness_stat             :242-260
  cam_trajectory/trajectories :323-401
  test_1and2                  :403-515
  test_3                      :517-627
"""
from math import asin
import numpy as np

RSEED = 123456789            # :370
POSE_OFFSET = 40.0           # default_params["cam_pose_offset"] :276
RESOLUTION = (640, 480)      # :274
NUM_POSES = 40               # :382
MAX_SIDEWAYS = 12.0          # :383
MAX_TOWARDS = 12.0           # :384
THRESH_MAX = 1.0 ** 2        # :372
THRESH_MIN = 1.0 ** 2        # :373


def finite_3D_points(r=4):
    """:21-34 -- integer grid inside the radius-r ball, homogeneous, x outer / z inner."""
    return np.array([(x, y, z, 1.0) for x in range(-r, r + 1)
                     for y in range(-r, r + 1)
                     for z in range(-r, r + 1)
                     if (x * x + y * y + z * z) <= r * r])


def distort_normalized(x, y, k1=0.0, k2=0.0, p1=0.0, p2=0.0):
    """Published OpenCV 2.4 projectPoints distortion model (k1, k2, p1, p2), as used through
    cv2.projectPoints at :140-141 with dist_coeffs = [k1, 0, 0, 0]."""
    r2 = x * x + y * y
    g = 1 + k1 * r2 + k2 * r2 * r2
    return (x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x),
            y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y)


def undistort_normalized(xd, yd, k1=0.0, k2=0.0, p1=0.0, p2=0.0, iters=5):
    """Published OpenCV 2.4 cvUndistortPoints: `iters` (= 5 in 2.4.x) fixed-point iterations of
    x <- (x0 - delta(x)) / (1 + k1 r^2 + k2 r^4), used through cv2.undistortPoints at :173."""
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + k1 * r2 + k2 * r2 * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) * icdist
        y = (yd - dy) * icdist
    return x, y


class Camera:
    def __init__(self, resolution=RESOLUTION, k1=0.0):
        self.f = float(min(resolution))                      # :97
        self.c = np.array(resolution) / 2.0                  # :98
        self.k1 = k1                                         # dist_coeffs = [k1, 0, 0, 0]  :104-105
        self.P = None

    def camera_pose(self, offset, sideways=0.0, towards=0.0, angle=0.0):
        """:109-125."""
        ca, sa = np.cos(angle), np.sin(angle)
        R = np.array([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])   # Rodrigues((0,angle,0))
        centre = np.array([sideways, 0.0, -offset + towards])
        t = -R.dot(centre)
        self.P = np.concatenate([R, t.reshape(3, 1)], axis=1)
        return self

    def project_points(self, points_3D):
        """:129-147 with k1 == 0."""
        q = self.P.dot(points_3D.T).T
        x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
        if self.k1:
            x, y = distort_normalized(x, y, self.k1)
        self.points_2D_exact = self.f * np.stack([x, y], axis=1) + self.c
        return self.points_2D_exact

    def apply_noise(self, sigma, discretized):
        """:149-162 -- uses the legacy global numpy RNG exactly like the reference."""
        if sigma:
            p = self.points_2D_exact + np.random.normal(0, sigma, self.points_2D_exact.shape)
        else:
            p = self.points_2D_exact
        if discretized:
            p = np.rint(p)
        self.points_2D = p

    def normalized_points(self):
        """:164-173: k1 == 0 shortcut, otherwise cv2.undistortPoints."""
        u = (self.points_2D - self.c) / self.f
        if not self.k1:
            return u
        x, y = undistort_normalized(u[:, 0], u[:, 1], self.k1)
        return np.stack([x, y], axis=1)


def last_pose_of_trajectory(k):
    """(sideways, towards, angle) of the last node of trajectory k (0-based), :385-401."""
    if k == 0:
        return MAX_SIDEWAYS, 0.0, 0.0
    if k == 1:
        return 0.0, MAX_TOWARDS, 0.0
    if k == 2:
        return MAX_SIDEWAYS, MAX_TOWARDS, 0.0
    if k == 3:
        a = asin(MAX_SIDEWAYS / POSE_OFFSET)
    elif k == 4:
        a = asin(POSE_OFFSET / POSE_OFFSET)
    else:
        raise ValueError(k)
    return POSE_OFFSET * np.sin(a), POSE_OFFSET * (1 - np.cos(a)), a


def error_rms(error_vectors):
    """:205-217."""
    errors = np.sum(error_vectors ** 2, axis=1)
    return np.sqrt(np.mean(errors)), np.sqrt(np.median(errors)), errors


def robustness_stat(errors, statuses):
    """:242-260."""
    positives_max = errors <= THRESH_MAX
    positives_min = errors <= THRESH_MIN
    positives_est = statuses > 0
    false_pos = np.logical_and(~positives_max, positives_est)
    false_neg = np.logical_and(positives_min, ~positives_est)
    return np.mean(false_pos), np.mean(false_neg)


def test_3_cell(traj, noise_type, sigma, methods, num_trials=100):
    """
    One (trajectory, noise type, sigma) cell of test_3 (:566-603) for the given triangulators.
    `methods` is a list of callables (u (2,N,2), P (2,3,4)) -> (x, status).
    Returns a list of (err3D_mean, err3D_median, false_pos, false_neg) per method.
    """
    points_3D = finite_3D_points(4)
    k1 = 0.3 if noise_type == 2 else 0.0                     # params["cam_k1"] :275, :555-565
    cam1 = Camera(k1=k1).camera_pose(POSE_OFFSET, 0.0, 0.0, 0.0)
    cam2 = Camera(k1=k1).camera_pose(POSE_OFFSET, *last_pose_of_trajectory(traj))
    cam1.project_points(points_3D)
    cam2.project_points(points_3D)
    P = np.stack([cam1.P, cam2.P])
    errs = [[] for _ in methods]
    stats = [[] for _ in methods]
    state = np.random.get_state()
    try:
        np.random.seed(RSEED)                                   # reset_random() :576
        for _ in range(num_trials):
            cam1.apply_noise(sigma, noise_type >= 1)            # :578
            cam2.apply_noise(sigma, noise_type >= 1)            # :579
            u = np.stack([cam1.normalized_points(), cam2.normalized_points()])
            for m, method in enumerate(methods):
                x, status = method(u, P)
                errs[m].append(np.asarray(x, dtype=np.float64) - points_3D[:, 0:3])
                stats[m].append(np.asarray(status))
    finally:
        np.random.set_state(state)
    out = []
    for m in range(len(methods)):
        mean, median, errors = error_rms(np.concatenate(errs[m]))
        fpos, fneg = robustness_stat(errors, np.concatenate(stats[m]))
        out.append((mean, median, fpos, fneg))
    return out


def trajectory_pose(traj, pose_idx, num_poses=NUM_POSES):
    """(sideways, towards, angle) of node `pose_idx` of trajectory `traj` (0-based), :323-401."""
    lin = lambda a, b: np.linspace(a, b, num_poses)[pose_idx]
    if traj == 0:
        return lin(0, MAX_SIDEWAYS), 0.0, 0.0
    if traj == 1:
        return 0.0, lin(0, MAX_TOWARDS), 0.0
    if traj == 2:
        return MAX_SIDEWAYS, lin(0, MAX_TOWARDS), 0.0
    if traj == 3:
        a = lin(asin(0.0 / POSE_OFFSET), asin(MAX_SIDEWAYS / POSE_OFFSET))
    elif traj == 4:
        a = lin(asin(MAX_SIDEWAYS / POSE_OFFSET), asin(POSE_OFFSET / POSE_OFFSET))
    else:
        raise ValueError(traj)
    return POSE_OFFSET * np.sin(a), POSE_OFFSET * (1 - np.cos(a)), a


def test_1and2_cell(traj, pose_idx, methods, num_trials=100, sigma=0.8, k1=0.3):
    """One (trajectory, pose) cell of test_1and2 (:434-476): k1 = 0.3, sigma 0.8 px, discretised."""
    points_3D = finite_3D_points(4)
    cam1 = Camera(k1=k1).camera_pose(POSE_OFFSET, 0.0, 0.0, 0.0)
    cam2 = Camera(k1=k1).camera_pose(POSE_OFFSET, *trajectory_pose(traj, pose_idx))
    cam1.project_points(points_3D)
    cam2.project_points(points_3D)
    P = np.stack([cam1.P, cam2.P])
    errs = [[] for _ in methods]
    stats = [[] for _ in methods]
    state = np.random.get_state()
    try:
        np.random.seed(RSEED)                                   # reset_random() :446
        for _ in range(num_trials):
            cam1.apply_noise(sigma, True)
            cam2.apply_noise(sigma, True)
            u = np.stack([cam1.normalized_points(), cam2.normalized_points()])
            for m, method in enumerate(methods):
                x, status = method(u, P)
                errs[m].append(np.asarray(x, dtype=np.float64) - points_3D[:, 0:3])
                stats[m].append(np.asarray(status))
    finally:
        np.random.set_state(state)
    out = []
    for m in range(len(methods)):
        mean, median, errors = error_rms(np.concatenate(errs[m]))
        fpos, fneg = robustness_stat(errors, np.concatenate(stats[m]))
        out.append((mean, median, fpos, fneg))
    return out
