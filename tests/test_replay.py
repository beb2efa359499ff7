"""
SURVEY.md 8(f) rank 3 (triangulation part): replaying the keyframe triangulation of the reference's SLAM
loop from its recorded tracks reproduces the reference's OWN map -- an output of the real reference
(OpenCV 2.4 + its C kernel) -- to float32 storage precision.  This pins the oracle (CPU) and the GPU path
against real reference output, beyond the synthetic goldens.
"""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SVO = os.path.join(HERE, "golden", "ba_svo")
TOL = 1e-5


def _data(mqs):
    io = mqs.ba_io
    return io.load_data(io.create_filenames(SVO, "slam2", 1), 50)


def _rel(x, ref):
    return np.linalg.norm(x - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1.0)


def test_oracle_replay_reproduces_reference_map(mqs, c_oracle):
    from oracle import harness_np as H

    def oracle_tri(p0, p1, K, dist, P0, P1):
        u = []
        for p in (p0, p1):
            x, y = H.undistort_normalized((p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1], *dist)
            u.append(np.stack([x, y], 1))
        return c_oracle.iterative_LS_triangulation(np.stack(u), np.stack([P0, P1]))

    data = _data(mqs)
    out = mqs.slam_replay.replay_keyframe_triangulation(data, triangulate=oracle_tri)
    done = np.isfinite(out["points"][:, 0])
    assert out["n_triangulated"] == 946 and len(out["keyframes"]) >= 10      # the other 100 are the init points
    assert (out["status"][done] >= 0).all()                                   # slam2.py:589 keeps status >= 0
    rel = _rel(out["points"][done], data.points3D[done])
    assert rel.max() < TOL and np.median(rel) < 2e-7                          # float32 storage: ~6e-8


@pytest.mark.gpu
def test_gpu_replay_reproduces_reference_map(gpu):
    data = _data(gpu)
    out = gpu.slam_replay.replay_keyframe_triangulation(data)
    done = np.isfinite(out["points"][:, 0])
    assert out["n_triangulated"] == 946
    rel = _rel(out["points"][done], data.points3D[done])
    assert rel.max() < TOL and np.median(rel) < 2e-7
    assert (out["status"][done] == 1).all()
    assert max(k[2] for k in out["keyframes"]) <= 300                        # slam2.py:1080-1082 target_amount_keypoints


# ------------------------------------------------------------------------------------------------
# Full per-frame replay (pose -> triangulate -> refined pose -> re-triangulate) against the reference's
# recorded trajectory AND map -- both outputs of the real reference run.
# ------------------------------------------------------------------------------------------------
def _oracle_backends(c_oracle):
    from oracle import harness_np as H, pnp_np

    def tri(p0, p1, K, dist, P0, P1):
        u = []
        for p in (p0, p1):
            x, y = H.undistort_normalized((p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1], *dist)
            u.append(np.stack([x, y], 1))
        return c_oracle.iterative_LS_triangulation(np.stack(u), np.stack([P0, P1]))

    def pnp(X, uv, intr, P0):
        rv, tv, _, _ = pnp_np.solve_pnp(X, uv, intr, pnp_np.rodrigues_inv(P0[:, :3]), P0[:, 3])
        return np.c_[pnp_np.rodrigues(rv), tv]

    return pnp, tri


def _check_replay(data, out, pose_tol_R, pose_tol_t, map_tol):
    F = len(out["poses"])
    rec = np.array([data.poses[0][f][1] for f in range(F)])
    dR = np.abs(out["poses"][:, :9] - rec[:, :9]).max(axis=1)
    dt = np.abs(out["poses"][:, 9:] - rec[:, 9:]).max(axis=1)
    assert dR.max() < pose_tol_R and dt.max() < pose_tol_t, (dR.max(), dt.max())
    done = np.isfinite(out["points"][:, 0])
    assert done.sum() == 1046                                                  # every landmark of the recorded map
    rel = _rel(out["points"][done], data.points3D[done])
    assert rel.max() < map_tol, rel.max()
    return dR, dt, rel


def test_oracle_frame_replay_per_frame(mqs, c_oracle):
    """Each frame restarted from the recorded state: isolates the per-frame arithmetic."""
    data = _data(mqs)
    pnp, tri = _oracle_backends(c_oracle)
    out = mqs.slam_replay.replay_frames(data, solve_pnp=pnp, triangulate=tri, chained=False)
    dR, dt, rel = _check_replay(data, out, 1e-6, 2e-6, 1e-5)                   # measured: 2.7e-7, 6.5e-7, 2.4e-6
    assert np.median(dR) < 1e-7 and np.median(dt) < 3e-7 and np.median(rel) < 2e-7


def test_oracle_frame_replay_chained(mqs, c_oracle):
    """186 frames from the 2-D tracks, the initial map and the first pose only."""
    data = _data(mqs)
    pnp, tri = _oracle_backends(c_oracle)
    out = mqs.slam_replay.replay_frames(data, solve_pnp=pnp, triangulate=tri, chained=True)
    _check_replay(data, out, 5e-5, 1e-3, 1e-4)                                # measured: 9.3e-6, 2.2e-4, 2.4e-5
    assert sum(1 for fr in out["frames"] if fr[2] > 0) >= 170                  # keyframes
    from util import ate_rmse
    gt = [(t, p[9:]) for t, p in mqs.ba_io.load_trajectory(os.path.join(SVO, "traj_groundtruth.txt"))]
    est = [(data.poses[0][f][0], out["poses"][f][9:]) for f in range(len(out["poses"]))]
    assert ate_rmse(est, gt)[0] == pytest.approx(0.395356, abs=2e-3)            # results_ate-slam2.txt


@pytest.mark.gpu
def test_gpu_frame_replay(gpu):
    data = _data(gpu)
    out = gpu.slam_replay.replay_frames(data, chained=False)
    dR, dt, rel = _check_replay(data, out, 1e-6, 2e-6, 1e-5)
    assert np.median(dR) < 1e-7 and np.median(dt) < 3e-7 and np.median(rel) < 2e-7
    out = gpu.slam_replay.replay_frames(data, chained=True)
    _check_replay(data, out, 5e-5, 1e-3, 1e-4)                                # measured: 9.3e-6, 2.2e-4, 2.4e-5
    # the replayed trajectory has the accuracy of the reference's own run against ground truth
    from util import ate_rmse
    gt = [(t, p[9:]) for t, p in gpu.ba_io.load_trajectory(os.path.join(SVO, "traj_groundtruth.txt"))]
    est = [(data.poses[0][f][0], out["poses"][f][9:]) for f in range(len(out["poses"]))]
    assert ate_rmse(est, gt)[0] == pytest.approx(0.395356, abs=2e-3)            # results_ate-slam2.txt


@pytest.mark.gpu
def test_fused_keyframe_step_equals_the_eight_calls(gpu):
    """mqs_keyframe_step (one launch per frame: pose, undistort + triangulate, refined pose, re-triangulate) against the same
    frame done with the separate host-pointer calls, on every frame of the recorded SVO run: identical poses, landmarks and
    status codes (the same device functions in the same summation order), and the replay is faster for it."""
    data = _data(gpu)
    a = gpu.slam_replay.replay_frames(data, chained=False, fused=True)
    b = gpu.slam_replay.replay_frames(data, chained=False, fused=False)
    assert np.abs(a["poses"][1:] - b["poses"][1:]).max() <= 1e-12
    np.testing.assert_array_equal(a["status"], b["status"])
    fin = np.isfinite(b["points"][:, 0])
    np.testing.assert_array_equal(np.isfinite(a["points"][:, 0]), fin)
    assert np.abs(a["points"][fin] - b["points"][fin]).max() <= 1e-12 * np.abs(b["points"][fin]).max()
    assert [fr[2] for fr in a["frames"]] == [fr[2] for fr in b["frames"]]
    # chained: the fused path is what replay_frames uses by default
    c = gpu.slam_replay.replay_frames(data, chained=True)
    _check_replay(data, c, 5e-5, 1e-3, 1e-4)
    # the entry point's contract on one frame: no new tracks -> the pose alone, both poses equal
    cal = data.calibrations[0]
    intr = np.array([cal[0], cal[1], cal[3], cal[4], cal[5], cal[6], cal[7], cal[8], 0.0])
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (40, 3)) + [0, 0, 6.0]
    P = np.concatenate([np.eye(3), [[0.1], [-0.05], [0.2]]], axis=1)
    q = X @ P[:, :3].T + P[:, 3]
    uv = np.stack([intr[0] * q[:, 0] / q[:, 2] + intr[2], intr[1] * q[:, 1] / q[:, 2] + intr[3]], axis=1)
    P1, P2, x, st, info = gpu.pnp.keyframe_step(X, uv, np.concatenate([intr[:4], np.zeros(5)]), np.eye(3, 4))
    assert np.abs(P1 - P).max() < 1e-8 and np.array_equal(P1, P2) and x.shape == (0, 3) and st.shape == (0,) and info[2] == 40
    with pytest.raises(ValueError):
        gpu.pnp.keyframe_step(X[:2], uv[:2], intr, np.eye(3, 4))
