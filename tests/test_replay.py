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
