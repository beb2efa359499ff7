"""
The detect / track / match front-end of slam.py (main_loop, :57-226) on the GPU path: `match_OF_based` against the
statement-by-statement restatement in the oracle (bit-exact, incl. masks, duplicates and error ties), and the whole frame
step on a rendered frame pair (FAST key points of the right image matched to the tracked left points)."""
import numpy as np
import pytest

from oracle import matching_np as M


def _tuples(d):
    return {k: (m.queryIdx, m.trainIdx, m.distance) for k, m in d.items()}


def test_oracle_match_OF_based_small_case():
    fast = np.array([[10, 10], [20, 20], [20.5, 20], [50, 50]], dtype=np.float32)
    flow = np.array([[10.2, 10.1], [20.1, 20.0], [10.0, 10.3], [50.0, 51.0], [90, 90]], dtype=np.float32)
    err = np.array([3.0, 1.0, 2.0, 20.0, 1.0], dtype=np.float32)
    st = np.array([1, 1, 1, 1, 0])
    best = M.match_OF_based(flow, fast, err, st, 2.0, 0.7)
    # queries 0 and 2 both reach FAST point 0: the lower flow error (query 2) wins; query 1 passes the ratio test
    # (0.1 / 0.4 = 0.25 < 0.7); query 3 has err >= 12; query 4 has status 0
    assert _tuples(best).keys() == {0, 1} and best[0].queryIdx == 2 and best[1].queryIdx == 1
    masked = M.match_OF_based(flow, fast, err, st, 2.0, 0.7, left_point_idxs={0, 1})
    assert masked[0].queryIdx == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_match_OF_based_equals_reference_statements(seed, gpu):
    rng = np.random.default_rng(seed)
    fast = (np.rint(rng.uniform(0, 200, (600, 2)) * 2) / 2).astype(np.float32)
    src = rng.integers(0, 600, 500)
    flow = (fast[src] + np.rint(rng.normal(0, 0.8, (500, 2)) * 4) / 4).astype(np.float32)
    err = rng.integers(0, 30, 500).astype(np.float32) / 2                       # ties and values beyond max_OF_error
    st = (rng.random(500) > 0.1).astype(np.uint8)
    F = gpu.slam_frontend
    for radius, ratio, mask in ((2.0, 0.7, None), (4.0, 1.0, set(range(0, 500, 3))), (1.0, 0.5, None)):
        got = F.match_OF_based(flow, fast, err, st.reshape(-1, 1), radius, ratio, mask)
        ref = M.match_OF_based(flow, fast, err, st, radius, ratio, mask)
        assert _tuples(got) == _tuples(ref) and len(ref) > 20


@pytest.mark.gpu
def test_frame_step_on_rendered_pair(gpu):
    seq = gpu.synthetic.PlaneSequence(frames=40)
    left, right = seq.render(3), seq.render(4)
    F = gpu.slam_frontend
    left_pts, _ = gpu.features.FastFeatureDetector().detect_arrays(left)
    assert len(left_pts) > 200
    tri = set(range(0, len(left_pts), 2))
    fast_pts, matches, (m_tri, m_non), mean_flow, new_tri, cb = F.main_loop(left_pts, left, right, tri)
    assert len(matches) > 0.3 * len(left_pts) and new_tri == set(matches) and cb is None
    assert len(m_tri) + len(m_non) == len(matches) and all(m.queryIdx in tri for m in m_tri)
    # each FAST point is used once, every match lies within the FAST radius of the flow point, and the matched
    # displacements agree with the mean flow (a smooth camera motion over a plane; the few mis-tracked points that land on
    # some FAST point are what the reference leaves to a later epipolar filter, slam.py:221)
    assert len({m.trainIdx for m in matches.values()}) == len(matches)
    disp = np.array([fast_pts[m.trainIdx] - left_pts[m.queryIdx] for m in matches.values()])
    assert np.percentile(np.linalg.norm(disp - mean_flow, axis=1), 95) < 4.0 and np.linalg.norm(mean_flow) > 0.5
    assert max(m.distance for m in matches.values()) <= 2.0
