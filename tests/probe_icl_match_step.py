#!/usr/bin/env python3
"""slam.py's detect -> track -> match frame step on real frames of the reference's example sequence: GPU path against the oracle,
and the matches against the epipolar geometry of the exact poses."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # (under tests/: it uses the oracle as the checker)
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mqslam_amd, run_icl_nuim
from oracle import features_np as Fn, matching_np as Mn
from test_icl_nuim import quat_to_R
seq = np.load(run_icl_nuim.FIX)
K = seq["K"]
for a, b in ((10, 11), (30, 32), (60, 62), (70, 75)):
    I, J = seq["frames"][a], seq["frames"][b]
    left, _ = mqslam_amd.features.FastFeatureDetector().detect_arrays(I)
    right_fast, matches, _, mean_flow, _, _ = mqslam_amd.slam_frontend.main_loop(left, I, J, set())
    flow, st, err = Fn.calc_optical_flow_pyr_lk(I, J, left)
    ref = Mn.match_OF_based(flow, right_fast, err.reshape(-1), st.reshape(-1), 2.0, 0.7)
    got_pairs = set((m.queryIdx, t) for t, m in matches.items()); ref_pairs = set((m.queryIdx, t) for t, m in ref.items())
    Ra, Rb = quat_to_R(seq["traj_groundtruth"][a, 4:8]).T, quat_to_R(seq["traj_groundtruth"][b, 4:8]).T
    ca, cb = seq["traj_groundtruth"][a, 1:4], seq["traj_groundtruth"][b, 1:4]
    R = Rb @ Ra.T; t = Rb @ (ca - cb)
    E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    Fm = np.linalg.inv(K).T @ E @ np.linalg.inv(K)
    gp = sorted(got_pairs)
    x1 = np.c_[np.array([left[i] for i, _ in gp], float), np.ones(len(gp))]; x2 = np.c_[np.array([right_fast[j] for _, j in gp], float), np.ones(len(gp))]
    l2, l1 = x1 @ Fm.T, x2 @ Fm
    sampson = np.abs(np.sum(x2 * l2, axis=1)) / np.sqrt(l2[:, 0] ** 2 + l2[:, 1] ** 2 + l1[:, 0] ** 2 + l1[:, 1] ** 2)
    print(json.dumps({"frames": [a, b], "fast_corners_left_right": [len(left), len(right_fast)], "matches_gpu": len(got_pairs), "matches_oracle": len(ref_pairs),
                      "identical_pairs": len(got_pairs & ref_pairs), "sampson_px_median_p95_max": [round(float(np.median(sampson)), 3), round(float(np.percentile(sampson, 95)), 3), round(float(sampson.max()), 3)],
                      "mean_flow_px": [round(float(v), 2) for v in mean_flow]}))
