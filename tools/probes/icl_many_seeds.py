#!/usr/bin/env python3
"""The reference's example sequence (80 frames) through the device loop over many RANSAC seeds: the distribution of its distance
from the reference's committed trajectory and from the exact one, plain and with the bundle adjustment per keyframe."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, run_icl_nuim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = {}
for name, kw in (("plain", {}), ("ba_per_keyframe", {"bundle_adjust": "keyframe"}), ("plain_with_second_pass_screen_1px", {"screen": 1.0})):
    rows = [run_icl_nuim.run(80, seed=s, **kw) for s in range(n)]
    q = lambda key, f=lambda r: r: [round(float(v), 5) for v in np.percentile([f(r)[key] if f is not None else r[key] for r in rows], [0, 25, 50, 75, 100])]
    out[name] = {"seeds": n, "accepted_every_frame": all(r["accepted"] == 80 for r in rows),
                 "ours_vs_reference_rmse_m_min_q1_median_q3_max": q("ours_vs_reference_rmse_m"),
                 "ours_vs_groundtruth_rmse_m_min_q1_median_q3_max": q("ours_vs_groundtruth_rmse_m"),
                 "orientation_ours_vs_groundtruth_deg_min_q1_median_q3_max": q("ours_vs_groundtruth", lambda r: r["orientation_rmse_deg"]),
                 "keyframes_min_max": [min(r["keyframes"] for r in rows), max(r["keyframes"] for r in rows)],
                 "first_keyframe_min_max": [min(r["keyframe_frames"][1] for r in rows), max(r["keyframe_frames"][1] for r in rows)],
                 "per_seed_ours_vs_groundtruth_mm": [round(1e3 * r["ours_vs_groundtruth_rmse_m"], 1) for r in rows]}
out["reference_vs_groundtruth_rmse_m"] = rows[0]["reference_vs_groundtruth_rmse_m"]
out["reference_orientation_vs_groundtruth_deg"] = rows[0]["orientation_rmse_deg"]["reference_vs_groundtruth"]
print(json.dumps(out))
