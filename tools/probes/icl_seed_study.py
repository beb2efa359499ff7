#!/usr/bin/env python3
"""The reference's example sequence (ICL-NUIM lr kt3, its 200 committed frames) through the device loop over RANSAC seeds: plain,
with the resident adjuster per keyframe at this build's noise values, and at the reference's own (BA_info.noise.*-slam2.txt).
    python tools/probes/icl_seed_study.py [frames=200] [seeds=8]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import run_icl_nuim

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
keys = ("ours_vs_groundtruth_rmse_m", "ours_vs_groundtruth_max_m", "ours_vs_reference_rmse_m", "online_vs_groundtruth_rmse_m", "keyframes", "accepted", "frames_per_s",
        "landmarks_screened_out")
out = {"frames": frames, "runs": {}}
for name, kw in (("plain", {}), ("ba_per_keyframe", {"bundle_adjust": "keyframe"}), ("ba_per_keyframe_reference_noise", {"bundle_adjust": "keyframe", "noise": "reference"})):
    rows = []
    for seed in range(seeds):
        r = run_icl_nuim.run(frames, seed=seed, **kw)
        rows.append({k: r[k] for k in keys if k in r})
        rows[-1]["seed"] = seed
        ref = r["reference_vs_groundtruth_rmse_m"]
    e = np.array([x["ours_vs_groundtruth_rmse_m"] for x in rows])
    out["runs"][name] = {"rmse_vs_groundtruth_mm": {"min": round(1e3 * e.min(), 2), "median": round(1e3 * float(np.median(e)), 2), "max": round(1e3 * e.max(), 2)},
                         "frames_per_s_median": float(np.median([x["frames_per_s"] for x in rows])), "per_seed": rows}
out["reference_vs_groundtruth_rmse_mm"] = round(1e3 * ref, 2)
print(json.dumps(out))
