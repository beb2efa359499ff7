#!/usr/bin/env python3
"""Round 6: the BA loop against the resident adjuster's workgroup count (MQS_SLAM_BA_GROUPS=G python tools/probes/ba_groups_study.py)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
R.run(80, "keyframe")
runs = [R.run(200, "keyframe") for _ in range(3)]
print(json.dumps({"groups": os.environ.get("MQS_SLAM_BA_GROUPS", "default"), "best_frames_per_s": max(r["frames_per_s"] for r in runs),
                  "adjust_ms_median": runs[0].get("adjust_ms_median"), "rmse_mm": round(1e3 * runs[0]["ours_vs_groundtruth_rmse_m"], 3)}))
