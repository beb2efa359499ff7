#!/usr/bin/env python3
"""The resident in-loop adjuster (mqs_slam_bundle_adjust) against its host-built twin on every keyframe of the rendered sequence:
prints per adjustment the largest pose / landmark difference and both reports."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import run_slam_loop

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = run_slam_loop.run_device(frames, bundle_adjust="keyframe", keep=True, ba_check=True)
slam = out.pop("slam")
rows = []
for c in slam.ba_checks:
    dp = float(np.abs(c["host_poses"] - c["device_poses"]).max())
    n = min(len(c["host_points"]), len(c["device_points"]))
    dx = float(np.abs(c["host_points"][:n] - c["device_points"][:n]).max())
    hr, dr = c["host_report"], c["device_report"]
    nr = min(len(c["host_retired"]), len(c["device_retired"]))
    rows.append({"frame": c["frame"], "max_pose_diff": dp, "max_point_diff": dx,
                 "retired_differs": int((c["host_retired"][:nr] != c["device_retired"][:nr]).sum()),
                 "host": {k: hr[k] for k in ("poses", "landmarks", "landmarks_adjusted", "observations", "passes", "landmarks_screened_out", "lm_iterations", "cost_before", "cost_after", "adjust_ms")},
                 "device": {k: dr[k] for k in ("poses", "landmarks", "landmarks_adjusted", "observations", "passes", "landmarks_screened_out", "lm_iterations", "cost_before", "cost_after", "lm_trials", "grid_barriers", "adjust_ms")}})
    print(json.dumps(rows[-1]))
slam.close()
print(json.dumps(out))
