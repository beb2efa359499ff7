#!/usr/bin/env python3
"""Round 6: the residual screen behind every K iterations of an adjustment (mqs_slam_ba_params.screen_iterations) -- accuracy, frames/s
and Levenberg-Marquardt trials per run on the reference's example sequence (200 frames, many seeds) and on the rendered sequence.
    python tools/probes/screen_legs_study.py [icl seeds=16]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import run_icl_nuim as R
import run_slam_loop as L

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R.run(200, "keyframe", 0)
for K in (0, 2, 3, 4, 5):
    runs = [R.run(200, "keyframe", seed, screen_iterations=K) for seed in range(seeds)]
    e = [1e3 * r["ours_vs_groundtruth_rmse_m"] for r in runs]
    print(json.dumps({"sequence": "icl-nuim 200", "screen_iterations": K, "seeds": seeds, "rmse_mm": [round(x, 2) for x in e], "rmse_mm_median": round(float(np.median(e)), 2),
                      "rmse_mm_max": round(max(e), 2), "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs])),
                      "landmarks_screened_out": [r["landmarks_screened_out"] for r in runs][:6], "fallbacks": sum(len(r["fallbacks"]) for r in runs)}), flush=True)
for K in (0, 3, 4):
    runs = [R.run(200, "keyframe", seed, screen_iterations=K, window=None) for seed in range(4)]
    print(json.dumps({"sequence": "icl-nuim 200, every frame", "screen_iterations": K, "rmse_mm": [round(1e3 * r["ours_vs_groundtruth_rmse_m"], 2) for r in runs],
                      "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs]))}), flush=True)
for frames in (60, 90):
    for K in (0, 3, 4):
        runs = [L.run_device(frames, bundle_adjust="keyframe", reassociate=True, seed=seed, repeats=2, ba_screen_iterations=K) for seed in range(4)]
        print(json.dumps({"sequence": "rendered %d" % frames, "screen_iterations": K, "rmse_mm": [round(1e3 * r["trajectory_rmse"], 2) for r in runs],
                          "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs]))}), flush=True)
