#!/usr/bin/env python3
"""Round 6: the in-loop adjustment over a SELECTION of the accepted frames (mqs_slam_bundle_adjust_window) on the reference's example
sequence -- accuracy against the exact trajectory and frames/s per (dense window, keyframe history, landmark-prior sigma), four seeds
each, next to the full adjustment and the plain loop.  One JSON line per configuration.
    python tools/probes/icl_selection_study.py [frames] [seeds]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import run_icl_nuim as R

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R.run(frames, "keyframe", 0, window=3)            # warm: code objects, allocations
configs = [("plain", None), ("every frame", {"window": None})]
for kw in (2, 3, 4, 6):
    for sig in (0.02, 0.0):
        configs.append(("window %d, every keyframe in front, landmark prior %g" % (kw, sig), {"window": kw, "window_point_sigma": sig}))
configs += [("window 3, every keyframe, no carry", {"window": 3, "carry": False}),
            ("window 4, history 8", {"window": 4, "history": 8}), ("window 4, history 16", {"window": 4, "history": 16}),
            ("window 10, no history (round 4's form)", {"window": 10, "history": 0})]
for name, kw in configs:
    runs = [R.run(frames, None if kw is None else "keyframe", seed, **(kw or {})) for seed in range(seeds)]
    print(json.dumps({"config": name, "frames": frames,
                      "rmse_mm": [round(1e3 * r["ours_vs_groundtruth_rmse_m"], 2) for r in runs],
                      "frames_per_s": [r["frames_per_s"] for r in runs],
                      "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs])),
                      "poses_in_the_last_adjustment": [r.get("poses_in_the_last_adjustment") for r in runs],
                      "ms_per_adjustment_first_to_last": runs[0].get("ms_per_adjustment_first_to_last"),
                      "engines": runs[0].get("engines"), "fallbacks": sum(len(r.get("fallbacks", [])) for r in runs)}), flush=True)
