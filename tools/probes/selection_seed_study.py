#!/usr/bin/env python3
"""Round 6: which selection becomes the default of the in-loop adjustment -- the reference's example sequence (200 frames) over many
seeds, the rendered sequence at 60 / 90 frames over four, a 400-frame rendered run (more accepted frames than the resident adjuster
holds poses): accuracy, frames/s, the engine of every adjustment.  One JSON line per configuration.
    python tools/probes/selection_seed_study.py [icl seeds=16]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import run_icl_nuim as R
import run_slam_loop as L

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R.run(200, "keyframe", 0, window=3)
for name, kw in (("every frame", {"window": None}), ("window 2", {"window": 2}), ("window 3", {"window": 3}), ("window 4", {"window": 4}), ("window 3, history 6", {"window": 3, "history": 6}),
                 ("window 4, history 8", {"window": 4, "history": 8})):
    runs = [R.run(200, "keyframe", seed, **kw) for seed in range(seeds)]
    e = [1e3 * r["ours_vs_groundtruth_rmse_m"] for r in runs]
    print(json.dumps({"sequence": "icl-nuim 200", "config": name, "seeds": seeds, "rmse_mm": [round(x, 2) for x in e], "rmse_mm_median": round(float(np.median(e)), 2),
                      "rmse_mm_max": round(max(e), 2), "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs])),
                      "poses_in_the_last_adjustment": [r["poses_in_the_last_adjustment"] for r in runs][:4], "fallbacks": sum(len(r["fallbacks"]) for r in runs)}), flush=True)
for frames in (60, 90):
    for name, kw in (("every frame", {"ba_window_keyframes": None}), ("window 3", {"ba_window_keyframes": 3}), ("window 4", {"ba_window_keyframes": 4}), ("window 3, history 6", {"ba_window_keyframes": 3, "ba_history_keyframes": 6})):
        runs = [L.run_device(frames, bundle_adjust="keyframe", reassociate=True, seed=seed, repeats=2, **kw) for seed in range(4)]
        print(json.dumps({"sequence": "rendered %d" % frames, "config": name, "rmse_mm": [round(1e3 * r["trajectory_rmse"], 2) for r in runs],
                          "median_frames_per_s": float(np.median([r["frames_per_s"] for r in runs])), "keyframes": runs[0]["keyframes"],
                          "poses_in_the_last_adjustment": runs[0]["bundle_adjust_per_keyframe"]["last"]["poses"]}), flush=True)
for name, kw in (("every frame asked for: beyond 256 frames every keyframe + the latest frames", {"ba_window_keyframes": None}), ("window 3", {"ba_window_keyframes": 3}), ("window 3, history 6", {"ba_window_keyframes": 3, "ba_history_keyframes": 6})):
    r = L.run_device(400, bundle_adjust="keyframe", reassociate=True, seed=1, keep=True, **kw)
    s = r.pop("slam")
    print(json.dumps({"sequence": "rendered 400", "config": name, "accepted": r["accepted"], "keyframes": r["keyframes"], "rmse_mm": round(1e3 * r["trajectory_rmse"], 2),
                      "frames_per_s": r["frames_per_s"], "engines": sorted(set(x["engine"] for x in s.ba_reports)), "adjustments": len(s.ba_reports),
                      "poses_per_adjustment": [x["poses"] for x in s.ba_reports][::4], "fallbacks": s.ba_fallbacks,
                      "adjust_ms_median": float(np.median([x["adjust_ms"] for x in s.ba_reports]))}), flush=True)
    s.close()
