import os, sys, json, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, run_icl_nuim, run_slam_loop
t0 = time.time()
out = {"icl200": {}, "rendered90": {}}
for name, kw in (("plain", {}), ("ba", {"bundle_adjust": "keyframe"})):
    rows = [run_icl_nuim.run(200, seed=s, **kw) for s in range(32)]
    out["icl200"][name] = {"seeds": 32, "accepted_min": min(r["accepted"] for r in rows), "rmse_mm_min_med_max": [round(1e3 * float(v), 2) for v in np.percentile([r["ours_vs_groundtruth_rmse_m"] for r in rows], [0, 50, 100])],
                           "fps_median": round(float(np.median([r["frames_per_s"] for r in rows])), 1)}
for name, kw in (("plain", {}), ("ba_reassoc", {"bundle_adjust": "keyframe", "reassociate": True})):
    rows = [run_slam_loop.run_device(90, seed=s, **kw) for s in range(1, 17)]
    out["rendered90"][name] = {"seeds": 16, "accepted_min": min(r["accepted"] for r in rows), "rmse_min_med_max": [round(float(v), 5) for v in np.percentile([r["trajectory_rmse"] for r in rows], [0, 50, 100])],
                               "fps_median": round(float(np.median([r["frames_per_s"] for r in rows])), 1)}
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))
