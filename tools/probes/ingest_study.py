#!/usr/bin/env python3
"""Round 6: configs[4] with the frames arriving INSIDE the timed loop (slam_device.FrameUploader: side-stream upload under the previous
frames' kernels) against the same runs with every frame on the device beforehand.  One JSON line per sequence / adjustment / source.
    python tools/probes/ingest_study.py [repeats=3]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import run_icl_nuim as R
import run_slam_loop as L

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R.run(80, "keyframe", 0)
for frames in (80, 200):
    for ba in (None, "keyframe"):
        row = {"sequence": "icl-nuim %d" % frames, "ba": ba}
        for up in (None, "pinned", "pageable"):
            runs = [R.run(frames, ba, 0, upload=up) for _ in range(rep)]
            row[up or "resident"] = {"frames_per_s": [r["frames_per_s"] for r in runs], "best": max(r["frames_per_s"] for r in runs),
                                     "rmse_mm": round(1e3 * runs[0]["ours_vs_groundtruth_rmse_m"], 3)}
        row["with_upload_over_resident"] = {k: round(row[k]["best"] / row["resident"]["best"], 3) for k in ("pinned", "pageable")}
        print(json.dumps(row), flush=True)
for ba in (None, "keyframe"):
    row = {"sequence": "rendered 60", "ba": ba}
    for up in (None, "pinned", "pageable"):
        r = L.run_device(60, repeats=rep, bundle_adjust=ba, reassociate=bool(ba), upload=up)
        row[up or "resident"] = {"frames_per_s": r["frames_per_s"], "rmse_mm": round(1e3 * r["trajectory_rmse"], 3)}
    row["with_upload_over_resident"] = {k: round(row[k]["frames_per_s"] / row["resident"]["frames_per_s"], 3) for k in ("pinned", "pageable")}
    print(json.dumps(row), flush=True)
