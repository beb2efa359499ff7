#!/usr/bin/env python3
"""Round 6: the next image pair ahead of its frame (mqs_slam_set_next / mqs_slam_pipeline) -- frames/s of every loop leg, frames resident and
arriving inside the timed loop, in three forms: everything inside the frame's call; the pair's pyramid and tracker on the side stream
(MQS_SLAM_TRACK_AHEAD=0: the pyramid alone); and the frame's pose kernels enqueued behind the frame in front as well.
One JSON line per leg.    python tools/probes/prepare_next_study.py [repeats=3]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
import run_slam_loop as L
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FORMS = (("in the call", False, False), ("prepared ahead", True, False), ("enqueued ahead", True, True))
R.run(80, "keyframe", 0)
for frames in (80, 200):
    for ba in (None, "keyframe"):
        row = {"sequence": "icl-nuim %d" % frames, "ba": ba}
        for up in (None, "pageable"):
            for name, ahead, pipe in FORMS:
                runs = [R.run(frames, ba, 0, upload=up, prepare_next=ahead, pipeline=pipe) for _ in range(rep)]
                row["%s, %s" % (up or "resident", name)] = {"best_frames_per_s": max(r["frames_per_s"] for r in runs),
                                                            "rmse_mm": round(1e3 * runs[0]["ours_vs_groundtruth_rmse_m"], 3)}
        print(json.dumps(row), flush=True)
for ba in (None, "keyframe"):
    row = {"sequence": "rendered 60", "ba": ba}
    for up in (None, "pageable"):
        for name, ahead, pipe in FORMS:
            r = L.run_device(60, repeats=rep, bundle_adjust=ba, reassociate=bool(ba), upload=up, prepare_next=ahead, pipeline=pipe)
            row["%s, %s" % (up or "resident", name)] = {"frames_per_s": r["frames_per_s"], "rmse_mm": round(1e3 * r["trajectory_rmse"], 3)}
    print(json.dumps(row), flush=True)
