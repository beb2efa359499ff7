#!/usr/bin/env python3
"""Round 6 soak: the loop's three forms (everything in the frame's call / pair prepared ahead / frame enqueued ahead) x frames resident / arriving in
ordinary memory x plain / BA per keyframe, over seeds, each form three times: every run of a seed has to report the same run (keyframes, landmarks,
errors to the last printed digit) as the in-call form -- the forms move the same kernels on the same inputs to other streams and other times, nothing
else.    python tools/probes/soak_ahead_forms.py [seeds=16] [frames=200]"""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
import run_slam_loop as L
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
KEYS = ("accepted", "keyframes", "landmarks", "keyframe_frames", "ours_vs_groundtruth_rmse_m", "ours_vs_groundtruth_max_m", "ours_vs_reference_rmse_m",
        "ours_vs_reference_max_m", "every_10th_frame_ours_ref_gt_error_mm", "orientation_rmse_deg")
FORMS = (("in the call", False, False), ("prepared ahead", True, False), ("enqueued ahead", True, True))
t0 = time.time()
R.run(80)
runs = mismatches = 0
bad = []
for ba in (None, "keyframe"):
    for seed in range(seeds):
        want = None
        for up in (None, "pageable"):
            for name, ahead, pipe in FORMS:
                for rep in range(3 if ahead else 1):
                    r = R.run(frames, ba, seed, upload=up, prepare_next=ahead, pipeline=pipe)
                    sig = json.dumps([r[k] for k in KEYS])
                    runs += 1
                    if want is None:
                        want = sig
                    elif sig != want:
                        mismatches += 1
                        bad.append({"ba": ba, "seed": seed, "upload": up, "form": name})
rend = 0
for ba in (None, "keyframe"):
    for seed in range(1, 1 + min(seeds, 8)):
        want = None
        for name, ahead, pipe in FORMS:
            for up in (None, "pageable"):
                r = L.run_device(90, seed=seed, bundle_adjust=ba, reassociate=bool(ba), upload=up, prepare_next=ahead, pipeline=pipe)
                sig = json.dumps([r[k] for k in ("accepted", "keyframes", "landmarks_triangulated", "trajectory_rmse", "trajectory_max_err", "tracks_at_the_end")])
                rend += 1
                if want is None:
                    want = sig
                elif sig != want:
                    mismatches += 1
                    bad.append({"rendered": True, "ba": ba, "seed": seed, "upload": up, "form": name})
print(json.dumps({"icl_runs": runs, "rendered_runs": rend, "mismatches": mismatches, "first_mismatches": bad[:8], "seconds": round(time.time() - t0, 1)}))
