#!/usr/bin/env python3
"""Round 6 soak: the decision kernel as two workgroups (the keyframe test beside the pose refinement, its own copy of the inlier test) against one
workgroup (MQS_SLAM_DECIDE_SPLIT=0), over seeds on the 200 example frames, plain and with BA per keyframe: the same run to the last printed digit.
python tools/probes/soak_decide_split.py [seeds=16]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
KEYS = ("accepted", "keyframes", "landmarks", "keyframe_frames", "ours_vs_groundtruth_rmse_m", "ours_vs_groundtruth_max_m", "ours_vs_reference_rmse_m",
        "ours_vs_reference_max_m", "every_10th_frame_ours_ref_gt_error_mm", "orientation_rmse_deg")
R.run(80)
runs = mism = 0
for ba in (None, "keyframe"):
    for seed in range(seeds):
        sig = {}
        for split in ("0", "1"):
            os.environ["MQS_SLAM_DECIDE_SPLIT"] = split
            r = R.run(200, ba, seed)
            sig[split] = json.dumps([r[k] for k in KEYS])
            runs += 1
        mism += sig["0"] != sig["1"]
print(json.dumps({"runs": runs, "mismatches": mism}))
