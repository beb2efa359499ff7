#!/usr/bin/env python3
"""Host-call time of goodFeaturesToTrack (300 corners, quality 0.01, distance 12 -- the loop's detection) on a rendered frame and on a frame of
the reference's example sequence; with a -DMQS_GFTT_EXPERIMENT_SORT_ONLY build the call returns the NUMBER OF CANDIDATES instead of corners and
skips the selection (A/B: MQS_LIB_PATH=...)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mqslam_amd, run_icl_nuim
F = mqslam_amd.features
imgs = {"rendered": mqslam_amd.synthetic.PlaneSequence(frames=60).render(10), "example_sequence": np.ascontiguousarray(run_icl_nuim.load_sequence(3)["frames"][2])}
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH)}
for name, img in imgs.items():
    pts = F.goodFeaturesToTrack(img, 300, 0.01, 12)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); F.goodFeaturesToTrack(img, 300, 0.01, 12); ts.append(time.perf_counter() - t0)
    out[name] = {"returned": int(len(pts)), "host_call_us_median": round(1e6 * float(np.median(ts)), 1)}
print(json.dumps(out))
