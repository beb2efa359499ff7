#!/usr/bin/env python3
"""In-loop adjustment on the reference's example sequence (80 frames, 8 seeds) under different border margins."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, mqslam_amd, run_icl_nuim
S = mqslam_amd.slam_device.DeviceMonoSlam
orig = S.__init__
for margin in (0.0, 1e-6, 3.0, 6.0):
    def init(self, *a, _m=margin, **k):
        orig(self, *a, **k); self.ba_border_margin = _m
    S.__init__ = init
    r = [run_icl_nuim.run(80, bundle_adjust="keyframe", seed=s)["ours_vs_groundtruth_rmse_m"] for s in range(8)]
    print(margin, [round(1e3 * v, 1) for v in r], "median", round(1e3 * float(np.median(r)), 2))
