#!/usr/bin/env python3
"""The in-loop adjustment on the reference's example sequence under two sets of noise models: this build's defaults and the values
the reference's own files hold for its run on this sequence (BA_info.noise.*-slam2.txt)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, mqslam_amd, run_icl_nuim
S = mqslam_amd.slam_device.DeviceMonoSlam
orig = S.__init__
def make(kind):
    def init(self, *a, **k):
        orig(self, *a, **k)
        if kind == "reference":
            self.ba_point_sigma = 0.2
            self.ba_pose_sigmas = (0.02, 0.02, 0.02, 0.1, 0.1, 0.1)
    return init
for kind in ("default", "reference"):
    S.__init__ = make(kind)
    for seed in range(4):
        o = run_icl_nuim.run(80, bundle_adjust="keyframe", seed=seed)
        print(kind, seed, o["ours_vs_groundtruth_rmse_m"], o["ours_vs_groundtruth_max_m"], o["orientation_rmse_deg"]["ours_vs_groundtruth"], o.get("online_vs_groundtruth_rmse_m"))
