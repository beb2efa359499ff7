#!/usr/bin/env python3
"""Frames per second of the loop with bundle adjustment per keyframe on the reference's example sequence (second run in one process),
and the adjustment reports' times."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, mqslam_amd, run_icl_nuim
S = mqslam_amd.slam_device.DeviceMonoSlam
reports = []
orig = S._bundle_adjust
def traced(self):
    orig(self); reports.append(self.ba_reports[-1])
S._bundle_adjust = traced
for rep in range(3):
    reports.clear()
    out = run_icl_nuim.run(80, bundle_adjust="keyframe", seed=0)
    print(out["frames_per_s"], [(r["frame"], r["passes"], r["lm_iterations"], r["build_ms"], r["adjust_ms"], r["write_back_ms"]) for r in reports])
