#!/usr/bin/env python3
"""Per keyframe of the device loop on the reference's example sequence: tracks before the frame, landmarks added, tracks after the top-up."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd, run_icl_nuim
d = np.load(run_icl_nuim.FIX)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"]]
s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=0, max_homography_points="reference")
s.start(imgs[0], pts[vis], uv[vis])
print("after start: tracks", len(s.tracks()[0]), "landmarks", len(s.objp))
for k in range(1, len(imgs)):
    r = s.handle_new_frame(imgs[k])
    rep = s.reports[-1] if s.reports else None
    if r == 2 or k % 10 == 0:
        s.finish()
        pts_, base_, lm_, tid_ = s.tracks()
        print(k, "ret", r, "report", [round(float(x), 3) for x in rep] if rep is not None else None, "| tracks now", len(pts_), "with landmark", int((lm_ >= 0).sum()), "map", len(s.objp))
