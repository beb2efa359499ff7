#!/usr/bin/env python3
"""Per frame of the plain device loop on the reference's example sequence: distance from the exact trajectory and the frame's report."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, torch, mqslam_amd, run_icl_nuim
frames, seed = int(sys.argv[1]), int(sys.argv[2])
d = np.load(run_icl_nuim.FIX)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"][:frames]]
gt, ref = d["traj_groundtruth"][:, 1:4], d["traj_slam2"][:, 1:4]
s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, max_homography_points="reference")
s.start(imgs[0], pts[vis], uv[vis])
for k in range(1, frames):
    r = s.handle_new_frame(imgs[k])
    s.finish()
    rep = s.reports[-1]
    P = s.poses[k]
    c = -P[:, :3].T @ P[:, 3]
    kf = ""
    if r == 2:
        kf = "KEYFRAME added %d tracks-after %d map %d" % (int(s._pres_copy[25]) if hasattr(s, "_pres_copy") else -1, len(s.tracks()[0]), len(s.objp))
    print(k, r, "err mm %.2f (ref %.2f)" % (1e3 * np.linalg.norm(c - gt[k]), 1e3 * np.linalg.norm(ref[k] - gt[k])), "tracks %d lm-tracks %d inl %d lost %.3f outl %.3f rms %.3f ratio %.4f" % (rep[2], rep[3], rep[4], rep[7], rep[8], rep[9], rep[10]), kf)
