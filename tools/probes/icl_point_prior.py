#!/usr/bin/env python3
"""Why the in-loop adjustment ends FARTHER from the exact trajectory than the plain loop on the first 80 frames of the reference's example
sequence (and much nearer over 200): the 23 start-up points are the data set's EXACT model points; the plain loop uses them as given,
the adjuster -- like bundle_adjust.cpp:277-281 -- only as a gauge prior (sigma 0.25 m here, 0.2 in the reference's noise file) and moves
them.  This probe runs the loop with that prior at several sigmas.   python tools/probes/icl_point_prior.py [frames=80] [seeds=8]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import mqslam_amd, run_icl_nuim

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = run_icl_nuim.load_sequence(frames)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"][:frames]]
gt = run_icl_nuim.centres_from_tum(d["traj_groundtruth"][:frames])
out = {"frames": frames}
for sigma in (0.25, 0.05, 0.01, 0.002):
    rows, moved = [], []
    for seed in range(seeds):
        s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, bundle_adjust="keyframe", max_homography_points="reference")
        s.ba_point_sigma = sigma
        s.start(imgs[0], pts[vis], uv[vis])
        for k in range(1, frames):
            s.handle_new_frame(imgs[k])
        s.finish()
        c = np.array([-P[:, :3].T @ P[:, 3] for P in s.poses])
        rows.append(float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))))
        moved.append(float(np.sqrt(np.mean(np.sum((s.objp[:vis.sum()].astype(np.float64) - pts[vis]) ** 2, axis=1)))))
        s.close()
    out["point_sigma_%g" % sigma] = {"rmse_vs_groundtruth_mm": [round(1e3 * r, 2) for r in rows], "median_mm": round(1e3 * float(np.median(rows)), 2),
                                     "start_up_points_moved_rms_mm": round(1e3 * float(np.median(moved)), 2)}
print(json.dumps(out))
