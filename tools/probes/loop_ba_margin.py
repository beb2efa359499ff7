#!/usr/bin/env python3
"""The in-loop adjustment on the rendered sequence under different border margins / minimum observation counts."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
frames = int(sys.argv[1])
seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
gt = seq.centres()
for margin, minobs in ((1e-6, 3), (3.0, 3), (6.0, 3), (10.0, 3)):
    s = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=0, bundle_adjust="keyframe")
    s.ba_border_margin, s.ba_min_observations = margin, minobs
    s.start(imgs[0], objp[vis], imgp[vis])
    for k in range(1, frames):
        s.handle_new_frame(imgs[k])
    s.finish()
    c = np.array([-P[:, :3].T @ P[:, 3] for P in s.poses])
    co = np.array([-P[:, :3].T @ P[:, 3] for P in s.poses_online])
    r = lambda c: round(float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))), 5)
    first = s.ba_reports[0]
    print(json.dumps({"margin": margin, "min_obs": minobs, "rmse": r(c), "online": r(co), "screened": int(s.retired_landmarks().sum()), "first_adjustment": [first["frame"], first["landmarks_adjusted"], first["observations"]]}))
    s.close()
