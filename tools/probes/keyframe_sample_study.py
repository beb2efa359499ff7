#!/usr/bin/env python3
"""keyframe_test with the reference's random quarter of the tracks (slam2.py:48, 1088-1089) against all tracks: trajectory
error of the device-resident loop on the rendered sequence over several seeds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
frames = 60
seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
objp, imgp = objp[vis], imgp[vis]
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
gt = seq.centres()
out = {}
for name, mh in (("reference_quarter", None), ("all_tracks", 0)):
    rows = []
    for seed in range(8):
        slam = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=seed, max_homography_points=mh)
        slam.start(imgs[0], objp, imgp)
        rets = [2] + [slam.handle_new_frame(imgs[k]) for k in range(1, frames)]
        traj = slam.trajectory()
        ok = np.isfinite(traj[:, 0])
        err = np.linalg.norm(traj[ok] - gt[ok], axis=1)
        rows.append({"seed": seed, "accepted": int(ok.sum()), "keyframes": int(sum(r == 2 for r in rets)),
                     "rmse": round(float(np.sqrt(np.mean(err ** 2))), 5)})
        slam.close()
    out[name] = rows
out["path_length"] = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
print(json.dumps(out))
