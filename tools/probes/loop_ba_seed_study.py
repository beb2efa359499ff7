#!/usr/bin/env python3
"""The device-resident loop with the bundle adjustment per keyframe (and the re-association) over several RANSAC seeds and sequence
lengths: trajectory RMSE plain / online / adjusted, landmarks screened out, corners re-associated."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
out = []
for frames in (40, 60, 90):
    seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    objp, imgp = objp[vis], imgp[vis]
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
    gt = seq.centres()
    def rmse(poses):
        c = np.array([(-P[:, :3].T @ P[:, 3]) if P is not None else [np.nan] * 3 for P in poses])
        ok = np.isfinite(c[:, 0])
        return float(np.sqrt(np.mean(np.sum((c[ok] - gt[ok]) ** 2, axis=1)))), int(ok.sum())
    for seed in (0, 1, 2, 3):
        row = {"frames": frames, "seed": seed}
        for name, kw in (("plain", {}), ("ba", {"bundle_adjust": "keyframe"}), ("ba_reassoc", {"bundle_adjust": "keyframe", "reassociate": True})):
            s = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=seed, **kw)
            s.start(imgs[0], objp, imgp)
            for k in range(1, frames):
                s.handle_new_frame(imgs[k])
            s.finish()
            r, acc = rmse(s.poses)
            row[name] = {"rmse": round(r, 5), "accepted": acc, "keyframes": len(s.keyframes), "landmarks": int(len(s.objp))}
            if kw:
                row[name]["rmse_online"] = round(rmse(s.poses_online)[0], 5)
                row[name]["screened_out"] = int(s.retired_landmarks().sum())
                row[name]["reassociated"] = int(s.reassociated)
            s.close()
        out.append(row)
print(json.dumps(out))
