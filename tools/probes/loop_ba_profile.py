#!/usr/bin/env python3
"""Per-frame trajectory error of the device-resident loop: plain, with the bundle adjustment per keyframe (online / adjusted), and the
post-hoc adjustment of the plain run's recorded problem (tools/run_slam_loop.py --ba)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import numpy as np, torch, mqslam_amd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
objp, imgp = objp[vis], imgp[vis]
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
gt = seq.centres()
def centres(poses):
    out = np.full((len(poses), 3), np.nan)
    for i, P in enumerate(poses):
        if P is not None: out[i] = -P[:, :3].T @ P[:, 3]
    return out
res = {}
for name, kw in (("plain", {}), ("ba", {"bundle_adjust": "keyframe"})):
    slam = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, **kw)
    slam.start(imgs[0], objp, imgp)
    snaps = []
    if kw:
        orig = slam._bundle_adjust
        def wrapped(orig=orig, slam=slam):
            orig()
            e = np.linalg.norm(centres(slam.poses) - gt[:len(slam.poses)], axis=1)
            snaps.append([round(1e3 * float(v), 1) for v in e])
        slam._bundle_adjust = wrapped
    for k in range(1, frames): slam.handle_new_frame(imgs[k])
    slam.finish()
    e = np.linalg.norm(centres(slam.poses) - gt, axis=1)
    res[name] = {"rmse": float(np.sqrt(np.nanmean(e ** 2))), "err_per_frame_x1000": [round(1e3 * float(v), 1) for v in e], "keyframes": slam.keyframes}
    if kw:
        lm, ps, uv = slam.read_log()
        pts = slam.objp.astype(np.float64)
        cnt, rms = [], []
        for k, f in enumerate(slam._accepted):
            sel = ps == k
            P = slam.poses[f]
            X = pts[lm[sel]] @ P[:, :3].T + P[:, 3]
            pr = X[:, :2] / X[:, 2:3]
            px = np.stack([seq.K[0, 0] * pr[:, 0] + seq.K[0, 2], seq.K[1, 1] * pr[:, 1] + seq.K[1, 2]], 1)
            d = np.linalg.norm(px - uv[sel], axis=1)
            cnt.append(int(sel.sum())); rms.append(round(float(np.sqrt(np.mean(d ** 2))), 2) if sel.any() else None)
        res[name]["obs_per_pose"] = cnt
        res[name]["reproj_rms_per_pose"] = rms
        res[name]["n0"] = int(slam._n0)
        res[name]["snaps"] = snaps
        eo = np.linalg.norm(centres(slam.poses_online) - gt, axis=1)
        res[name]["online_err_per_frame_x1000"] = [round(1e3 * float(v), 1) for v in eo]
        res[name]["reports"] = [(r["frame"], r["lm_iterations"], round(r["cost_before"], 1), round(r["cost_after"], 1)) for r in slam.ba_reports]
    slam.close()
print(json.dumps(res))
