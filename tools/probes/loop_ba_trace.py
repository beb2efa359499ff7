#!/usr/bin/env python3
"""Per keyframe of the device-resident loop with bundle adjustment: trajectory RMSE (frames so far) before and after the
adjustment, the adjustment's own report.  Arguments: frames seed [reassociate]."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
frames, seed = int(sys.argv[1]), int(sys.argv[2])
re = len(sys.argv) > 3 and sys.argv[3] == "1"
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
    return round(float(np.sqrt(np.mean(np.sum((c[ok] - gt[:len(c)][ok]) ** 2, axis=1)))), 5)
S = mqslam_amd.slam_device.DeviceMonoSlam
orig = S._bundle_adjust
def traced(self):
    before = rmse(self.poses)
    orig(self)
    r = dict(self.ba_reports[-1])
    r["rmse_before"], r["rmse_after"] = before, rmse(self.poses)
    # how far the true map is from the estimate: the plane z = 0 (all landmarks of the rendering lie on it)
    pts = self.objp.astype(np.float64)
    r["map_abs_z_p50_p99_max"] = [round(float(v), 4) for v in (np.percentile(np.abs(pts[:, 2]), 50), np.percentile(np.abs(pts[:, 2]), 99), np.abs(pts[:, 2]).max())]
    for k in ("build_ms", "adjust_ms", "write_back_ms"): r.pop(k)
    print(json.dumps(r))
S._bundle_adjust = traced
s = S(seq.K, seq.dist, (seq.H, seq.W), seed=seed, bundle_adjust="keyframe", reassociate=re)
s.start(imgs[0], objp, imgp)
for k in range(1, frames):
    s.handle_new_frame(imgs[k])
s.finish()
print(json.dumps({"final": rmse(s.poses), "online": rmse(s.poses_online), "keyframes": s.keyframes}))
