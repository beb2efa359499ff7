#!/usr/bin/env python3
"""Per keyframe of the device loop with bundle adjustment on the reference's example sequence: distance from the exact trajectory
(frames so far) before and after the adjustment, and the adjustment's own report.  Arguments: frames seed."""
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
gt = d["traj_groundtruth"][:, 1:4]
def rmse(poses):
    c = np.array([(-P[:, :3].T @ P[:, 3]) if P is not None else [np.nan] * 3 for P in poses])
    ok = np.isfinite(c[:, 0])
    e = np.linalg.norm(c[ok] - gt[:len(c)][ok], axis=1)
    return round(float(np.sqrt(np.mean(e ** 2))), 5), round(float(e[-1]), 5)
S = mqslam_amd.slam_device.DeviceMonoSlam
orig = S._bundle_adjust
def traced(self):
    before = rmse(self.poses)
    orig(self)
    r = dict(self.ba_reports[-1])
    r["rmse_before,last"], r["rmse_after,last"] = before, rmse(self.poses)
    for k in ("build_ms", "adjust_ms", "write_back_ms"): r.pop(k)
    print(json.dumps(r))
S._bundle_adjust = traced
s = S(K, dist, (H, W), seed=seed, bundle_adjust="keyframe", max_homography_points="reference")
s.start(imgs[0], pts[vis], uv[vis])
for k in range(1, frames):
    s.handle_new_frame(imgs[k])
s.finish()
print(json.dumps({"final": rmse(s.poses), "online": rmse(s.poses_online), "keyframes": s.keyframes}))
