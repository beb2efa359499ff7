#!/usr/bin/env python3
"""Dumps the sparse problem of the first adjustment with at least `minP` poses on the reference's example sequence (seed given), then
runs it through the library's LM driver and through the host loop, verbosely."""
import os, sys, json, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, torch, mqslam_amd, run_icl_nuim
seed, minP = int(sys.argv[1]), int(sys.argv[2])
d = np.load(run_icl_nuim.FIX)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"]]
SB = mqslam_amd.sparse_ba.SparseBundleAdjuster
orig_init = SB.__init__
caught = {}
def init(self, problem, device="cuda:0"):
    orig_init(self, problem, device)
    if len(problem.poses) >= minP and "p" not in caught:
        caught["p"] = problem
SB.__init__ = init
s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, bundle_adjust="keyframe", max_homography_points="reference")
s.start(imgs[0], pts[vis], uv[vis])
for k in range(1, len(imgs)):
    s.handle_new_frame(imgs[k])
    if "p" in caught:
        break
SB.__init__ = orig_init
pr = caught["p"]
print("P", len(pr.poses), "N", len(pr.points), "M", len(pr.obs_pose), "finite points", bool(np.isfinite(pr.points).all()), "finite poses", bool(np.isfinite(pr.poses).all()),
      "finite uv", bool(np.isfinite(pr.obs_uv).all()), "max |point|", float(np.abs(pr.points).max()))
a = SB(pr)
print("half_bandwidth", a.half_bandwidth, "n6", a.n6)
print("native:", a.optimize(mode="lm"))
b = SB(pr)
print("host loop:")
print(b.optimize_host_loop(mode="lm", verbose=True))
c = SB(pr)
S, g = c.linearize(-1e-5)
S = S.cpu().numpy().copy(); g = g.cpu().numpy().copy()
print("S finite", bool(np.isfinite(S).all()), "g finite", bool(np.isfinite(g).all()), "S sym err", float(np.abs(S - S.T).max()), "min eig of S + 1e-5 I", float(np.linalg.eigvalsh(S + 1e-5 * np.eye(len(S))).min()))
x = c.solve(-1e-5).cpu().numpy().copy()
want = np.linalg.solve(S + 1e-5 * np.eye(len(S)), g)
print("solve: max |x - numpy| / max |numpy| =", float(np.abs(x - want).max() / np.abs(want).max()), "bad", int(c.bad.item()), "max |numpy x|", float(np.abs(want).max()))
err = np.abs(x - want)
print("first wrong index", int(np.argmax(err > 1e-6 * np.abs(want).max())), "n wrong", int((err > 1e-6 * np.abs(want).max()).sum()))
np.savez(os.path.join(ROOT, "gpurun_out", "icl_bad_system.npz"), S=S, g=g, x=x)
e = SB(pr)
e.step(-1e-5)
dp = (e.points_new - e.points).cpu().numpy()
dq = (e.poses_new - e.poses).cpu().numpy()
nobs = np.diff(np.asarray(pr.obs_ptr))
w_old, w_new = e.worst_residuals(), e.worst_residuals(e.poses_new, e.points_new)
top = np.argsort(-np.linalg.norm(dp, axis=1))[:6]
print("max |dpose rows|", float(np.abs(dq).max()))
for i in top:
    print("landmark", int(i), "obs", int(nobs[i]), "|dp|", float(np.linalg.norm(dp[i])), "point", pr.points[i].round(3).tolist(), "worst px before", float(w_old[i]), "after", float(w_new[i]),
          "poses seen", sorted(set(np.asarray(pr.obs_pose)[pr.obs_ptr[i]:pr.obs_ptr[i + 1]].tolist()))[:4], "..")
top = np.argsort(-w_new)[:6]
print("worst after:", [(int(i), int(nobs[i]), round(float(w_new[i]), 1), round(float(w_old[i]), 2), round(float(np.linalg.norm(dp[i])), 4)) for i in top])
print("cost of the new estimate", e.cost(e.poses_new, e.points_new), "with the old points", e.cost(e.poses_new, e.points), "old poses new points", e.cost(e.poses, e.points_new))
i = 244
ks = range(pr.obs_ptr[i], pr.obs_ptr[i + 1])
P0, P1 = e.poses.cpu().numpy(), e.poses_new.cpu().numpy()
X0, X1 = pr.points[i], e.points_new.cpu().numpy()[i]
print("landmark 244 point", X0.tolist(), "->", X1.tolist())
for k in ks:
    j = int(pr.obs_pose[k])
    z = lambda P, X: float(P[j, :9].reshape(3, 3)[:, 2] @ (X - P[j, 9:]))
    print("  pose", j, "uv", np.asarray(pr.obs_uv).reshape(-1, 2)[k].round(2).tolist(), "depth old %.5f new-poses-old-point %.5f new %.5f" % (z(P0, X0), z(P1, X0), z(P1, X1)),
          "centre", P0[j, 9:].round(4).tolist())
