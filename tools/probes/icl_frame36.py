#!/usr/bin/env python3
"""The keyframe refinement at frame 36 of the reference's example sequence (the frame at which, taken as the first keyframe, the
refined pose lands 28 mm from the exact one): the host-driven loop with the keyframe forced there; what solvePnP is given and
what it returns."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, mqslam_amd, run_icl_nuim
L = mqslam_amd.slam_loop
force = int(sys.argv[1]) if len(sys.argv) > 1 else 36
d = np.load(run_icl_nuim.FIX)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
gt = d["traj_groundtruth"][:, 1:4]
state = {"frame": 0}
L.keyframe_test = lambda *a, **k: state["frame"] == force
orig = L.pnp.solvePnP
def spy(obj, img, K_, dist_, rvec=None, tvec=None, useExtrinsicGuess=False, **kw):
    ret = orig(obj, img, K_, dist_, rvec, tvec, useExtrinsicGuess=useExtrinsicGuess, **kw)
    if len(obj) > 100:
        cen = lambda r, t: (-mqslam_amd.pnp.Rodrigues(r).T @ np.asarray(t).reshape(3)).ravel()
        e0, _ = mqslam_amd.camera.reprojection_error(np.asarray(obj, float), np.asarray(img, float), K_, dist_, rvec, tvec)
        e1, _ = mqslam_amd.camera.reprojection_error(np.asarray(obj, float), np.asarray(img, float), K_, dist_, ret[1], ret[2])
        k = state["frame"]
        depth = (mqslam_amd.pnp.Rodrigues(rvec) @ np.asarray(obj, float).T + np.asarray(tvec, float).reshape(3, 1))[2]
        print(json.dumps({"frame": k, "points": len(obj), "old_landmarks": int(state.get("n_old", -1)),
                          "centre_error_before_mm": round(1e3 * float(np.linalg.norm(cen(rvec, tvec) - gt[k])), 2),
                          "centre_error_after_mm": round(1e3 * float(np.linalg.norm(cen(ret[1], ret[2]) - gt[k])), 2),
                          "reprojection_rms_before_after_px": [round(float(e0), 4), round(float(e1), 4)],
                          "depth_of_the_new_points_min_median_max": [round(float(v), 3) for v in (depth[21:].min(), np.median(depth[21:]), depth[21:].max())]}))
        np.savez(os.path.join(ROOT, "gpurun_out", "icl_frame%d_pnp.npz" % k), obj=np.asarray(obj), img=np.asarray(img), rvec=np.asarray(rvec), tvec=np.asarray(tvec),
                 rvec_out=np.asarray(ret[1]), tvec_out=np.asarray(ret[2]), gt_centre=gt[k], K=K_, dist=dist_)
    return ret
L.pnp.solvePnP = spy
s = L.MonoSlam(K, dist, (H, W), seed=0)
s.start(d["frames"][0], pts[vis], uv[vis])
for k in range(1, force + 3):
    state["frame"] = k
    s.handle_new_frame(d["frames"][k])
c = s.trajectory()
print("centre errors mm at", force - 1, force, force + 1, [round(1e3 * float(np.linalg.norm(c[k] - gt[k])), 2) for k in (force - 1, force, force + 1)])
