#!/usr/bin/env python3
"""End-to-end monocular loop (slam_loop.MonoSlam) on the rendered plane sequence; prints accuracy and frames/s."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mqslam_amd


def run(frames=60, verbose=False):
    seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    objp, imgp = objp[vis], imgp[vis]
    imgs = [seq.render(k) for k in range(frames)]
    slam = mqslam_amd.slam_loop.MonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, verbose=verbose)
    slam.start(imgs[0], objp, imgp)
    rets = [2]
    for k in range(1, frames):
        rets.append(slam.handle_new_frame(imgs[k]))
    traj, gt = slam.trajectory(), seq.centres()
    ok = np.isfinite(traj[:, 0])
    err = np.linalg.norm(traj[ok] - gt[ok], axis=1)
    new = slam.objp[len(objp):]
    return {"frames": frames, "accepted": int(ok.sum()), "keyframes": int(sum(r == 2 for r in rets)),
            "landmarks_triangulated": int(len(new)), "trajectory_rmse": float(np.sqrt(np.mean(err ** 2))),
            "trajectory_max_err": float(err.max()), "path_length": float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()),
            "map_plane_median_abs_z": float(np.median(np.abs(new[:, 2]))) if len(new) else None,
            "map_plane_p90_abs_z": float(np.percentile(np.abs(new[:, 2]), 90)) if len(new) else None,
            "frames_per_s": round((frames - 1) / sum(slam.timing), 1)}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, verbose="-v" in sys.argv)))
