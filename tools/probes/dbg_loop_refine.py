import os, sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np, torch, mqslam_amd, run_slam_loop
seq = mqslam_amd.synthetic.PlaneSequence(frames=30)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
objp, imgp = objp[vis], imgp[vis]
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(30)]
for refine in ("1", "0"):
    os.environ["MQS_SLAM_HOMOGRAPHY_REFINE"] = refine
    slam = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1)
    slam.start(imgs[0], objp, imgp)
    for k in range(1, 30):
        r = slam.handle_new_frame(imgs[k])
        rep = slam.reports[-1]
        print(refine, k, r, [round(float(x), 4) for x in rep[:12]])
    slam.close()
dev = run_slam_loop.run_device(60)
host = run_slam_loop.run(60) if hasattr(run_slam_loop, "run") else None
print({k: dev[k] for k in ("accepted", "keyframes", "landmarks_triangulated", "trajectory_rmse", "tracks_at_the_end")})
print({k: host[k] for k in ("accepted", "keyframes", "landmarks_triangulated", "trajectory_rmse")} if host else None)
