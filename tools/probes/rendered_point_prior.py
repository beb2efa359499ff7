import os, sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch, mqslam_amd
for frames in (60, 40, 90):
    seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
    gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
    objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
    imgp = seq.project(0, objp)
    vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
    imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
    gt = seq.centres()
    for sigma in (0.25, 0.05, 0.01):
        rows = []
        for seed in (1, 2, 3, 4):
            s = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=seed, bundle_adjust="keyframe", reassociate=True)
            s.ba_point_sigma = sigma
            s.start(imgs[0], objp[vis], imgp[vis])
            ok = all(s.handle_new_frame(imgs[k]) in (1, 2) for k in range(1, frames))
            s.finish()
            c = np.array([-P[:, :3].T @ P[:, 3] for P in s.poses])
            rows.append(round(float(np.sqrt(np.mean(np.sum((c - gt) ** 2, axis=1)))), 5))
            s.close()
        print(frames, sigma, rows, flush=True)
