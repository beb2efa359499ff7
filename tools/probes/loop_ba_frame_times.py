#!/usr/bin/env python3
"""Wall time of every handle_new_frame of the device loop with bundle adjustment per keyframe, beside the adjustment's own report."""
import os, sys, json, time
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
torch.cuda.synchronize()
for rep in range(3):
    s = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    s.start(imgs[0], objp, imgp)
    T = []
    t_all = time.perf_counter()
    for k in range(1, frames):
        t0 = time.perf_counter()
        r = s.handle_new_frame(imgs[k])
        T.append((k, r, 1e3 * (time.perf_counter() - t0)))
    s.finish()
    t_all = time.perf_counter() - t_all
    if rep == 2:
        rep_by_frame = {r["frame"]: r for r in s.ba_reports}
        for k, r, ms in T:
            if k in rep_by_frame:
                b = rep_by_frame[k]
                print(k, r, round(ms, 3), "ba:", b["build_ms"], b["adjust_ms"], b["write_back_ms"], "passes", b["passes"], "it", b["lm_iterations"])
        print("total ms", round(1e3 * t_all, 2), "fps", round(frames / t_all, 1), "non-BA frames median ms",
              round(float(np.median([ms for k, r, ms in T if k not in rep_by_frame])), 3))
    s.close()
