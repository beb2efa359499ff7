#!/usr/bin/env python3
"""Phase stamps of the resident in-loop adjuster (mqs_debug_slam_ba_stamps) on the rendered sequence: microseconds per phase of the
last adjustment and the per-adjustment wall time of the run."""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mqslam_amd
from mqslam_amd import _lib

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
objp, imgp = objp[vis], imgp[vis]
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
names = {0: "start", 1: "cleared", 2: "log->table", 3: "lists", 8: "hit-lists", 4: "lm-begin", 5: "screen", 6: "write-back", 7: "end", 10: "trial", 11: "records",
         12: "barrier", 13: "system", 14: "barrier", 15: "cholesky", 16: "backsolve", 17: "barrier", 18: "landmarks+cost", 19: "cost-reduced", 20: "sys-task-begin", 21: "sys-accumulated", 22: "sys-reduced", 40: "chol-task-begin", 41: "chol-loaded", 42: "chol-updated", 43: "chol-factored", 44: "chol-barrier", 30: "bs-step-begin", 31: "bs-diag-in-lds", 32: "bs-x", 33: "bs-cols"}
for rep in range(2):
    slam = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1, bundle_adjust="keyframe")
    slam.ba_workgroups = groups
    slam.start(imgs[0], objp, imgp)
    n = ctypes.c_int32(0)
    _lib.check(_lib.lib().mqs_debug_slam_ba_stamps(slam._h, None, 0, ctypes.byref(n)))        # on
    for k in range(1, frames):
        slam.handle_new_frame(imgs[k])
    slam.finish()
    if rep == 1:
        buf = np.zeros((2048, 2), np.int64)
        _lib.check(_lib.lib().mqs_debug_slam_ba_stamps(slam._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 2048, ctypes.byref(n)))
        st = buf[:n.value]
        t = (st[:, 1] - st[0, 1]) / 100.0
        agg = {}
        for k in range(1, len(st)):
            key = names.get(int(st[k, 0]), str(st[k, 0]))
            agg.setdefault(key, []).append(t[k] - t[k - 1])
        print(json.dumps({"last_adjustment_total_us": round(float(t[-1]), 1), "report": slam.ba_reports[-1],
                          "phase_us_sum": {k: round(float(np.sum(v)), 1) for k, v in agg.items()},
                          "phase_us_mean": {k: round(float(np.mean(v)), 2) for k, v in agg.items()}, "count": {k: len(v) for k, v in agg.items()}}))
        print(json.dumps({"adjust_ms": [r["adjust_ms"] for r in slam.ba_reports], "poses": [r["poses"] for r in slam.ba_reports],
                          "trials": [r["lm_trials"] for r in slam.ba_reports], "frames_per_s": round((frames - 1) / sum(slam.timing), 1)}))
    slam.close()
