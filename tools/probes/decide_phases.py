#!/usr/bin/env python3
"""Phases of frame_decide_kernel on the rendered sequence, from a -DMQS_DECIDE_STAMPS build (tools/build_variant.sh stamps slam_frame.hip
-DMQS_DECIDE_STAMPS; MQS_LIB_PATH=build/ab/libmqslam_stamps.so python tools/probes/decide_phases.py [frames]): microseconds from the kernel's
start to the end of: gates + reprojection RMS, commit of the tracks, sample, DLT sums, 9 x 9 eigenvectors, H from the null vector,
Levenberg-Marquardt refinement, singular values."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq = mqslam_amd.synthetic.PlaneSequence(frames=frames)
gx, gy = np.meshgrid(np.linspace(-4.5, 1.0, 8), np.linspace(-2.5, 2.0, 6))
objp = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], axis=1)
imgp = seq.project(0, objp)
vis = (imgp[:, 0] > 15) & (imgp[:, 0] < seq.W - 15) & (imgp[:, 1] > 15) & (imgp[:, 1] < seq.H - 15)
imgs = [torch.from_numpy(seq.render(k)).cuda() for k in range(frames)]
s = mqslam_amd.slam_device.DeviceMonoSlam(seq.K, seq.dist, (seq.H, seq.W), seed=1)
s.start(imgs[0], objp[vis], imgp[vis])
rows = []
for k in range(1, frames):
    ret = s.handle_new_frame(imgs[k])
    if ret == 1:                                  # accepted, no keyframe: the stamps have not been overwritten
        rows.append(np.array(s._res[28:36], dtype=np.float64) / 100.0)
rows = np.array(rows)
names = ["gates+rms", "commit", "sample", "dlt_sums", "jacobi9", "h_from_null_vector", "lm_refine", "singular_values"]
med = np.median(rows, axis=0)
print(json.dumps({"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "frames": len(rows), "end_of_phase_us_median": dict(zip(names, np.round(med, 2).tolist())),
                  "phase_us_median": dict(zip(names, np.round(np.diff(np.concatenate([[0.0], med])), 2).tolist()))}))
s.close()
