#!/usr/bin/env python3
"""Kernel time of the pyramidal Lucas-Kanade tracker on the rendered VGA pair (300 features) and a bit-for-bit fingerprint of
its outputs (A/B builds: MQS_LIB_PATH=...)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
seq = mqslam_amd.synthetic.PlaneSequence(frames=4)
a, b = seq.render(0), seq.render(2)
F = mqslam_amd.features
pts = F.goodFeaturesToTrack(a, 300, 0.01, 12)
pts = np.ascontiguousarray(pts.reshape(-1, 2), dtype=np.float32)
nxt, st, err = F.calcOpticalFlowPyrLK(a, b, pts)
h = hashlib.sha256(np.ascontiguousarray(nxt).tobytes() + np.ascontiguousarray(st).tobytes() + np.ascontiguousarray(err).tobytes()).hexdigest()[:16]
import time
ts = []
for _ in range(30):
    t0 = time.perf_counter(); F.calcOpticalFlowPyrLK(a, b, pts); ts.append(time.perf_counter() - t0)
print(json.dumps({"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "features": int(len(pts)), "tracked": int(np.asarray(st).sum()),
                  "outputs_sha256_16": h, "host_call_ms_median": round(1e3 * float(np.median(ts)), 4)}))
