#!/usr/bin/env python3
"""Stress of the flag-after-data hand-over inside the fused tail (csrc/ba.hip): one-call Gauss-Newton iterations from a
DIFFERENT linearisation point every time (so that a piece read before it landed -- i.e. the previous iteration's -- cannot go
unnoticed), a fingerprint of every iteration's summed system and poses.  Run once as is and once with MQS_BA_FINALIZE=kernel
(the finalize as a launch of its own): the fingerprints must be equal.

    python tools/probes/fused_finalize_stress.py [iterations] [landmarks]
"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
poses0 = ba.poses.clone()
points0 = ba.points.clone()
g = torch.Generator(device="cuda").manual_seed(7)
h = hashlib.sha256()
for k in range(iters):
    noise = 1e-3 * torch.randn(poses0.shape, generator=g, device="cuda", dtype=torch.float64)
    noise[:, :9] = 0.0                                            # keep the rotations rotations: move the camera centres only
    ba.poses.copy_(poses0 + noise)
    ba.points.copy_(points0)
    ba.gauss_newton_iteration(0.0)
    if k % 10 == 0 or k > iters - 20:
        h.update(ba.lin.cpu().numpy().tobytes())
        h.update(ba.poses.cpu().numpy().tobytes())
torch.cuda.synchronize()
h.update(ba.points.cpu().numpy().tobytes())
print(json.dumps({"mode": os.environ.get("MQS_BA_FINALIZE", "fused"), "iterations": iters, "landmarks": N, "sha256_16": h.hexdigest()[:16]}))
