#!/usr/bin/env python3
"""Lineariser time on the benchmark problem with and without an observation mask (10 % of the factors dropped)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, dev, seed=1)
out = {"N": N, "unmasked_us": [round(1e3 * ba.time_kernel("linearize"), 1) for _ in range(3)]}
g = torch.Generator(device="cpu").manual_seed(3)
mask = (torch.rand((4, N), generator=g) > 0.1).to(torch.uint8).to(dev)
bm = mqslam_amd.bundle_adjustment.BundleAdjuster(ba.poses, ba.calib, ba.sigma, ba.points, ba.obs, mask=mask, prior_w=ba.prior_w, prior_xyz=ba.prior_xyz)
out["masked_us"] = [round(1e3 * bm.time_kernel("linearize"), 1) for _ in range(3)]
print(json.dumps(out))
