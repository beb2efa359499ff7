#!/usr/bin/env python3
"""Kernel-only timing of the three triangulation kernels (hipEvents), for A/B builds: MQS_LIB_PATH=... python tools/bench_tri.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
u, P, _ = mqslam_amd.synthetic.triangulation_problem(N, C)
ud = torch.from_numpy(u).cuda(); Pd = torch.from_numpy(np.ascontiguousarray(P)).cuda()
D = mqslam_amd.device
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N, "C": C}
for rnd in range(3):
    for k in ("linear_ls", "iterative_ls", "linear_eigen"):
        ms = D.time_triangulation(k, ud, Pd, reps=30)
        out.setdefault(k, []).append(round(ms * 1e3, 2))
print(json.dumps(out))
