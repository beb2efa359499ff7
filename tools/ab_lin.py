#!/usr/bin/env python3
"""Lineariser / tail / iteration timing for A/B library builds, several rounds in one process (MQS_LIB_PATH selects the build):
    MQS_LIB_PATH=build/ab/libmqslam_X.so python tools/ab_lin.py [N] [C] [rounds] [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, C)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N, "C": C, "lin_us": [], "tail_us": [], "iter_us": []}
ba.time_kernel("linearize", reps=reps)
for r in range(rounds):
    out["lin_us"].append(round(1e3 * ba.time_kernel("linearize", reps=reps), 2))
    ba.linearize(0.0)
    out["tail_us"].append(round(1e3 * ba.time_kernel("solve_backsub", reps=reps), 2))
    out["iter_us"].append(round(1e3 * mqslam_amd.bundle_adjustment.time_iterations(ba, iters=reps, warm=20), 2))
print(json.dumps(out))
