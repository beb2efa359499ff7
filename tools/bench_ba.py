#!/usr/bin/env python3
"""Kernel timing of the BA launches (A/B builds: MQS_LIB_PATH=... python tools/bench_ba.py [N] [C])"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, C)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N, "C": C}
for rnd in range(2):
    out.setdefault("linearize_us", []).append(timed(lambda: ba.linearize(0.0)))
    out.setdefault("linearize_kernel_only_us", []).append(round(1e3 * ba.time_kernel("linearize"), 1))
    out.setdefault("solve_us", []).append(timed(lambda: ba.solve(0.0)))
    out.setdefault("backsub_us", []).append(timed(lambda: ba.backsub(0.0)))
    out.setdefault("cost_us", []).append(timed(lambda: ba.cost()))
    out.setdefault("gn_iter_us", []).append(timed(lambda: ba.gauss_newton_iteration(0.0)))
print(json.dumps(out))
