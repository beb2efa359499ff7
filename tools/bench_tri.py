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
# the step's fused linear-LS + iterative-LS launch (tri_kernel<C, 3>), torch events around 30 calls
x_ls = torch.empty((N, 3), dtype=torch.float64, device="cuda"); x_it = torch.empty_like(x_ls); st = torch.empty((N,), dtype=torch.int32, device="cuda")
for rnd in range(3):
    D.linear_and_iterative_LS_triangulation(ud, Pd, out_ls=x_ls, out_it=x_it, out_status=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        D.linear_and_iterative_LS_triangulation(ud, Pd, out_ls=x_ls, out_it=x_it, out_status=st)
    e1.record(); e1.synchronize()
    out.setdefault("ls_and_iterative_fused", []).append(round(e0.elapsed_time(e1) / 30 * 1e3, 2))
print(json.dumps(out))
