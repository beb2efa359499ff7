#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in host-pointer calls (numpy in, numpy out) at 1e6 x 4 and at the
reference's real size (N = 300, 2 views, slam2.py:1080-1082)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mqslam_amd
tc = mqslam_amd.triangulation_c
out = {}
u, P, _ = mqslam_amd.synthetic.triangulation_problem(1_000_000, 4)
for name, fn in (("linear_ls", tc.linear_LS_triangulation_nview), ("iterative_ls", tc.iterative_LS_triangulation_nview),
                 ("linear_eigen", tc.linear_eigen_triangulation_nview)):
    fn(u, P); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(u, P); ts.append(time.perf_counter() - t0)
    out["1e6x4_%s_ms" % name] = round(1e3 * min(ts), 3)
    out["1e6x4_%s_landmarks_per_s" % name] = round(1e6 / min(ts))
u2, P2, _ = mqslam_amd.synthetic.triangulation_problem(300, 2)
t = mqslam_amd.triangulation
t.iterative_LS_triangulation(u2[0], P2[0], u2[1], P2[1]); ts = []
for _ in range(50):
    t0 = time.perf_counter(); t.iterative_LS_triangulation(u2[0], P2[0], u2[1], P2[1]); ts.append(time.perf_counter() - t0)
out["300x2_iterative_ls_call_us"] = round(1e6 * np.median(ts), 1)
print(json.dumps(out))
