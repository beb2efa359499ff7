#!/usr/bin/env python3
"""mqs_sba_solve_banded_dev on banded systems of several shapes, chunked order (default) against the natural order
(MQS_SBA_PARTS=1 in the environment): python tools/bench_banded_solve.py [P:hb ...]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd

lib = mqslam_amd._lib.lib()
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(100, 17), (186, 101), (186, 35), (300, 65), (500, 101), (881, 101), (881, 35), (2000, 101)]
out = {"parts_env": os.environ.get("MQS_SBA_PARTS", "")}
for P, hb in shapes:
    n = 6 * P
    rng = np.random.default_rng(P + hb)
    S = np.zeros((n, n))
    for d in range(1, hb + 1):
        S[np.arange(n - d), np.arange(d, n)] = rng.standard_normal(n - d)
    S = S + S.T
    S[np.arange(n), np.arange(n)] = np.abs(S).sum(axis=1) + 1.0
    S0 = torch.from_numpy(S).cuda(); b0 = torch.from_numpy(rng.standard_normal(n)).cuda()
    bad = torch.zeros(1, dtype=torch.int32, device="cuda"); poses = torch.zeros((P, 12), dtype=torch.float64, device="cuda")
    ts = []
    for rep in range(5):
        A = S0.clone().reshape(-1); x = b0.clone()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        mqslam_amd._lib.check(lib.mqs_sba_solve_banded_dev(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(x.data_ptr()), P, hb, 0.0,
                                                           ctypes.c_void_p(poses.data_ptr()), None, ctypes.c_void_p(bad.data_ptr()),
                                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    need = lib.mqs_sba_solve_plan_dump(n, hb, int(os.environ.get("MQS_SBA_PARTS", "0") or 0), None, 0)
    out["%d:%d" % (P, hb)] = {"ms": round(min(ts[1:]), 3), "cut": bool(need)}
print(json.dumps(out))
