#!/usr/bin/env python3
"""Timing of mqs_sba_solve_banded_dev on DENSE (non-banded) reduced camera systems -- pose graphs with loop closures:
python tools/bench_dense_solve.py [P ...]   (A/B builds: MQS_LIB_PATH=...)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd

lib = mqslam_amd._lib.lib()
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH)}
for P in [int(v) for v in sys.argv[1:]] or [150, 300, 500, 881]:
    n = 6 * P
    g = torch.Generator(device="cuda").manual_seed(P)
    B = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    S0 = B @ B.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    b0 = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    poses = torch.zeros((P, 12), dtype=torch.float64, device="cuda")
    ref = torch.linalg.solve(S0, b0)
    times = []
    for rep in range(4):
        S = S0.clone().reshape(-1); x = b0.clone()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        mqslam_amd._lib.check(lib.mqs_sba_solve_banded_dev(ctypes.c_void_p(S.data_ptr()), ctypes.c_void_p(x.data_ptr()), P, n, 0.0,
                                                           ctypes.c_void_p(poses.data_ptr()), None, ctypes.c_void_p(bad.data_ptr()),
                                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record(); e1.synchronize()
        times.append(round(e0.elapsed_time(e1), 3))
    err = float((x - ref).abs().max() / ref.abs().max())
    out[str(n)] = {"ms": times, "rel_err": err, "bad": int(bad.item())}
print(json.dumps(out))
