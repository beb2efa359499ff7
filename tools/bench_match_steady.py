#!/usr/bin/env python3
"""Steady-clock timing of the packed-bit (FP4) matcher for A/B builds: MQS_LIB_PATH=... python tools/bench_match_steady.py [N] [D]
(120 untimed launches first: the clock under matrix load settles over ~60 ms; then 4 x 20 timed launches)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = mqslam_amd.matching
tb = M.binary_descriptors(N, D, seed=7); qb = M.binary_descriptors(N, D, seed=8, copies_of=tb.astype(np.uint8))
qp = torch.from_numpy(M.pack_bits(qb)).cuda(); tp = torch.from_numpy(M.pack_bits(tb)).cuda()
ws8 = torch.empty(int(mqslam_amd._lib.lib().mqs_match_knn2_bits_workspace_bytes(N, N, D)), dtype=torch.uint8, device="cuda")
idx8 = torch.empty((N, 2), dtype=torch.int32, device="cuda"); dist8 = torch.empty((N, 2), dtype=torch.float32, device="cuda")
for _ in range(120):
    M.knn2_bits_dev(qp, tp, idx8, dist8, ws8)
torch.cuda.synchronize()
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N, "D": D, "bits_ms": []}
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for rnd in range(4):
    e0.record()
    for _ in range(20):
        M.knn2_bits_dev(qp, tp, idx8, dist8, ws8)
    e1.record(); e1.synchronize()
    out["bits_ms"].append(round(e0.elapsed_time(e1) / 20, 4))
# (correctness of these launches against the oracle: tests/test_matching.py -- tools/ do not import oracle/)
print(json.dumps(out))
