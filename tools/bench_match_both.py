#!/usr/bin/env python3
"""Both matrix paths of the matcher (fp16 {0,1} and packed bits on FP4) on the BASELINE configs[2] pair, for the counter passes
(tools/pmc_pass.sh OUT.json tools/bench_match_both.py "SET" ...) and for timing: python tools/bench_match_both.py [N] [launches]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
M = mqslam_amd.matching
tb = M.binary_descriptors(N, 256, seed=7); qb = M.binary_descriptors(N, 256, seed=8, copies_of=tb.astype(np.uint8))
qd, td = torch.from_numpy(qb).cuda(), torch.from_numpy(tb).cuda()
qp, tp = torch.from_numpy(M.pack_bits(qb)).cuda(), torch.from_numpy(M.pack_bits(tb)).cuda()
lib = mqslam_amd._lib.lib()
ws16 = torch.empty(int(lib.mqs_match_knn2_f16_workspace_bytes(N, N)), dtype=torch.uint8, device="cuda")
ws4 = torch.empty(int(lib.mqs_match_knn2_bits_workspace_bytes(N, N, 256)), dtype=torch.uint8, device="cuda")
idx = torch.empty((N, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((N, 2), dtype=torch.float32, device="cuda")
out = {"N": N}
for name, fn in (("f16", lambda: M.knn2_dev(qd, td, idx, dist, ws16)), ("fp4", lambda: M.knn2_bits_dev(qp, tp, idx, dist, ws4))):
    for _ in range(L):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(L):
        fn()
    e1.record(); e1.synchronize()
    out[name + "_ms"] = round(e0.elapsed_time(e1) / L, 4)
print(json.dumps(out))
