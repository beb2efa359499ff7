#!/usr/bin/env python3
"""Matcher timing for A/B library builds (MQS_LIB_PATH selects the build): one 65 536 x 65 536 x 256 pair on the fp16 and on the FP4 matrix path,
hipEvents around `reps` launches after a warm-up, several rounds in one process.
    MQS_LIB_PATH=build/ab6/libmqslam_X.so python tools/ab_match.py [n=65536] [rounds=3] [reps=20]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
M = mqslam_amd.matching
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tb = M.binary_descriptors(n, 256, seed=7)
qb = M.binary_descriptors(n, 256, seed=8, copies_of=tb.astype(np.uint8))
q, t = torch.from_numpy(qb).cuda(), torch.from_numpy(tb).cuda()
qp, tp = torch.from_numpy(M.pack_bits(qb)).cuda(), torch.from_numpy(M.pack_bits(tb)).cuda()
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "n": n, "f16_ms": [], "fp4_ms": []}
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(60): M.knn2_dev(q, t)
i16, d16 = M.knn2_dev(q, t)
for r in range(rounds):
    out["f16_ms"].append(round(timed(lambda: M.knn2_dev(q, t)), 4))
for _ in range(120): M.knn2_bits_dev(qp, tp)
i4, d4 = M.knn2_bits_dev(qp, tp)
for r in range(rounds):
    out["fp4_ms"].append(round(timed(lambda: M.knn2_bits_dev(qp, tp)), 4))
torch.cuda.synchronize()
out["f16_frac_of_2.5PF"] = round(2.0 * n * n * 256 / (min(out["f16_ms"]) * 1e-3) / 2.5e15, 4)
out["fp4_frac_of_10P"] = round(2.0 * n * n * 256 / (min(out["fp4_ms"]) * 1e-3) / 1e16, 4)
out["paths_agree"] = bool(torch.equal(i16, i4) and torch.equal(d16, d4))
out["checksum"] = int(i16.to(torch.int64).sum().item())
print(json.dumps(out))
