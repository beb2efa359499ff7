#!/usr/bin/env python3
"""Timing of the matcher kernels (A/B builds: MQS_LIB_PATH=... python tools/bench_match.py [N] [D])"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = mqslam_amd.matching
tb = M.binary_descriptors(N, D, seed=7); qb = M.binary_descriptors(N, D, seed=8, copies_of=tb.astype(np.uint8))
q = torch.from_numpy(qb).cuda(); t = torch.from_numpy(tb).cuda()
idx = torch.empty((N, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((N, 2), dtype=torch.float32, device="cuda")
ws = torch.empty(int(mqslam_amd._lib.lib().mqs_match_knn2_f16_workspace_bytes(N, N)), dtype=torch.uint8, device="cuda")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N, "D": D}
for rnd in range(6):
    ms = timed(lambda: M.knn2_dev(q, t, idx, dist, ws))
    out.setdefault("f16_ms", []).append(round(ms, 3))
    out.setdefault("f16_TFLOPs", []).append(round(2.0 * N * N * D / (ms * 1e-3) / 1e12, 1))
qp = torch.from_numpy(M.pack_bits(qb)).cuda(); tp = torch.from_numpy(M.pack_bits(tb)).cuda()
if D in (128, 256, 512):
    ws8 = torch.empty(int(mqslam_amd._lib.lib().mqs_match_knn2_bits_workspace_bytes(N, N, D)), dtype=torch.uint8, device="cuda")
    idx8 = torch.empty_like(idx); dist8 = torch.empty_like(dist)
    for rnd in range(6):
        ms = timed(lambda: M.knn2_bits_dev(qp, tp, idx8, dist8, ws8))
        out.setdefault("bits_ms", []).append(round(ms, 3))
        out.setdefault("bits_Tops", []).append(round(2.0 * N * N * D / (ms * 1e-3) / 1e12, 1))
    M.knn2_dev(q, t, idx, dist, ws)
    out["bits_equals_f16"] = bool(torch.equal(idx, idx8) and torch.equal(dist, dist8))
print(json.dumps(out))
