#!/usr/bin/env python3
"""Where an in-loop adjustment over the whole run spends its time as the run grows (reference example sequence, 200 frames):
construction of the sparse problem on the device, the LM driver, the screens -- and the sizes behind them."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, torch, mqslam_amd, run_icl_nuim
SB = mqslam_amd.sparse_ba.SparseBundleAdjuster
rows = []
oi, oo, ow = SB.__init__, SB.optimize, SB.worst_residuals
def init(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); oi(self, *a, **k); torch.cuda.synchronize()
    rows.append({"P": self.P, "N": self.N, "M": self.M, "pairs": self.Q, "groups": self.G, "half_bandwidth": self.half_bandwidth, "construct_ms": round(1e3 * (time.perf_counter() - t), 2)})
def opt(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); h = oo(self, *a, **k); torch.cuda.synchronize()
    rows[-1]["lm_ms"] = round(1e3 * (time.perf_counter() - t), 2); rows[-1]["lm_accepted_iterations"] = len(h) - 1
    return h
def worst(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = ow(self, *a, **k); torch.cuda.synchronize()
    rows[-1]["screens_ms"] = round(rows[-1].get("screens_ms", 0.0) + 1e3 * (time.perf_counter() - t), 2)
    return r
SB.__init__, SB.optimize, SB.worst_residuals = init, opt, worst
run_icl_nuim.run(200, bundle_adjust="keyframe", seed=0)
rows.clear()
out = run_icl_nuim.run(200, bundle_adjust="keyframe", seed=0)
for r in rows[::2]:
    print(json.dumps(r))
print(out["frames_per_s"])
