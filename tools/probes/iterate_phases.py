#!/usr/bin/env python3
"""Where a launch of ba_iterate_kernel spends its time, per workgroup: 100 MHz wall-clock stamps at the phase boundaries of the LAST
launch of a run (probe build: tools/build_variant.sh itprobe ba.hip -DMQS_ITERATE_PROBE=1; MQS_LIB_PATH=build/ab/libmqslam_itprobe.so)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
lib = ctypes.CDLL(mqslam_amd._lib.LIB_PATH)
ba.gauss_newton_iterations(60)
ba.gauss_newton_iterations(41)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (8 * 256))()
assert lib.mqs_debug_iterate_probe(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(8, 256).astype(np.float64) * 0.01        # us
t0 = a[0].min()
names = ["start", "finalizer pieces done", "flags seen", "solved", "back-substituted", "cameras restaged", "linearised", "row written"]
out = {"N": N, "phases_us_since_first_workgroup_start": {}}
for k, nm in enumerate(names):
    v = a[k] - t0
    out["phases_us_since_first_workgroup_start"][nm] = {"median": round(float(np.median(v)), 2), "min": round(float(v.min()), 2), "max": round(float(v.max()), 2),
                                                       "finalizers_median": round(float(np.median(v[:48])), 2), "others_median": round(float(np.median(v[48:])), 2)}
print(json.dumps(out))
