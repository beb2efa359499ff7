#!/usr/bin/env python3
"""Shader cycles the waves of ba_linearize_wave_kernel<4> spend waiting for their loads (a probe build of the library:
tools/build_variant.sh probe ba.hip -DMQS_WL_PROBE_WAIT=1 [-DMQS_WL_PREFETCH=0]; MQS_LIB_PATH=build/ab/libmqslam_probe.so)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
L = mqslam_amd._lib.lib()
buf = (ctypes.c_ulonglong * 4096)()
for _ in range(20):
    ba.linearize(0.0)
torch.cuda.synchronize()
L.mqs_debug_wl_probe(None, 1)
reps = 10
for _ in range(reps):
    ba.linearize(0.0)
torch.cuda.synchronize()
L.mqs_debug_wl_probe(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, 1024).astype(np.float64) / reps
out = {"lib": os.path.basename(mqslam_amd._lib.LIB_PATH), "N": N}
for k, name in ((0, "head_wait"), (1, "camera_obs_wait"), (3, "chunks_total")):
    out[name + "_cycles_per_wave"] = {"mean": round(a[k].mean()), "min": round(a[k].min()), "max": round(a[k].max())}
print(json.dumps(out))
