#!/usr/bin/env python3
"""Round 6: phase stamps of EVERY adjustment of a run on the reference's example sequence with the selection form
(mqs_slam_bundle_adjust_window): per adjustment poses / passes / trials / ms and the phases' microseconds.
    python tools/probes/icl_selection_stamps.py [frames=200] [window=3] [seed=0]"""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mqslam_amd, run_icl_nuim
from mqslam_amd import _lib
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
window = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = run_icl_nuim.load_sequence(frames)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"][:frames]]
names = {0: "start", 1: "cleared", 2: "log->table", 3: "lists", 8: "hit-lists", 4: "lm-begin", 5: "screen", 6: "write-back", 7: "end", 10: "trial", 11: "records",
         12: "barrier", 13: "system", 14: "barrier", 15: "cholesky", 16: "backsolve", 17: "barrier", 18: "landmarks+cost", 19: "cost-reduced", 20: "sys-task-begin",
         21: "sys-accumulated", 22: "sys-reduced"}
for rep in range(2):
    s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, bundle_adjust="keyframe", max_homography_points="reference",
                                              ba_window_keyframes=window or None)
    s.ba_window_point_sigma = 0.0
    s.start(imgs[0], pts[vis], uv[vis])
    n = ctypes.c_int32(0)
    _lib.check(_lib.lib().mqs_debug_slam_ba_stamps(s._h, None, 0, ctypes.byref(n)))
    inner = s._bundle_adjust
    rows = []

    def adjust():
        inner()
        buf = np.zeros((2048, 2), np.int64)
        _lib.check(_lib.lib().mqs_debug_slam_ba_stamps(s._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 2048, ctypes.byref(n)))
        st = buf[:min(n.value, 2048)]
        t = (st[:, 1] - st[0, 1]) / 100.0
        agg = {}
        for k in range(1, len(st)):
            if int(st[k, 0]) in (20, 21, 22):
                continue
            agg.setdefault(names.get(int(st[k, 0]), str(st[k, 0])), []).append(t[k] - t[k - 1])
        r = s.ba_reports[-1]
        rows.append({"frame": r["frame"], "poses": r["poses"], "landmarks": r["landmarks"], "adjusted": r["landmarks_adjusted"], "observations": r["observations"],
                     "passes": r["passes"], "trials": r["lm_trials"], "iterations": r["lm_iterations"], "barriers": r["grid_barriers"],
                     "adjust_ms": r["adjust_ms"], "kernel_us": round(float(t[-1]), 1),
                     "phase_us": {k: round(float(np.sum(v)), 1) for k, v in agg.items()}})
    s._bundle_adjust = adjust
    for k in range(1, frames):
        s.handle_new_frame(imgs[k])
    s.finish()
    s.close()
for r in rows:
    print(json.dumps(r))
