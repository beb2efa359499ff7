#!/usr/bin/env python3
"""Timing of the image front-end on rendered frames (device-resident: kernels only; and the host-pointer calls)."""
import os, sys, json, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, mqslam_amd
from test_features import texture
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
I = texture(H, W, seed=1, blobs=1200)
J = texture(H, W, shift=(2.3, -1.1), seed=1, blobs=1200)
F = mqslam_amd.features
L = mqslam_amd._lib
pts = F.goodFeaturesToTrack(I, 300, 0.01, 7.0)
out = {"image": [H, W], "corners": len(pts)}
def timed_host(fn, reps=20):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)
out["gftt_host_call_us"] = timed_host(lambda: F.goodFeaturesToTrack(I, 300, 0.01, 7.0))
out["lk_host_call_us"] = timed_host(lambda: F.calcOpticalFlowPyrLK(I, J, pts))
dev = torch.device("cuda", 0)
dI, dJ, dP = torch.from_numpy(I).to(dev), torch.from_numpy(J).to(dev), torch.from_numpy(pts).to(dev)
ws1 = torch.empty(int(L.lib().mqs_gftt_workspace_bytes(W, H)), dtype=torch.uint8, device=dev)
ws2 = torch.empty(int(L.lib().mqs_lk_workspace_bytes(W, H, 3)), dtype=torch.uint8, device=dev)
oxy = torch.empty((300, 2), dtype=torch.float32, device=dev); on = torch.zeros(1, dtype=torch.int32, device=dev)
nq = torch.empty((len(pts), 2), dtype=torch.float32, device=dev); st = torch.empty(len(pts), dtype=torch.uint8, device=dev)
er = torch.empty(len(pts), dtype=torch.float32, device=dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def gftt():
    L.check(L.lib().mqs_good_features_to_track_dev(dI.data_ptr(), W, H, 300, ctypes.c_double(0.01), ctypes.c_double(7.0), None,
                                                   oxy.data_ptr(), 300, on.data_ptr(), ws1.data_ptr(), ws1.numel(), sp))
def lk():
    L.check(L.lib().mqs_calc_optical_flow_pyr_lk_dev(dI.data_ptr(), dJ.data_ptr(), W, H, dP.data_ptr(), len(pts), 21, 21, 3, 30,
                                                     ctypes.c_double(0.01), ctypes.c_double(1e-4), nq.data_ptr(), st.data_ptr(),
                                                     er.data_ptr(), ws2.data_ptr(), ws2.numel(), sp))
def timed_dev(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)
out["gftt_dev_us"] = timed_dev(gftt)
out["lk_dev_us"] = timed_dev(lk)
print(json.dumps(out))
