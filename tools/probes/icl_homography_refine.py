#!/usr/bin/env python3
"""keyframe_test's ratio w0 / w2 on the device loop's real tracks (reference example sequence): normalised DLT alone against DLT +
the Levenberg-Marquardt refinement of the transfer error that cv2.findHomography(method=0) applies (fundam.cpp: estimator.refine(M, m, H, 10))."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, torch, mqslam_amd, run_icl_nuim
from mqslam_amd.slam_loop import homography_dlt

def refine(H, p1, p2, iters=10):
    h = (H / H[2, 2]).ravel()[:8].copy()
    lam = 1e-3
    def res(h):
        Hm = np.append(h, 1.0).reshape(3, 3)
        q = np.c_[p1, np.ones(len(p1))] @ Hm.T
        return (q[:, :2] / q[:, 2:3] - p2).ravel(), q
    r, q = res(h)
    for _ in range(iters):
        w = 1.0 / q[:, 2]
        x, y = q[:, 0] * w, q[:, 1] * w
        J = np.zeros((2 * len(p1), 8))
        J[0::2, 0:2], J[0::2, 2] = p1 * w[:, None], w
        J[0::2, 6:8] = -p1 * (x * w)[:, None]
        J[1::2, 3:5], J[1::2, 5] = p1 * w[:, None], w
        J[1::2, 6:8] = -p1 * (y * w)[:, None]
        A, g = J.T @ J, J.T @ r
        while True:
            step = np.linalg.solve(A + lam * np.diag(np.diag(A)), -g)
            r2, q2 = res(h + step)
            if r2 @ r2 < r @ r or lam > 1e10:
                break
            lam *= 10
        if r2 @ r2 < r @ r:
            h, r, q, lam = h + step, r2, q2, lam / 10
    return np.append(h, 1.0).reshape(3, 3)

d = np.load(run_icl_nuim.FIX)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"]]
s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=0, max_homography_points=0)
s.start(imgs[0], pts[vis], uv[vis])
rng = np.random.RandomState(0)
for k in range(1, 60):
    r = s.handle_new_frame(imgs[k])
    rep = s.reports[-1]
    s.finish()
    if r == 2:
        print(k, "KEYFRAME (kernel ratio %.4f)" % rep[10]); continue
    p, b, lm, tid = s.tracks()
    u1 = mqslam_amd.camera.undistort_points(b.astype(np.float64), K, dist); u2 = mqslam_amd.camera.undistort_points(p.astype(np.float64), K, dist)
    H0 = homography_dlt(u1, u2); H1 = refine(H0, u1, u2)
    w0 = np.linalg.svd(H0, compute_uv=False); w1 = np.linalg.svd(H1, compute_uv=False)
    sub = [ (lambda i: (np.linalg.svd(homography_dlt(u1[i], u2[i]), compute_uv=False), np.linalg.svd(refine(homography_dlt(u1[i], u2[i]), u1[i], u2[i]), compute_uv=False)))(rng.permutation(len(u1))[:75]) for _ in range(5)]
    print(k, "n", len(p), "kernel %.4f | all: dlt %.4f refined %.4f | 75-samples dlt %s refined %s" % (rep[10], w0[0] / w0[2], w1[0] / w1[2], [round(a[0] / a[2], 4) for a, _ in sub], [round(bb[0] / bb[2], 4) for _, bb in sub]))
