#!/usr/bin/env python3
"""Round 6: several loops at once on one MI355X -- one handle and one host thread per camera (the reference is a MULTIPLE-quadrotor system; a loop
by itself keeps one to 258 small workgroups busy, i.e. a fraction of the device).  Every camera runs the 200 example frames (its own seed);
aggregate frames/s = cameras x frames / wall time of the slowest; the trajectories have to be the ones the cameras produce alone.
python tools/probes/multi_quadrotor_study.py [max_cameras=8] [frames=200]"""
import os, sys, json, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mqslam_amd, run_icl_nuim as R
kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = R.load_sequence(frames)
imgs_h = d["frames"][:frames]
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = imgs_h.shape[1:]
uv, vis = R.start_points(K, (H, W), P_init, pts)
objp, imgp = pts[vis], uv[vis]
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in imgs_h]          # the cameras share the frames (read-only): what differs is the seed
torch.cuda.synchronize()

def one(seed, ba, out, start):
    slam = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=seed, bundle_adjust=ba, max_homography_points="reference")
    start.wait()
    t0 = time.perf_counter()
    slam.start(imgs[0], objp, imgp)
    n = len(imgs)
    for k in range(1, n):
        slam.handle_new_frame(imgs[k], imgs[k + 1] if k + 1 < n else None)
    slam.finish()
    dt = time.perf_counter() - t0
    c = np.array([(-P[:, :3].T @ P[:, 3]) if P is not None else [np.nan] * 3 for P in slam.projection_matrices()])
    out[seed] = (dt, c)
    slam.close()

R.run(80)
for ba in (None, "keyframe"):
    alone = {}
    for seed in range(kmax):
        o = {}
        ev = threading.Event(); ev.set()
        one(seed, ba, o, ev)
        alone[seed] = o[seed]
    row = {"ba": ba, "alone_frames_per_s_median": round(float(np.median([frames / alone[s][0] for s in alone])), 1)}
    for k in (1, 2, 4, 8, 12, 16):
        if k > kmax:
            break
        best = None
        for rep in range(3):
            out, ev = {}, threading.Event()
            th = [threading.Thread(target=one, args=(s, ba, out, ev)) for s in range(k)]
            for t in th: t.start()
            time.sleep(0.05)
            t0 = time.perf_counter(); ev.set()
            for t in th: t.join()
            wall = time.perf_counter() - t0
            same = all(np.array_equal(out[s][1], alone[s][1], equal_nan=True) for s in range(k))
            if best is None or wall < best[0]:
                best = (wall, same)
        row["%d cameras" % k] = {"aggregate_frames_per_s": round(k * frames / best[0], 1), "same_trajectories_as_alone": bool(best[1])}
    print(json.dumps(row), flush=True)
