import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch, mqslam_amd, run_icl_nuim
d = run_icl_nuim.load_sequence(80)
K, dist, P_init, pts = d["K"], d["dist"], d["init_pose"], d["init_points"]
H, W = d["frames"].shape[1:]
uv, vis = run_icl_nuim.start_points(K, (H, W), P_init, pts)
imgs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in d["frames"][:80]]
for rep in range(2):
    s = mqslam_amd.slam_device.DeviceMonoSlam(K, dist, (H, W), seed=0, bundle_adjust="keyframe", max_homography_points="reference")
    s.start(imgs[0], pts[vis], uv[vis])
    for k in range(1, 80):
        s.handle_new_frame(imgs[k])
    s.finish()
    for r in s.ba_reports: print(json.dumps({k: r[k] for k in ("frame","poses","landmarks","observations","passes","lm_trials","lm_iterations","grid_barriers","adjust_ms","cost_before","cost_after")}))
    s.close()
