#!/usr/bin/env python3
"""Timing of the sparse BA launches on a synthetic problem shaped like the reference's largest data set
(ICL-NUIM kt2: 881 poses, 13 293 landmarks, 231 120 observations).  python tools/bench_sparse_ba.py [P N track]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mqslam_amd


def run(P=881, N=13293, track=17):
    rng = np.random.default_rng(0)
    # camera moving along x, looking down +z at a slab of points 8..12 units away
    poses = np.zeros((P, 12)); poses[:, [0, 4, 8]] = 1.0; poses[:, 9] = np.linspace(0, 0.05 * P, P)
    pts = np.stack([rng.uniform(-1, 0.05 * P + 1, N), rng.uniform(-3, 3, N), rng.uniform(8, 12, N)], 1)
    K = np.array([[481.2, 480.0, 0, 319.5, 239.5, 0, 0, 0, 0]])
    c = np.clip((pts[:, 0] / 0.05).astype(np.int64), 0, P - 1)
    a = np.maximum(0, np.minimum(P - track, c - track // 2))
    op = (a[:, None] + np.arange(track)[None, :]).reshape(-1)                        # track consecutive poses per landmark
    q = np.repeat(pts, track, axis=0) - poses[op, 9:]
    uv = np.stack([481.2 * q[:, 0] / q[:, 2] + 319.5, 480.0 * q[:, 1] / q[:, 2] + 239.5], 1) + rng.normal(0, 0.5, (len(op), 2))
    ptr = np.arange(N + 1, dtype=np.int64) * track
    SP = mqslam_amd.ba_io.SparseProblem
    pr = SP(poses=poses + 0, pose_cam=np.zeros(P, np.int32), pose_key=[(0, j) for j in range(P)], calib=K, sigma=np.array([1.0]),
            points=pts + rng.normal(0, 0.02, pts.shape), obs_ptr=ptr, obs_pose=op.astype(np.int32), obs_uv=uv,
            prior_w=np.where(np.arange(N) < 100, 16.0, 0.0), prior_xyz=pts.copy(), pose_prior_idx=np.array([0], np.int32),
            pose_prior_sigmas=np.array([[0.002] * 3 + [0.001] * 3]), odo_from=np.zeros(0, np.int32), odo_to=np.zeros(0, np.int32),
            odo_meas=np.zeros((0, 12)), odo_sigmas=np.zeros((0, 6)))
    t0 = time.time(); ba = mqslam_amd.sparse_ba.SparseBundleAdjuster(pr); t_setup = time.time() - t0
    # the set-up again, with the kernels loaded and the allocator warm: what a second bundle adjustment in a session pays;
    # and its pieces: the sort of the observations (device; the numpy twin beside it), the pair grouping on the device (with its
    # one synchronisation)
    del ba
    torch.cuda.synchronize()
    t0 = time.time(); ba = mqslam_amd.sparse_ba.SparseBundleAdjuster(pr); torch.cuda.synchronize(); t_setup_warm = time.time() - t0
    S = mqslam_amd.sparse_ba
    t0 = time.time(); prs = S.sort_observations_by_pose(pr); t_sort = time.time() - t0          # the numpy statement (round 3's set-up path)
    dptr, dpose = torch.from_numpy(np.asarray(prs.obs_ptr)).cuda(), torch.from_numpy(np.asarray(prs.obs_pose)).cuda()
    upose, uuv = torch.from_numpy(np.asarray(pr.obs_pose)).cuda(), torch.from_numpy(np.ascontiguousarray(pr.obs_uv)).cuda()
    S.sort_observations_dev(dptr, upose, uuv, N, P); torch.cuda.synchronize()
    t0 = time.time(); S.sort_observations_dev(dptr, upose, uuv, N, P); torch.cuda.synchronize(); t_sort_dev = time.time() - t0
    S.group_pairs_dev(prs.obs_ptr, dptr, dpose, P); torch.cuda.synchronize()
    t0 = time.time(); S.group_pairs_dev(prs.obs_ptr, dptr, dpose, P); torch.cuda.synchronize(); t_group = time.time() - t0

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 3)

    out = {"P": P, "N": N, "M": len(op), "pairs": ba.Q, "pose_pair_groups": ba.G, "setup_s": round(t_setup, 3),
           "setup_s_second_construction": round(t_setup_warm, 4), "setup_pieces_ms": {"sort_observations_device": round(1e3 * t_sort_dev, 3),
                                                                                      "sort_observations_numpy_not_on_the_path": round(1e3 * t_sort, 2),
                                                                                      "group_pairs_device": round(1e3 * t_group, 3)},
           "half_bandwidth": ba.half_bandwidth, "n": ba.n6}
    out["linearize_ms"] = timed(lambda: ba.linearize(1e-4))
    def lin_solve(): ba.linearize(1e-4); ba.solve(1e-4)
    out["linearize+solve_ms"] = timed(lin_solve)
    out["backsub_ms"] = timed(lambda: ba.backsub(1e-4))
    out["cost_ms"] = timed(lambda: ba.cost())
    t0 = time.time(); hist = ba.optimize(mode="lm"); out["lm_ms"] = round((time.time() - t0) * 1e3, 2); out["lm_iters"] = len(hist) - 1
    import ctypes
    need = mqslam_amd._lib.lib().mqs_sba_solve_plan_dump(ba.n6, ba.half_bandwidth, int(os.environ.get("MQS_SBA_PARTS", "0") or 0), None, 0)
    if need:
        buf = np.zeros(need, np.int32)
        mqslam_amd._lib.lib().mqs_sba_solve_plan_dump(ba.n6, ba.half_bandwidth, int(os.environ.get("MQS_SBA_PARTS", "0") or 0),
                                                       buf.ctypes.data_as(ctypes.c_void_p), need)
        st = buf[8:8 + 8 * int(buf[0])].reshape(-1, 8)
        out["solve_order"] = {"chunks": int(buf[6]), "levels": int(buf[0]), "dependent_factor_steps": int(st[:, 1].sum()),
                              "block_columns": (ba.n6 + 31) // 32}
    else:
        out["solve_order"] = {"chunks": 1, "dependent_factor_steps": (ba.n6 + 31) // 32, "block_columns": (ba.n6 + 31) // 32}
    out["cost0"], out["cost1"] = hist[0], hist[-1]
    return out


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:4]]
    print(json.dumps(run(*a)))
