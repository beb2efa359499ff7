"""
world_size-2 `gloo` test of the multi-GPU contract on CPU: landmarks are sharded
with the product's sharding helpers, each rank linearises ITS shard (the oracle stands in for the
HIP kernel, which cannot run here), the reduced camera systems are summed with the product's
all-reduce helper, every rank solves the same system, and the result equals the unsharded solve to
<= 1e-10 (sum order only).  Triangulation shards need no collective: checked by slicing.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import mqslam_amd
    from oracle import ba_np
    from ba_util import make_scene
    r, lr, w = mqslam_amd.sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sc = make_scene(101, 3, seed=4, distortion=True)           # 101: uneven shards
    pts, obs, mask, pw, px = mqslam_amd.sharding.shard_arrays(rank, world, sc["points"], sc["obs"], sc["mask"],
                                                              sc["prior_w"], sc["prior_xyz"])
    S, g, cost, nv, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], pts, obs, mask, pw, px)
    lin = torch.from_numpy(np.concatenate([S.reshape(-1), g, [cost, nv]]))
    lin_sync = mqslam_amd.sharding.all_reduce_sum_(lin.clone())
    # the overlapped form the GN iteration uses: start the reduce, do unrelated work, wait
    work = mqslam_amd.sharding.all_reduce_sum_(lin, async_op=True)
    assert work is not None
    unrelated = float(np.square(obs).sum())
    work.wait()
    assert torch.equal(lin, lin_sync) and np.isfinite(unrelated)
    n6 = 18
    Sr = lin[:n6 * n6].numpy().reshape(n6, n6)
    gr = lin[n6 * n6:n6 * n6 + n6].numpy()
    dpose = np.linalg.solve(Sr + 1e-6 * np.eye(n6), gr)      # identical on every rank: no broadcast
    dp = ba_np.backsub(pieces, dpose)                          # each rank updates only its own landmarks
    q.put((rank, lin.numpy().copy(), dpose, dp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_schur_allreduce_equals_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import ba_np
    from ba_util import make_scene
    import mqslam_amd
    world, port = 2, 29533 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = make_scene(101, 3, seed=4, distortion=True)
    S, g, cost, nv, pieces = ba_np.linearize(sc["poses"], sc["calib"], sc["sigma"], sc["points"], sc["obs"],
                                             sc["mask"], sc["prior_w"], sc["prior_xyz"])
    ref = np.concatenate([S.reshape(-1), g, [cost, nv]])
    for rank, lin, dpose, dp in res:
        assert np.abs(lin - ref).max() <= 1e-10 * np.abs(ref).max()
    np.testing.assert_array_equal(res[0][2], res[1][2])                    # same solve on every rank
    dpose = np.linalg.solve(S + 1e-6 * np.eye(18), g)
    dp_ref = ba_np.backsub(pieces, dpose)
    dp_all = np.concatenate([res[0][3], res[1][3]])
    assert np.abs(dp_all - dp_ref).max() <= 1e-9 * max(1.0, np.abs(dp_ref).max())
    # shard ranges tile [0, N) exactly
    spans = [mqslam_amd.sharding.landmark_shard(101, r, 2) for r in range(2)]
    assert spans == [(0, 50), (50, 101)]
    for w in (1, 3, 8):
        sp = [mqslam_amd.sharding.landmark_shard(1_000_003, r, w) for r in range(w)]
        assert sp[0][0] == 0 and sp[-1][1] == 1_000_003 and all(a[1] == b[0] for a, b in zip(sp, sp[1:]))


def test_triangulation_shards_need_no_collective(c_oracle):
    import mqslam_amd
    u, P, _ = mqslam_amd.synthetic.triangulation_problem(1001, 4)
    x, s = c_oracle.iterative_LS_triangulation(u, P)
    parts = []
    for r in range(3):
        a, b = mqslam_amd.sharding.landmark_shard(1001, r, 3)
        parts.append(c_oracle.iterative_LS_triangulation(np.ascontiguousarray(u[:, a:b]), P))
    np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), x)
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), s)
