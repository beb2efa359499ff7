"""
The multi-rank path on the real kernels: two ranks share the one GPU of the test
box (transport gloo -- one device cannot host two RCCL ranks; sharding.init_from_env's test hooks),
each holds its landmark shard, and three Gauss-Newton iterations with the overlapped all-reduce of
the reduced camera system give the poses and landmarks of the single-rank run to <= 1e-10.
Also: bench.py under torch.distributed.run prints the contract's JSON line with n_gpus = 2.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import mqslam_amd
from ba_util import make_scene
rank, local_rank, world = mqslam_amd.sharding.init_from_env()
dev = torch.device("cuda", 0)
sc = make_scene(4001, 4, seed=11)
pts, obs, mask, pw, px = mqslam_amd.sharding.shard_arrays(rank, world, sc["points"], sc["obs"], sc["mask"],
                                                          sc["prior_w"], sc["prior_xyz"])
t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
C = 4
pose_prior = (t(sc["poses_true"]), t(np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1))),
              t(np.array([1, 0, 0, 0], dtype=np.uint8), torch.uint8))
ba = mqslam_amd.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(pts), t(obs), None,
                                                 t(pw), t(px), pose_prior, True if world > 1 else None)
u, P, _ = mqslam_amd.synthetic.triangulation_problem(5000, 4)
ud, Pd = torch.from_numpy(u).to(dev), torch.from_numpy(np.ascontiguousarray(P)).to(dev)
xo = torch.empty((5000, 3), dtype=torch.float64, device=dev)
calls = []
def other_work():
    calls.append(1)
    mqslam_amd.device.linear_LS_triangulation(ud, Pd, out=xo)
costs = []
for _ in range(3):
    costs.append(ba.total_cost())
    ba.gauss_newton_iteration(overlap=other_work)
costs.append(ba.total_cost())
assert len(calls) == 3
np.savez(os.path.join({out!r}, "r%d_of_%d.npz" % (rank, world)), poses=ba.poses.cpu().numpy(),
         points=ba.points.cpu().numpy(), costs=np.array(costs), x=xo.cpu().numpy())
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
"""


def _run(world, tmp_path, script):
    env = dict(os.environ, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    if world == 1:
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def test_two_ranks_equal_one_rank(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    _run(1, tmp_path, str(script))
    _run(2, tmp_path, str(script))
    one = np.load(tmp_path / "r0_of_1.npz")
    two = [np.load(tmp_path / ("r%d_of_2.npz" % r)) for r in range(2)]
    np.testing.assert_array_equal(two[0]["poses"], two[1]["poses"])              # same solve on every rank, no broadcast
    assert np.abs(two[0]["poses"] - one["poses"]).max() <= 1e-10
    pts = np.concatenate([two[0]["points"], two[1]["points"]])
    assert pts.shape == one["points"].shape
    assert np.abs(pts - one["points"]).max() <= 1e-10 * max(1.0, np.abs(one["points"]).max())
    assert np.allclose(two[0]["costs"], one["costs"], rtol=1e-10, atol=0)
    assert one["costs"][-1] < one["costs"][0]
    np.testing.assert_array_equal(two[0]["x"], one["x"])                         # the overlapped work is untouched


def _bench_lines(r):
    """bench.py's output: (the ONE compact contract line of stdout, every leg in full from the one JSON line of stderr)."""
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    full = [l for l in r.stderr.splitlines() if l.startswith('{"metric"')]
    assert len(full) == 1, r.stderr[-2000:]
    return json.loads(lines[0]), json.loads(full[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_contract_two_ranks(gpu, tmp_path, scaling):
    """Plain `python bench.py --gpus 2` (no launcher around it): the bench starts its own two ranks, rank 0 prints ONE
    line with n_gpus = 2, and the sharded configs[3] problem reproduces the one-rank poses (ba_strong.ok)."""
    cmd = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--landmarks", "20000", "--strong-landmarks", "30001",
           "--scaling", scaling, "--no-match", "--no-replay", "--no-frontend", "--no-cpu-baseline"]
    env = dict(os.environ, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + cmd, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    head, out = _bench_lines(r)                                                  # rank 0 prints ONE line on stdout; the legs in full on stderr
    assert head["n_gpus"] == 2 and head["scaling"] == scaling and head["ba_strong"]["ok"] is True
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] > 0
    if scaling == "weak":
        assert out["config"]["landmarks_per_gpu"] == 20000 and out["config"]["landmarks_total"] == 40000
        assert out["ba"]["landmarks_total"] == 40000
    else:
        assert out["config"]["landmarks_per_gpu"] == 10000 and out["config"]["landmarks_total"] == 20000
    bs = out["ba_strong"]
    assert bs["ok"] is True and bs["poses_identical_on_all_ranks"] and bs["max_abs_pose_diff_vs_one_rank"] <= 1e-10
    assert bs["rccl_world_size"] == 2 and bs["cost_after"] < bs["cost_before"]
    assert bs["cost_after"] == pytest.approx(bs["cost_after_one_rank"], rel=1e-9)
    assert "torch.distributed(gloo)" in out["transport"]
    for key in ("roofline", "cpu_baseline", "vs_baseline", "dtype", "unit", "metric", "ms_per_step"):
        assert key in out


def test_bench_contract_single_gpu(gpu):
    """`python bench.py` (N = 1) at a reduced size: ONE JSON line with the contract's keys, the roofline object of the
    dominant kernel and the CPU baseline timed beside it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--landmarks", "50000",
                        "--descriptors", "4096", "--no-replay", "--no-frontend"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    head, out = _bench_lines(r)
    assert len(json.dumps(head)) < 6000                                          # the driver keeps an 8 KB tail: the contract line fits whole
    for key in ("metric", "value", "value_cold", "ms_per_step", "ms_per_step_cold", "config", "dtype", "roofline", "cpu_baseline", "ba", "match", "loop"):
        assert key in head, key
    assert head["value"] == out["value"] and head["roofline"]["frac"] == out["roofline"]["frac"] and head["cpu_baseline"]["value"] == out["cpu_baseline"]["value"]
    assert head["cpu_baseline"]["ba_gn_iters_per_s_all_cores"] > 0 and "all 50000 landmarks" in out["cpu_baseline"]["ba"]["sample"]
    for key, val in (("n_gpus", 1), ("steps", 3), ("warmup", 1), ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None),
                     ("dtype", "f64"), ("data", "synthetic"), ("unit", "landmarks/s")):
        assert out[key] == val, key
    assert out["value"] > 0 and out["ms_per_step"] > 0 and "workload" in out["config"]
    assert out["ms_per_step_cold"] > 0 and out["value_cold"] > 0 and "value_from" in out
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "fp64_valu") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-2) and rf["achieved"] > 0     # both are rounded
    assert rf["hbm"]["peak"] == 8000.0 and rf["hbm"]["frac"] == pytest.approx(rf["hbm"]["achieved"] / 8000.0, rel=1e-2)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "landmarks/s" and "sample" in cb
    assert cb["parity"]["rel_err_p99.9"] < 1e-5 and cb["parity"]["status_mismatch_frac"] < 0.005
    assert out["match"]["packed_bits_fp4"]["equals_fp16_path"] is True
    assert out["match"]["cross_match_4_cameras"]["pairs_this_rank"] == 6


def test_nccl_backend_plumbing_single_rank(gpu, tmp_path):
    """What a 1-GPU box can run of the N-GPU path with the REAL backend: a torch.distributed group on `nccl` (= RCCL) with one
    rank, the communicator id agreed and broadcast over it (sharding.init_c_comm), the library's own RCCL communicator, a
    sharded-API BundleAdjuster on it, and bench.py's transport probe.  (Two RCCL ranks cannot share one device.)"""
    script = tmp_path / "nccl1.py"
    script.write_text(r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() %% 200))
import torch, torch.distributed as dist
import mqslam_amd
from ba_util import make_scene
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cc = mqslam_amd.sharding.init_c_comm(0, 1, 0)
probe = torch.arange(602, dtype=torch.float64, device="cuda")
want = probe.clone(); dist.all_reduce(want)
cc.all_reduce_sum_(probe); torch.cuda.synchronize()
assert torch.equal(probe, want)
sc = make_scene(500, 4, seed=2)
t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt)
mk = lambda pg: mqslam_amd.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(sc["points"]), t(sc["obs"]),
                                                           None, t(sc["prior_w"]), t(sc["prior_xyz"]), None, pg)
a, b, c = mk(cc), mk(True), mk(None)
for ba in (a, b, c):
    ba.gauss_newton_iterations(3)
torch.cuda.synchronize()
assert torch.equal(a.poses, c.poses) and torch.equal(b.poses, c.poses) and torch.equal(a.points, c.points)
# ncclAllReduce enqueued by the library between the lineariser and the solve (a) == torch's all_reduce between the two halves (b)
assert torch.equal(a.lin, b.lin) and torch.equal(a.lin, c.lin)
assert a.total_cost() == c.total_cost() == b.total_cost()
dist.barrier(); cc.close(); dist.destroy_process_group()
print("nccl-ok")
""" % (ROOT, ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0 and "nccl-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_multi_gpu_code_path_on_the_real_backend(gpu):
    """bench.py's N-GPU branch (nccl process group, the library's RCCL communicator verified against torch's sum, the one-call
    iteration with the collective issued from C, the ba_strong leg) with ONE rank: everything a 1-GPU box can run of what
    `python bench.py --gpus 8` runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MQS_DIST_BACKEND", "MQS_SHARED_GPU")}
    env.update(MQS_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--landmarks", "20000",
                        "--strong-landmarks", "30001", "--no-match", "--no-replay", "--no-frontend", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    head, out = _bench_lines(r)
    assert out["n_gpus"] == 1 and out["transport"].startswith("peer stores over xGMI via the C ABI"), out["transport"]
    bs = out["ba_strong"]
    assert bs["ok"] is True and bs["backend"] == "nccl" and bs["rccl_world_size"] == 1 and bs["all_reduce_us"] > 0
    assert bs["max_abs_pose_diff_vs_one_rank"] == 0.0          # one rank: the sharded run IS the one-rank run
    # both transports of the sharded leg timed in the same run, each verified: the peer stores and the RCCL all-reduce north_star names
    tr = bs["ba_strong_transports"]
    assert tr["peer_stores"]["verified"] and tr["rccl_all_reduce"]["verified"] and tr["chosen"] in ("peer_stores", "rccl_all_reduce")
    assert tr["peer_stores"]["ms_per_iter"] > 0 and tr["rccl_all_reduce"]["ms_per_iter"] > 0
    assert bs["ms_per_iter"] == min(tr["peer_stores"]["ms_per_iter"], tr["rccl_all_reduce"]["ms_per_iter"])
    assert head["ba_strong"]["ba_strong_transports"] == tr


PEER_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import mqslam_amd
from ba_util import make_scene
sh = mqslam_amd.sharding
rank, local_rank, world = sh.init_from_env()
dev = torch.device("cuda", 0)
cc = sh.init_peer_comm(rank, world, 0)
assert cc.peer_state() == (1 if world > 1 else 2), cc.peer_state()       # two processes on one GPU: stand-alone wait kernel
# (a) the collective by itself: 40 back-to-back reductions of exactly representable values
for it in range(40):
    n = 602 if it % 2 == 0 else 2
    buf = (torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * it) * (rank + 1)
    cc.all_reduce_sum_(buf)
    want = (torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * it) * (world * (world + 1) // 2)
    torch.cuda.synchronize()
    assert torch.equal(buf, want), (it, rank)
# (b) Gauss-Newton on the sharded scene, one library call per iteration: the finalize kernel stores the rank's reduced system
#     into both receive buffers, the consumer adds the rows in rank order
sc = make_scene(4001, 4, seed=11, distortion=True)
pts, obs, mask, pw, px = sh.shard_arrays(rank, world, sc["points"], sc["obs"], sc["mask"], sc["prior_w"], sc["prior_xyz"])
t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
C = 4
pose_prior = (t(sc["poses_true"]), t(np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1))),
              t(np.array([1, 0, 0, 0], dtype=np.uint8), torch.uint8))
ba = mqslam_amd.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(pts), t(obs), None,
                                                 t(pw), t(px), pose_prior, cc)
costs = [ba.total_cost()]
ba.gauss_newton_iterations(2)
ba.gauss_newton_iteration()
costs.append(ba.total_cost())
hist = ba.optimize(iters=6, mode="lm")                   # the split form: linearise, all-reduce, solve, back-substitute
torch.cuda.synchronize()
assert not cc.peer_timed_out()
np.savez(os.path.join({out!r}, "peer_{tag}_r%d_of_%d.npz" % (rank, world)), poses=ba.poses.cpu().numpy(),
         points=ba.points.cpu().numpy(), costs=np.array(costs + hist), lin=ba.lin.cpu().numpy())
if world > 1:
    torch.distributed.barrier()
cc.close()
if world > 1:
    torch.distributed.destroy_process_group()
print("peer-ok")
"""


def test_peer_transport_two_processes_on_one_gpu(gpu, tmp_path):
    """The one-shot all-reduce over peer-mapped receive buffers (csrc/comm.hip) with two PROCESSES on the one GPU of the test
    box: the buffers are exchanged as hipIpc handles over the gloo group, every reduction is the ranks' rows added in rank
    order.  Checked: the collective alone (exact integers), the one-call Gauss-Newton iteration whose finalize kernel is the
    send side, the LM loop on the split form -- against the same scene solved by ONE process, with the stand-alone wait
    kernel (the default when ranks share a GPU) and with the wait fused into the solve / back-substitution kernel
    (MQS_PEER_FUSED=1: what ranks on separate GPUs run; small enough here that the waiting kernel does not fill the chip)."""
    outs = {}
    for tag, fused in (("gather", "0"), ("fused", "1")):
        script = tmp_path / ("peer_%s.py" % tag)
        script.write_text(PEER_WORKER.format(root=ROOT, out=str(tmp_path), tag=tag))
        env = dict(os.environ, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", MQS_PEER_FUSED=fused)
        for world in (1, 2):
            if world == 1:
                cmd = [sys.executable, str(script)]
                e = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MQS_PEER_FUSED")}
            else:
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                       "127.0.0.1", "--master-port", str(29300 + os.getpid() % 250), str(script)]
                e = env
            r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "peer-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        one = np.load(tmp_path / ("peer_%s_r0_of_1.npz" % tag))
        two = [np.load(tmp_path / ("peer_%s_r%d_of_2.npz" % (tag, r))) for r in range(2)]
        np.testing.assert_array_equal(two[0]["poses"], two[1]["poses"])          # rank-ordered sum: the same bits on every rank
        np.testing.assert_array_equal(two[0]["lin"], two[1]["lin"])
        assert np.abs(two[0]["poses"] - one["poses"]).max() <= 1e-9
        pts = np.concatenate([two[0]["points"], two[1]["points"]])
        assert np.abs(pts - one["points"]).max() <= 1e-9 * max(1.0, np.abs(one["points"]).max())
        assert np.allclose(two[0]["costs"][:2], one["costs"][:2], rtol=1e-10, atol=0)
        assert one["costs"][1] < one["costs"][0]
        outs[tag] = two
    # where the wait happens does not change a bit
    for r in range(2):
        for key in ("poses", "points", "lin", "costs"):
            np.testing.assert_array_equal(outs["gather"][r][key], outs["fused"][r][key])


def test_peer_transport_single_rank_is_the_plain_iteration(gpu):
    """One rank over the peer transport (its own receive buffer, fused wait): the iteration equals the plain one bit for bit --
    the single-GPU run of the exact launch sequence N ranks on N GPUs issue."""
    import torch
    from ba_util import make_scene
    cc = gpu.sharding.init_peer_comm(0, 1, 0)
    try:
        assert cc.peer_state() == 2 and gpu._lib.lib().mqs_comm_world_size(cc.ctx.handle) == 1
        sc = make_scene(3000, 3, seed=4, masked_frac=0.2)
        t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt)
        mk = lambda pg: gpu.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(sc["points"]),
                                                             t(sc["obs"]), t(sc["mask"], torch.uint8), t(sc["prior_w"]),
                                                             t(sc["prior_xyz"]), None, pg)
        a, b = mk(cc), mk(None)
        a.gauss_newton_iterations(3)
        b.gauss_newton_iterations(3)
        torch.cuda.synchronize()
        assert torch.equal(a.poses, b.poses) and torch.equal(a.points, b.points) and torch.equal(a.lin, b.lin)
        assert not cc.peer_timed_out()
    finally:
        cc.close()


PEER_TIMEOUT_WORKER = """
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import mqslam_amd
from ba_util import make_scene
sh = mqslam_amd.sharding
rank, local_rank, world = sh.init_from_env()
dev = torch.device("cuda", 0)
cc = sh.init_peer_comm(rank, world, 0)
sc = make_scene(4001, 4, seed=11)
pts, obs, mask, pw, px = sh.shard_arrays(rank, world, sc["points"], sc["obs"], sc["mask"], sc["prior_w"], sc["prior_xyz"])
t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
ba = mqslam_amd.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(pts), t(obs), None,
                                                 t(pw), t(px), None, cc)
ba.gauss_newton_iterations(2)              # both ranks: healthy (the first reduction of a problem waits in a kernel of its own)
torch.distributed.barrier()
if rank == 0:
    poison = 4321.5
    ba.poses_new.fill_(poison); ba.points_new.fill_(poison)
    t0 = time.time()
    try:
        ba.gauss_newton_iterations(1)      # rank 1 never issues this one: its row does not arrive
        print("no-error")
    except RuntimeError as e:
        assert "did not arrive" in str(e), str(e)
        assert 1.5 < time.time() - t0 < 20.0, time.time() - t0
        # the rows were never read: nothing was published on top of the poison (wait inside the tail), or NaN was (the solve
        # enqueued behind the stand-alone wait kernel ran on a system that kernel had set to NaN)
        for tns in (ba.poses, ba.points):
            assert bool((tns == poison).all()) or bool(torch.isnan(tns).all()), tns
        assert cc.peer_timed_out()
        try:
            ba.gauss_newton_iteration()
            print("not-sticky")
        except RuntimeError:
            print("timeout-ok")
else:
    time.sleep(6.0)
    print("timeout-ok")
torch.distributed.barrier()
cc.close()
torch.distributed.destroy_process_group()
"""


def test_peer_row_that_never_arrives_is_an_error(gpu, tmp_path):
    """A rank that does not issue its iteration: the waiting rank's bounded wait (2 s) gives up, and that is a RuntimeError from
    `gauss_newton_iterations` -- not poses computed from a half-empty receive buffer -- with the stand-alone wait kernel and with
    the wait inside the fused tail."""
    for tag, fused in (("gather", "0"), ("fused", "1")):
        script = tmp_path / ("peer_timeout_%s.py" % tag)
        script.write_text(PEER_TIMEOUT_WORKER.format(root=ROOT))
        env = dict(os.environ, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", MQS_PEER_FUSED=fused)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(29600 + os.getpid() % 250), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count("timeout-ok") == 2, r.stdout[-2000:] + r.stderr[-4000:]


PEER_RUN_WORKER = """
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import mqslam_amd
from ba_util import make_scene
sh = mqslam_amd.sharding
rank, local_rank, world = sh.init_from_env()
dev = torch.device("cuda", 0)
cc = sh.init_peer_comm(rank, world, 0)
sc = make_scene(26_000, 4, seed=13, distortion=True, masked_frac=0.1)
pts, obs, mask, pw, px = sh.shard_arrays(rank, world, sc["points"], sc["obs"], sc["mask"], sc["prior_w"], sc["prior_xyz"])
t = lambda a, dt=torch.float64: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
C = 4
pose_prior = (t(sc["poses_true"]), t(np.tile([0.02, 0.02, 0.02, 0.1, 0.1, 0.1], (C, 1))), t(np.array([1, 0, 0, 0], dtype=np.uint8), torch.uint8))
ba = mqslam_amd.bundle_adjustment.BundleAdjuster(t(sc["poses"]), t(sc["calib"]), t(sc["sigma"]), t(pts), t(obs), t(mask, torch.uint8),
                                                 t(pw), t(px), pose_prior, cc)
c0 = ba.total_cost()
ba.gauss_newton_iterations(5)          # first reduction: one-workgroup wait; then lineariser, 3 x ba_iterate_kernel, tail -- all over the peer transport
c1 = ba.total_cost()
torch.cuda.synchronize()
assert not cc.peer_timed_out()
np.savez(os.path.join({out!r}, "peerrun_r%d_of_%d.npz" % (rank, world)), poses=ba.poses.cpu().numpy(), points=ba.points.cpu().numpy(),
         costs=np.array([c0, c1]), lin=ba.lin.cpu().numpy())
if world > 1:
    torch.distributed.barrier()
cc.close()
if world > 1:
    torch.distributed.destroy_process_group()
print("peer-run-ok")
"""


def test_run_of_iterations_over_the_peer_transport_two_processes(gpu, tmp_path):
    """mqs_ba_gn_iterations_dev with its one-launch iterations (ba_iterate_kernel) between two PROCESSES that share the test box's
    GPU, the wait for the peers' finalizer pieces inside the kernels (MQS_PEER_FUSED=1: 2 x 51 workgroups, one per CU -- all
    resident): the finalizer pieces of every launch go to both receive buffers, both ranks fold them in rank order.  Identical
    bits on both ranks; the one-process solve of the whole scene to 1e-9; the two-launch iteration (MQS_BA_ITERATE=0) gives the
    same bits as the one-launch form."""
    script = tmp_path / "peer_run.py"
    script.write_text(PEER_RUN_WORKER.format(root=ROOT, out=str(tmp_path)))
    res = {}
    for tag, extra in (("one_launch", {}), ("two_launch", {"MQS_BA_ITERATE": "0"})):
        env = dict(os.environ, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1", MASTER_ADDR="127.0.0.1", MQS_PEER_FUSED="1", **extra)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29900 + os.getpid() % 90), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count("peer-run-ok") == 2, r.stdout[-2000:] + r.stderr[-4000:]
        res[tag] = [dict(np.load(tmp_path / ("peerrun_r%d_of_2.npz" % k))) for k in range(2)]
    e1 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script)], env=dict(e1, MQS_DIST_BACKEND="gloo", MQS_SHARED_GPU="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "peer-run-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    one = np.load(tmp_path / "peerrun_r0_of_1.npz")
    two = res["one_launch"]
    np.testing.assert_array_equal(two[0]["poses"], two[1]["poses"])
    np.testing.assert_array_equal(two[0]["lin"], two[1]["lin"])
    assert np.abs(two[0]["poses"] - one["poses"]).max() <= 1e-9
    pts = np.concatenate([two[0]["points"], two[1]["points"]])
    assert np.abs(pts - one["points"]).max() <= 1e-9 * max(1.0, np.abs(one["points"]).max())
    assert two[0]["costs"][1] < two[0]["costs"][0]
    for k in range(2):
        for key in ("poses", "points", "lin"):
            np.testing.assert_array_equal(res["one_launch"][k][key], res["two_launch"][k][key])
