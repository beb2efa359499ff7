"""
Multi-GPU decomposition of the hot path: one process per GPU, landmarks sharded
in contiguous ranges, camera poses / calibrations replicated.

  triangulation, matching   no collective at all (every landmark / query row is independent);
  bundle adjustment         ONE sum all-reduce per Gauss-Newton iteration of the reduced camera
                            system [S | g | cost | count] = (6C)^2 + 6C + 2 doubles (4.8 KB at C = 4:
                            latency-bound on xGMI, not bandwidth-bound); every rank then solves the
                            same small system, so no broadcast follows.

torch.distributed is used as the transport: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests of this logic.
"""
import os


def landmark_shard(N, rank, world):
    """Contiguous range [start, stop) of the N landmarks owned by `rank` out of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d out of range for world size %d" % (rank, world))
    return (rank * N) // world, ((rank + 1) * N) // world


def unit_shard(n_units, rank, world):
    """Indices of the independent work units (matcher camera pairs, sequences) dealt round-robin to `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d out of range for world size %d" % (rank, world))
    return list(range(rank, n_units, world))


def shard_arrays(rank, world, points, obs, mask=None, prior_w=None, prior_xyz=None):
    """Slices the per-landmark arrays (numpy or torch; landmark axis = 0 for points/priors, 1 for
    obs/mask) for this rank.  Returns copies that start at a fresh (aligned) allocation."""
    a, b = landmark_shard(points.shape[0], rank, world)

    def cp(x):
        return x.clone() if hasattr(x, "clone") else x.copy()

    return (cp(points[a:b]), cp(obs[:, a:b]), None if mask is None else cp(mask[:, a:b]),
            None if prior_w is None else cp(prior_w[a:b]), None if prior_xyz is None else cp(prior_xyz[a:b]))


def all_reduce_sum_(tensor, group=None, async_op=False):
    """In-place sum over ranks of the reduced camera system (no-op without an initialised group).
    Returns the tensor, or with async_op=True the work handle (None when there is nothing to wait for)."""
    import torch.distributed as dist
    work = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else tensor


def init_from_env(backend="nccl"):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and initialises the
    process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: MQS_DIST_BACKEND=gloo runs the multi-rank code path without RCCL, and
    # MQS_SHARED_GPU=1 puts every rank on device 0 (a 1-GPU box cannot host two RCCL ranks)
    backend = os.environ.get("MQS_DIST_BACKEND", backend)
    if os.environ.get("MQS_SHARED_GPU", "0") == "1":
        local_rank = 0
    # MQS_FORCE_DIST=1: initialise the group even for ONE rank, so that a 1-GPU box can run the N-GPU code path of bench.py
    # (transport probe, C-ABI communicator, ba_strong) on the real backend
    if (world > 1 or os.environ.get("MQS_FORCE_DIST", "0") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


class CComm:
    """The library-level transport: an RCCL communicator held by an `mqs_ctx` (csrc/comm.hip), so that the one collective
    of the hot path -- the sum of the reduced camera system -- is issued from C between the kernels of an iteration
    (`mqs_ba_gn_iteration_dev`) instead of from the interpreter.  One per process (one process per GPU)."""

    def __init__(self, ctx, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world

    def all_reduce_sum_(self, tensor):
        """In-place sum over the ranks of a float64 device tensor, asynchronous on the current stream."""
        import ctypes
        import torch
        from . import _lib
        if tensor.dtype != torch.float64 or not tensor.is_cuda or not tensor.is_contiguous():
            raise ValueError("all_reduce_sum_ takes a contiguous float64 device tensor")
        _lib.check(_lib.lib().mqs_comm_all_reduce_sum_f64_dev(
            self.ctx.handle, ctypes.c_void_p(tensor.data_ptr()), tensor.numel(),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return tensor

    def close(self):
        from . import _lib
        if self.ctx is not None:
            _lib.check(_lib.lib().mqs_comm_destroy(self.ctx.handle))
            self.ctx.close()
            self.ctx = None


def init_c_comm(rank, world, device_index):
    """Creates this rank's library context on `device_index` and joins the RCCL communicator of `world` ranks.  The 128-byte
    unique id is made on rank 0 and carried to the other ranks by the torch.distributed group that `init_from_env`
    initialised (any host-side channel would do: the C ABI only sees the bytes).  Collective: every rank must call it.
    Before the (blocking, collective) communicator set-up every rank checks locally that RCCL can be bound and the ranks
    agree on that over the torch group, so that a rank without it makes ALL ranks raise instead of leaving the others
    waiting inside the collective."""
    import ctypes
    import torch
    import torch.distributed as dist
    from . import _lib
    L = _lib.lib()
    ident = (ctypes.c_uint8 * 128)()
    rc = L.mqs_comm_unique_id(ident)                  # binds RCCL (dlopen) and makes an id: only rank 0's is used
    if dist.is_available() and dist.is_initialized():
        on_gpu = dist.get_backend() == "nccl"
        dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            raise RuntimeError("RCCL is not available on every rank (%s)" % (L.mqs_last_error().decode() if rc else "another rank",))
        t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        ident = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
    elif world > 1:
        raise RuntimeError("init_c_comm needs an initialised torch.distributed group to carry the communicator id")
    else:
        _lib.check(rc)
    ctx = _lib.Context(device_index)
    _lib.check(L.mqs_comm_init_rank(ctx.handle, ident, rank, world))
    return CComm(ctx, rank, world)
