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
    """The library-level transport: an RCCL communicator and / or the peer transport (plain stores into receive buffers the
    ranks map from each other) held by an `mqs_ctx` (csrc/comm.hip), so that the one collective of the hot path -- the sum of
    the reduced camera system -- is issued from C between the kernels of an iteration (`mqs_ba_gn_iteration_dev`) or, over
    the peer transport, INSIDE them, instead of from the interpreter.  One per process (one process per GPU)."""

    def __init__(self, ctx, rank, world, transport="rccl"):
        self.ctx, self.rank, self.world, self.transport = ctx, rank, world, transport

    def peer_state(self):
        """0: no peer transport, 1: open with a stand-alone wait kernel (a peer shares this GPU), 2: open, fused wait."""
        from . import _lib
        return int(_lib.lib().mqs_comm_peer_state(self.ctx.handle))

    def peer_timed_out(self):
        """True when a consumer kernel gave up waiting for a peer's row (2 s): the results since are not sums."""
        import ctypes
        import torch
        from . import _lib
        flag = ctypes.c_int(0)
        _lib.check(_lib.lib().mqs_comm_peer_timed_out(self.ctx.handle, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                      ctypes.byref(flag)))
        return flag.value != 0

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
            _lib.check(_lib.lib().mqs_comm_destroy(self.ctx.handle))      # releases the peer transport too
            self.ctx.close()
            self.ctx = None


def _agree(ok, dev):
    """True only if `ok` holds on every rank (one MIN all-reduce over the torch group; no group: the local value)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def open_peer_transport(ctx, rank, world, device_index):
    """Gives the library context `ctx` the peer transport of csrc/comm.hip: this rank's receive buffer is exported as an
    opaque blob, the blobs of all ranks travel over the torch.distributed group (any host channel would do), every rank maps
    the others' buffers.  Collective: every rank must call it.  The ranks agree on success after each step, so a rank that
    cannot export or map makes ALL ranks return False (and none is left waiting); the context is then unchanged."""
    import ctypes
    import torch
    import torch.distributed as dist
    from . import _lib
    L = _lib.lib()
    grouped = dist.is_available() and dist.is_initialized()
    if world > 1 and not grouped:
        raise RuntimeError("open_peer_transport needs an initialised torch.distributed group to carry the buffer handles")
    on_gpu = grouped and dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")
    nb = int(_lib.MQS_PEER_HANDLE_BYTES)
    blob = (ctypes.c_uint8 * nb)()
    rc = L.mqs_comm_peer_export(ctx.handle, rank, world, blob) if world <= int(_lib.MQS_PEER_MAX_WORLD) else -1
    if not _agree(rc == 0, dev):
        if rc == 0:
            L.mqs_comm_peer_close(ctx.handle)
        return False
    mine = torch.tensor(list(blob), dtype=torch.uint8, device=dev)
    if grouped:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allb = torch.cat(parts).cpu().numpy().tobytes()
    else:
        allb = bytes(blob)
    handles = (ctypes.c_uint8 * (nb * world)).from_buffer_copy(allb)
    rc = L.mqs_comm_peer_open(ctx.handle, handles)
    if not _agree(rc == 0, dev):
        L.mqs_comm_peer_close(ctx.handle)
        return False
    if grouped:
        dist.barrier()                    # every rank has mapped every buffer before anyone stores into one
    return True


def init_peer_comm(rank, world, device_index):
    """A library context with the peer transport only (no RCCL communicator): what the two-processes-on-one-GPU tests use,
    and any host program whose ranks can exchange 128 bytes each.  Raises if the transport cannot be opened on every rank."""
    from . import _lib
    ctx = _lib.Context(device_index)
    if not open_peer_transport(ctx, rank, world, device_index):
        ctx.close()
        raise RuntimeError("the peer transport could not be opened on every rank (%s)" % (_lib.lib().mqs_last_error().decode(),))
    return CComm(ctx, rank, world, transport="peer")


def init_c_comm(rank, world, device_index, peer=False):
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
    cc = CComm(ctx, rank, world)
    if peer and open_peer_transport(ctx, rank, world, device_index):
        cc.transport = "peer+rccl"        # the BA system travels as peer stores; RCCL serves what does not fit a row
    return cc
