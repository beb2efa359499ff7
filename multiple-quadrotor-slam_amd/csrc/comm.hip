// Multi-GPU transport of the bundle-adjustment path: ONE sum all-reduce of the reduced camera system
// ((6C)^2 + 6C + 2 doubles, 4.8 KB at C = 4) per Gauss-Newton iteration, issued from C on the caller's stream
// ("a comm in mqs_ctx", 8(e)).  One process per GPU, one communicator per mqs_ctx.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: the library must load on machines without RCCL, and
// inside a PyTorch process it must use the ONE librccl.so.1 that process already holds (torch ships its own copy with the
// same SONAME; RTLD_NOLOAD finds it first).  The unique id is produced on rank 0 and handed to the other ranks by the
// host program over whatever it has (torch.distributed's store, MPI, a file): the boundary moves 128 opaque bytes.
#include "mqs_common.h"
#include "peer_dev.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <new>

namespace {

// the part of rccl.h this file needs (rccl/rccl.h:40-43, 187, 220, 260, 339, 448, 467, 611)
struct UniqueId { char internal[128]; };
typedef void *Comm;
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(Comm *, int, UniqueId, int);
typedef int (*fn_comm_destroy)(Comm);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int /*datatype*/, int /*op*/, Comm, hipStream_t);
typedef const char *(*fn_error_string)(int);
constexpr int kNcclSum = 0, kNcclFloat64 = 8;

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false;
    char why[256] = "";                  // the loader's message of the first failed attempt (dlerror() is consumed by reading it)
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.all_reduce) return MQS_OK;
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names) {          // a copy the process already holds wins (torch's)
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        for (int k = 0; !h && k < 3; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            g_rccl.handle = h;
            g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
            g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
            g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
            g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
            g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
            if (!(g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.error_string && g_rccl.all_reduce)) {
                g_rccl.all_reduce = nullptr;
                snprintf(g_rccl.why, sizeof(g_rccl.why), "a symbol of the RCCL API is missing in the loaded library");
            }
        } else {
            const char *why = dlerror();
            snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", why ? why : "dlopen failed");
        }
    }
    if (!g_rccl.all_reduce) {
        mqs_set_error("RCCL is not available (librccl.so.1: %s)", g_rccl.why);
        return MQS_E_RCCL;
    }
    return MQS_OK;
}

#define MQS_RCCL_CHECK(expr)                                                                             \
    do {                                                                                                 \
        int _r = (expr);                                                                                 \
        if (_r != 0) {                                                                                   \
            mqs_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.error_string(_r), __FILE__, __LINE__);  \
            return MQS_E_RCCL;                                                                           \
        }                                                                                                \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Peer transport: the one collective of the path as plain stores over xGMI.
//
// The message is 4.8 KB once per iteration: a generic RCCL all-reduce is a launch of its own (plus the library's protocol
// set-up) between two kernels that take 20 us each on a 125 k-landmark shard.  Here every rank owns a small RECEIVE buffer in
// its own HBM (fine-grained, so that what a peer stores becomes visible to a system-scope acquire), mapped into every other
// rank's address space through hipIpc handles -- the host program carries the handles like it carries RCCL's unique id.  A
// reduction = every rank stores its row into slot [rank] of every peer's buffer and then, behind a system-scope release,
// a sequence number into that slot's flags; the consumer spins (bounded) until all slots carry the sequence number and adds
// the rows IN RANK ORDER -- the same bits on every rank, no reduction tree, no extra launch: the stores ride in the finalize
// kernel of the lineariser, the wait and the sum in the prologue of the fused tail (ba.hip).  Two parities alternate so that
// a fast rank's next row never lands on one a slow rank is still reading (a rank cannot run two reductions ahead: the
// one in between needs the slow rank's row).  mqs_comm_all_reduce_sum_f64_dev runs the same protocol in one small kernel
// for any buffer of up to kPeerRowDoubles doubles.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPeerMaxWorld = MQS_PEER_MAX_WORLD;
constexpr int kPeerRowDoubles = MQS_FIN_PIECES * MQS_PEER_QUARTER_STRIDE;   // a rank's slot: 5 120 doubles (the fused finalize's piece rows; any row <= that)
constexpr int kPeerFlagsPerRank = 64;                   // one per piece of a rank's row: 6 finalize workgroups, or the fused tail's 48 finalizer pieces

struct PeerComm {
    int rank = 0, world = 1;
    bool shared_device = false;          // two ranks live on one GPU (tests): never spin inside a chip-filling kernel
    char *mine = nullptr;                // this rank's receive buffer
    char *peer[kPeerMaxWorld] = {};      // rank q's receive buffer as mapped here (peer[rank] == mine)
    bool opened[kPeerMaxWorld] = {};
    unsigned long long seq = 0;          // sequence number of the last reduction issued
    bool open = false;
    char bus_id[32] = {};
};

size_t peer_rows_off(int world, int parity) { return (size_t)parity * world * kPeerRowDoubles * 8; }
size_t peer_flags_off(int world, int parity) { return (size_t)2 * world * kPeerRowDoubles * 8 + (size_t)parity * world * kPeerFlagsPerRank * 8; }
size_t peer_timeout_off(int world) { return peer_flags_off(world, 2); }
size_t peer_bytes(int world) { return peer_timeout_off(world) + 256; }

// the whole reduction in one workgroup: push `buf` to every rank, wait for every rank's row, sum in rank order into `buf`
__global__ __launch_bounds__(256) void peer_all_reduce_kernel(double *__restrict__ buf, int n, mqs_peer_push push, mqs_peer_recv recv)
{
    __shared__ int timed_out;
    const int tid = threadIdx.x;
    if (tid == 0) timed_out = 0;
    for (int i = tid; i < n; i += 256) mqs::peer::push_entry(push, i, buf[i]);
    mqs_stores_landed();
    __syncthreads();                     // the row has landed everywhere before its flags go up (peer_dev.h)
    if (tid < push.world * recv.flags_per_rank)
        __hip_atomic_store(push.flag[tid / recv.flags_per_rank] + tid % recv.flags_per_rank, push.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    mqs::peer::wait_and_sum(buf, n, recv, tid, 256, &timed_out);      // a wait that gave up leaves NaN in `buf` (wait_and_sum): nothing that looks like a sum
}

// the wait + sum alone (the rows were pushed by the lineariser's finalize kernel): a launch of its own for ranks that share a GPU
__global__ __launch_bounds__(256) void peer_gather_kernel(double *__restrict__ out, int n, mqs_peer_recv recv)
{
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    mqs::peer::wait_and_sum(out, n, recv, threadIdx.x, 256, &timed_out);
}

PeerComm *peer_of(const mqs_ctx *ctx) { return ctx ? static_cast<PeerComm *>(ctx->peer) : nullptr; }

void peer_release(mqs_ctx *ctx)
{
    PeerComm *pc = peer_of(ctx);
    if (!pc) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int q = 0; q < pc->world; ++q)
        if (pc->opened[q] && pc->peer[q]) (void)hipIpcCloseMemHandle(pc->peer[q]);
    if (pc->mine) (void)hipFree(pc->mine);
    delete pc;
    ctx->peer = nullptr;
}

}  // namespace

void mqs_comm_release(mqs_ctx *ctx)
{
    if (ctx && ctx->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(static_cast<Comm>(ctx->comm));
    if (ctx) peer_release(ctx);
    if (ctx) { ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 1; }
}

// ba_iter.hip: the next reduction over the peer transport.  Fills where this rank's row goes (push) and where all rows
// arrive (recv); returns 0 without an open peer transport or when the row does not fit.  *fused_wait: the consumer kernel
// may wait for the rows itself (no peer shares this rank's GPU; MQS_PEER_FUSED=0/1 overrides).
int mqs_comm_peer_next(mqs_ctx *ctx, int64_t n, int flags_used, mqs_peer_push *push, mqs_peer_recv *recv, int *fused_wait)
{
    PeerComm *pc = peer_of(ctx);
    if (!pc || !pc->open || n > kPeerRowDoubles || flags_used < 1 || flags_used > kPeerFlagsPerRank) return 0;
    const unsigned long long seq = ++pc->seq;
    const int parity = (int)(seq & 1);
    push->world = pc->world;
    push->seq = seq;
    for (int q = 0; q < pc->world; ++q) {
        push->dst[q] = reinterpret_cast<double *>(pc->peer[q] + peer_rows_off(pc->world, parity)) + (size_t)pc->rank * kPeerRowDoubles;
        push->flag[q] = reinterpret_cast<unsigned long long *>(pc->peer[q] + peer_flags_off(pc->world, parity)) +
                        (size_t)pc->rank * kPeerFlagsPerRank;
    }
    recv->rows = reinterpret_cast<const double *>(pc->mine + peer_rows_off(pc->world, parity));
    recv->flags = reinterpret_cast<const unsigned long long *>(pc->mine + peer_flags_off(pc->world, parity));
    recv->world = pc->world;
    recv->row_stride = kPeerRowDoubles;
    recv->flags_per_rank = flags_used;
    recv->flags_stride = kPeerFlagsPerRank;
    recv->seq = seq;
    recv->timeout_flag = reinterpret_cast<int *>(pc->mine + peer_timeout_off(pc->world));
    recv->status = nullptr;
    recv->spin_ticks = mqs::peer::kSpinTicks;
    if (fused_wait) {
        const char *e = getenv("MQS_PEER_FUSED");
        *fused_wait = (e && *e) ? (e[0] != '0') : !pc->shared_device;
    }
    return 1;
}

bool mqs_comm_peer_fused(const mqs_ctx *ctx)
{
    const PeerComm *pc = peer_of(ctx);
    if (!pc || !pc->open) return false;
    const char *e = getenv("MQS_PEER_FUSED");
    return (e && *e) ? (e[0] != '0') : !pc->shared_device;
}

int mqs_comm_peer_gather(const mqs_peer_recv *recv, double *out, int64_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(peer_gather_kernel, dim3(1), dim3(256), 0, stream, out, (int)n, *recv);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

extern "C" {

int mqs_comm_unique_id(uint8_t *id128)
{
    MQS_ARG_CHECK(id128 != nullptr, "id128 must not be null");
    int rc = rccl_load();
    if (rc != MQS_OK) return rc;
    UniqueId id;
    MQS_RCCL_CHECK(g_rccl.get_unique_id(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return MQS_OK;
}

int mqs_comm_init_rank(mqs_ctx *ctx, const uint8_t *id128, int rank, int world)
{
    MQS_ARG_CHECK(ctx != nullptr && id128 != nullptr, "ctx and id128 must not be null");
    MQS_ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "0 <= rank < world");
    MQS_ARG_CHECK(ctx->comm == nullptr, "this context already has a communicator");
    int rc = rccl_load();
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm comm = nullptr;
    MQS_RCCL_CHECK(g_rccl.comm_init_rank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return MQS_OK;
}

int mqs_comm_world_size(const mqs_ctx *ctx)
{
    if (ctx && ctx->comm) return ctx->comm_world;
    const PeerComm *pc = peer_of(ctx);
    return (pc && pc->open) ? pc->world : 0;
}

// ---- peer transport set-up: export -> (host program gathers the handles of all ranks) -> open ----
int mqs_comm_peer_export(mqs_ctx *ctx, int rank, int world, uint8_t *handle)
{
    MQS_ARG_CHECK(ctx != nullptr && handle != nullptr, "ctx and handle must not be null");
    MQS_ARG_CHECK(world >= 1 && world <= kPeerMaxWorld && rank >= 0 && rank < world, "0 <= rank < world <= MQS_PEER_MAX_WORLD");
    MQS_ARG_CHECK(ctx->peer == nullptr, "this context already has a peer transport");
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "the handle blob reserves 64 bytes for the IPC handle");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    PeerComm *pc = new (std::nothrow) PeerComm();
    if (!pc) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
    pc->rank = rank; pc->world = world;
    // fine-grained device memory: a peer's stores over the fabric are visible to this GPU's system-scope acquire loads
    // (ordinary hipMalloc memory may sit in this GPU's L2 as a stale line).  Plain hipMalloc as the fallback where the
    // extended allocator refuses: still correct between processes that share ONE GPU (one L2).
    void *buf = nullptr;
    if (hipExtMallocWithFlags(&buf, peer_bytes(world), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&buf, peer_bytes(world), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            hipError_t e = hipMalloc(&buf, peer_bytes(world));
            if (e != hipSuccess) { delete pc; mqs_set_error("peer receive buffer: %s", hipGetErrorString(e)); return MQS_E_NOMEM; }
        }
    }
    pc->mine = static_cast<char *>(buf);
    hipError_t e = hipMemset(buf, 0, peer_bytes(world));
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, buf);
    if (e != hipSuccess) {
        (void)hipFree(buf);
        delete pc;
        mqs_set_error("peer receive buffer could not be exported (hipIpcGetMemHandle: %s)", hipGetErrorString(e));
        return MQS_E_HIP;
    }
    memset(handle, 0, MQS_PEER_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    (void)hipDeviceGetPCIBusId(pc->bus_id, (int)sizeof(pc->bus_id), ctx->device);
    memcpy(handle + 64, pc->bus_id, 32);
    ctx->peer = pc;
    return MQS_OK;
}

int mqs_comm_peer_open(mqs_ctx *ctx, const uint8_t *handles)
{
    PeerComm *pc = peer_of(ctx);
    MQS_ARG_CHECK(pc != nullptr && handles != nullptr, "mqs_comm_peer_export first; handles must not be null");
    MQS_ARG_CHECK(!pc->open, "the peer transport is already open");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    // do ANY two ranks share a GPU?  Decided from all handles, so that every rank reaches the same answer (the answer selects the
    // protocol of an iteration: ranks must not disagree on it)
    for (int a = 0; a < pc->world; ++a)
        for (int b = a + 1; b < pc->world; ++b)
            if (memcmp(handles + (size_t)a * MQS_PEER_HANDLE_BYTES + 64, handles + (size_t)b * MQS_PEER_HANDLE_BYTES + 64, 32) == 0)
                pc->shared_device = true;
    for (int q = 0; q < pc->world; ++q) {
        const uint8_t *hq = handles + (size_t)q * MQS_PEER_HANDLE_BYTES;
        if (q == pc->rank) { pc->peer[q] = pc->mine; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, hq, sizeof(h));
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            mqs_set_error("rank %d's receive buffer could not be mapped (hipIpcOpenMemHandle: %s)", q, hipGetErrorString(e));
            return MQS_E_HIP;
        }
        pc->peer[q] = static_cast<char *>(p);
        pc->opened[q] = true;
    }
    pc->open = true;
    return MQS_OK;
}

int mqs_comm_peer_close(mqs_ctx *ctx)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    peer_release(ctx);
    return MQS_OK;
}

// 0: no peer transport; 1: open, consumers wait in a kernel of their own (a peer shares this GPU); 2: open, fused wait
int mqs_comm_peer_state(const mqs_ctx *ctx)
{
    const PeerComm *pc = peer_of(ctx);
    if (!pc || !pc->open) return 0;
    return pc->shared_device ? 1 : 2;
}

// device int of the context's receive buffer that a consumer sets when a row did not arrive within the spin bound (2 s);
// copies it to *timed_out (synchronises `stream`)
int mqs_comm_peer_timed_out(mqs_ctx *ctx, void *stream, int *timed_out)
{
    PeerComm *pc = peer_of(ctx);
    MQS_ARG_CHECK(pc != nullptr && timed_out != nullptr, "no peer transport");
    MQS_HIP_CHECK(hipMemcpyAsync(timed_out, pc->mine + peer_timeout_off(pc->world), sizeof(int), hipMemcpyDeviceToHost,
                                 static_cast<hipStream_t>(stream)));
    MQS_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return MQS_OK;
}

int mqs_comm_destroy(mqs_ctx *ctx)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    mqs_comm_release(ctx);
    return MQS_OK;
}

int mqs_comm_all_reduce_sum_f64_dev(mqs_ctx *ctx, double *buf, int64_t n, void *stream)
{
    MQS_ARG_CHECK(ctx != nullptr && (ctx->comm != nullptr || peer_of(ctx) != nullptr), "the context has no communicator (mqs_comm_init_rank / mqs_comm_peer_open)");
    MQS_ARG_CHECK(buf != nullptr && n >= 0, "buf must not be null");
    if (n == 0) return MQS_OK;
    {
        // the peer transport takes what fits its rows (the BA system does); anything larger goes to RCCL
        mqs_peer_push push;
        mqs_peer_recv recv;
        if (mqs_comm_peer_next(ctx, n, 1, &push, &recv, nullptr)) {
            hipLaunchKernelGGL(peer_all_reduce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), buf, (int)n, push, recv);
            MQS_HIP_CHECK(hipGetLastError());
            return MQS_OK;
        }
    }
    MQS_ARG_CHECK(ctx->comm != nullptr, "the buffer does not fit the peer transport and the context has no RCCL communicator");
    MQS_RCCL_CHECK(g_rccl.all_reduce(buf, buf, (size_t)n, kNcclFloat64, kNcclSum, static_cast<Comm>(ctx->comm),
                                     static_cast<hipStream_t>(stream)));
    return MQS_OK;
}

}  // extern "C"
