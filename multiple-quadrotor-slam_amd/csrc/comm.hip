// Multi-GPU transport of the bundle-adjustment path: ONE sum all-reduce of the reduced camera system
// ((6C)^2 + 6C + 2 doubles, 4.8 KB at C = 4) per Gauss-Newton iteration, issued from C on the caller's stream
// ("a comm in mqs_ctx", 8(e)).  One process per GPU, one communicator per mqs_ctx.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: the library must load on machines without RCCL, and
// inside a PyTorch process it must use the ONE librccl.so.1 that process already holds (torch ships its own copy with the
// same SONAME; RTLD_NOLOAD finds it first).  The unique id is produced on rank 0 and handed to the other ranks by the
// host program over whatever it has (torch.distributed's store, MPI, a file): the boundary moves 128 opaque bytes.
#include "mqs_common.h"
#include <dlfcn.h>

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
            if (!(g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.error_string))
                g_rccl.all_reduce = nullptr;
        }
    }
    if (!g_rccl.all_reduce) {
        const char *why = dlerror();
        mqs_set_error("RCCL is not available (librccl.so.1 could not be loaded: %s)", why ? why : "symbols missing");
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

}  // namespace

void mqs_comm_release(mqs_ctx *ctx)
{
    if (ctx && ctx->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(static_cast<Comm>(ctx->comm));
    if (ctx) { ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 1; }
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

int mqs_comm_world_size(const mqs_ctx *ctx) { return (ctx && ctx->comm) ? ctx->comm_world : 0; }

int mqs_comm_destroy(mqs_ctx *ctx)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    mqs_comm_release(ctx);
    return MQS_OK;
}

int mqs_comm_all_reduce_sum_f64_dev(mqs_ctx *ctx, double *buf, int64_t n, void *stream)
{
    MQS_ARG_CHECK(ctx != nullptr && ctx->comm != nullptr, "the context has no communicator (mqs_comm_init_rank)");
    MQS_ARG_CHECK(buf != nullptr && n >= 0, "buf must not be null");
    if (n == 0) return MQS_OK;
    MQS_RCCL_CHECK(g_rccl.all_reduce(buf, buf, (size_t)n, kNcclFloat64, kNcclSum, static_cast<Comm>(ctx->comm),
                                     static_cast<hipStream_t>(stream)));
    return MQS_OK;
}

}  // extern "C"
