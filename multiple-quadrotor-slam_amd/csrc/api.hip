// Context, error reporting and device discovery of libmqslam_hip.so.
#include "mqs_common.h"
#include <string.h>
#include <new>

#ifndef MQS_ZERO_COPY
#define MQS_ZERO_COPY 1            // 0: always stage through the device scratch (A/B builds)
#endif

namespace {
thread_local char g_err[512] = "";
}

void mqs_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mqs_ctx_reserve(mqs_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->dbuf_bytes) return MQS_OK;
    if (ctx->dbuf) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->dbuf);
        ctx->dbuf = nullptr;
        ctx->dbuf_bytes = 0;
    }
    size_t want = bytes + bytes / 4;      // grow-only with slack, so repeated calls rarely reallocate
    hipError_t e = hipMalloc(&ctx->dbuf, want);
    if (e != hipSuccess) {
        mqs_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        ctx->dbuf = nullptr;
        return MQS_E_NOMEM;
    }
    ctx->dbuf_bytes = want;
    return MQS_OK;
}

int mqs_stage_begin(mqs_ctx *ctx, size_t bytes, mqs_stage *st)
{
    st->ctx = ctx;
    st->n_out = 0;
    st->zero_copy = false;
    if (bytes <= kZeroCopyMax && MQS_ZERO_COPY) {
        if (!ctx->hbuf) {
            // coherent (fine-grained) pinned memory: kernel writes are visible to the host once the stream has drained
            if (hipHostMalloc(&ctx->hbuf, kZeroCopyMax, hipHostMallocDefault) == hipSuccess) ctx->hbuf_bytes = kZeroCopyMax;
            else { ctx->hbuf = nullptr; (void)hipGetLastError(); }
        }
        if (ctx->hbuf) {
            st->base = static_cast<char *>(ctx->hbuf);
            st->zero_copy = true;
            return MQS_OK;
        }
    }
    const int rc = mqs_ctx_reserve(ctx, bytes);
    st->base = static_cast<char *>(ctx->dbuf);
    return rc;
}

extern "C" {

const char *mqs_last_error(void) { return g_err; }

const char *mqs_version(void) { return "mqslam-hip 0.1 (gfx950)"; }

int mqs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mqs_create(int device_id, mqs_ctx **out)
{
    MQS_ARG_CHECK(out != nullptr, "out must not be null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mqs_set_error("no HIP device visible");
        return MQS_E_NODEVICE;
    }
    MQS_ARG_CHECK(device_id >= 0 && device_id < n, "device_id out of range");
    MQS_HIP_CHECK(hipSetDevice(device_id));
    mqs_ctx *ctx = new (std::nothrow) mqs_ctx();
    if (!ctx) {
        mqs_set_error("out of host memory");
        return MQS_E_NOMEM;
    }
    ctx->device = device_id;
    ctx->dbuf = nullptr;
    ctx->dbuf_bytes = 0;
    ctx->hbuf = nullptr;
    ctx->hbuf_bytes = 0;
    ctx->comm = nullptr;
    ctx->peer = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        mqs_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return MQS_E_HIP;
    }
    *out = ctx;
    return MQS_OK;
}

void mqs_destroy(mqs_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    mqs_comm_release(ctx);
    if (ctx->dbuf) (void)hipFree(ctx->dbuf);
    if (ctx->hbuf) (void)hipHostFree(ctx->hbuf);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mqs_synchronize(mqs_ctx *ctx)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

}  // extern "C"
