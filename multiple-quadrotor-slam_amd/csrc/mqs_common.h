// Internal helpers shared by the HIP translation units of libmqslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mqslam.h"

void mqs_set_error(const char *fmt, ...);

#define MQS_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            mqs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MQS_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define MQS_ARG_CHECK(cond, msg)                                                              \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            mqs_set_error("bad argument: %s (%s)", msg, #cond);                               \
            return MQS_E_ARG;                                                                 \
        }                                                                                     \
    } while (0)

struct mqs_ctx {
    int device;
    hipStream_t stream;
    void *dbuf;        // grow-only device scratch for the host-pointer entry points
    size_t dbuf_bytes;
};

// Ensures ctx->dbuf holds at least `bytes`; returns MQS_OK or an error code.
int mqs_ctx_reserve(mqs_ctx *ctx, size_t bytes);

// Wave-private LDS hand-off (one lane writes, other lanes of the SAME wavefront read later): the LDS pipe executes a
// wavefront's accesses in order, so all that is needed is that the compiler keeps the order (memory clobber) and the
// writes have been issued (lgkmcnt).  A C++ fence at wavefront scope also waits for every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)) -- which puts the latency of prefetches issued just before it on the serial chain.
__device__ __forceinline__ void mqs_wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

static inline bool mqs_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Grid for streaming one-thread-per-item kernels: one workgroup per `block` items (the hardware
// dispatcher load-balances them over the 256 CUs; a capped grid with a grid-stride loop left a
// 2.67-round tail at 1e6 items); beyond 2^20 workgroups the kernels grid-stride.
static inline unsigned mqs_stream_grid(int64_t n, int block)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 1048576) g = 1048576;
    return (unsigned)g;
}
