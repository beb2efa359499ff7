// Internal helpers shared by the HIP translation units of libmqslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
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
    void *hbuf;        // pinned, GPU-visible host buffer for the small-call path (see mqs_stage)
    size_t hbuf_bytes;
    void *comm;        // RCCL communicator of this rank (comm.hip), or null: one process per GPU, one ctx per process
    void *peer;        // peer transport of this rank (comm.hip: receive buffer + the peers' mapped buffers), or null
    int comm_rank, comm_world;
};

// comm.hip <-> ba.hip / ba_iter.hip: one reduction over the peer transport (see comm.hip).  push = where this rank's row goes
// in every rank's receive buffer (slot [rank] of the current parity) and the flags behind it; recv = this rank's own buffer.
struct mqs_peer_push {
    double *dst[MQS_PEER_MAX_WORLD];
    unsigned long long *flag[MQS_PEER_MAX_WORLD];
    int world;
    unsigned long long seq;
};
struct mqs_peer_recv {
    const double *rows;                  // [world][row_stride]
    const unsigned long long *flags;     // [world][flags_stride]; the first flags_per_rank of a rank are in use: >= seq when its piece landed
    int world, row_stride, flags_per_rank, flags_stride;
    unsigned long long seq;
    int *timeout_flag;                   // device int, set when a row did not arrive within the spin bound
    int *status;                         // null, or the host-visible status word of the problem this reduction belongs to (ba_iter.hip)
    long long spin_ticks;                // bound of the wait in ticks of the 100 MHz wall clock (peer_dev.h: kSpinTicks)
};
// values of a problem's status word (sticky; mqs_ba_problem_status): which bounded wait of an iteration gave up
constexpr int MQS_STATUS_FINALIZE_TIMEOUT = 1;     // the fused tail, for its own launch's finalizer workgroups
constexpr int MQS_STATUS_PEER_TIMEOUT = 2;         // a consumer of the peer transport, for another rank's row
int mqs_comm_peer_next(mqs_ctx *ctx, int64_t n, int flags_used, mqs_peer_push *push, mqs_peer_recv *recv, int *fused_wait);
int mqs_comm_peer_gather(const mqs_peer_recv *recv, double *out, int64_t n, hipStream_t stream);
// an open peer transport whose consumers may wait inside their own kernel (no two ranks share a GPU; MQS_PEER_FUSED overrides)
bool mqs_comm_peer_fused(const mqs_ctx *ctx);
// ba.hip: the lineariser with its finalize kernel storing the rank's reduced system into the peers' buffers (push != null)
int mqs_ba_linearize_push(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                          const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                          double lambda, double *out, void *workspace, int64_t workspace_bytes, hipStream_t stream,
                          const mqs_peer_push *push);
int mqs_ba_finalize_groups(int C);       // workgroups of the finalize kernel = flags a rank sets per reduction
// the finalize fused into the tail (ba.hip): where the partial rows are, where the quarter sums and their flags go
constexpr int MQS_FIN_PIECES = 8;                  // pieces the partial rows are cut into (fused finalize: one workgroup per piece and 64 slots)
constexpr int MQS_PEER_QUARTER_STRIDE = 640;       // doubles between the piece rows (>= (6 * 4)^2 + 6 * 4 + 2); MQS_FIN_PIECES of them = a rank's slot
struct mqs_ba_fin {
    const double *partials;
    int nrows;
    double *quarters;
    unsigned long long *flags;
    unsigned long long epoch;
    const mqs_peer_push *push;           // non-null: the quarters also go into every rank's receive buffer
    int *status;                         // null, or the problem's host-visible status word (set when the wait for the finalizers gives up)
    int withhold;                        // test hook (mqs_debug_ba_withhold_flag): the finalizer piece whose flag is NOT raised; -1 = none
};
int mqs_ba_linearize_for_fused_tail(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                                    const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                                    double lambda, void *workspace, int64_t workspace_bytes, hipStream_t stream, mqs_ba_fin *fin);
// a run of iterations with ONE launch per iteration (ba.hip: ba_iterate_kernel = the tail of iteration k + the lineariser of k + 1)
bool mqs_ba_iterate_eligible(int C, int64_t N);
int mqs_ba_iterate_launch(const mqs_peer_recv *peer, const mqs_ba_fin *fin, int C, const double *poses, const double *calib, const double *sigma,
                          const double *points, const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                          int64_t N, double lambda, const double *prior_poses, const double *prior_sigmas, const uint8_t *prior_mask,
                          double *lin_out, double *dpose, double *poses_out, double *info, double *points_out, void *workspace,
                          hipStream_t stream, mqs_ba_fin *fin_next);
bool mqs_ba_wave_path(int C);            // wave lineariser + fused tail serve this camera count (and are not switched off)
bool mqs_ba_fused_finalize_enabled();    // MQS_BA_FINALIZE=kernel keeps the finalize launch (A/B)
int mqs_ba_tail_launch(const double *lin, const mqs_peer_recv *peer, const mqs_ba_fin *fin, int C, const double *poses, const double *calib,
                       const double *sigma, const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                       const double *prior_xyz, int64_t N, double lambda, const double *prior_poses, const double *prior_sigmas,
                       const uint8_t *prior_mask, double *lin_out, double *dpose, double *poses_out, double *info,
                       double *points_out, hipStream_t stream);

// Launchers the device-resident frame loop (slam_frame.hip) shares with the public entry points.  `n` / `N` are CAPACITIES
// (grids and workspaces are sized for them); n_dev (device memory, may be null) holds the live count the kernels use.
int mqs_lk_launch(const uint8_t *prev_img, const uint8_t *next_img, int W, int H, const float *prev_pts, int n, const int32_t *n_dev,
                  int win_w, int win_h, int max_level, int max_iter, double eps, double min_eig_threshold, float *next_pts,
                  uint8_t *status, float *err, void *workspace, int64_t workspace_bytes, hipStream_t stream, int phases);   // phases: 1 pyramid, 2 tracker, 3 both
int mqs_pnp_ransac_launch(const double *objp, const double *imgp, int N, const int32_t *n_dev, const double *intr,
                          const int32_t *samples, int B, int sample_size, double reproj_error, int sample_iters, int max_iter,
                          double eps, double *pose_out, int32_t *sel_out, uint8_t *mask, double *info, void *workspace,
                          hipStream_t stream, int end_in_caller);
void mqs_pnp_workspace_layout(void *workspace, int B, double **poses, int32_t **counts, int32_t **inlier_idx);
int mqs_keyframe_step_launch(const double *objp, const double *imgp, int n_old, const double *p0, const double *p1, int n_new,
                             const double *intr, const double *P_prev, const double *P0, double tolerance, int max_iter, double eps,
                             double second_pass_screen_px, double *scratch, double *pose_out, double *x_out, int32_t *status_out, double *info,
                             hipStream_t stream);

// comm.hip: releases ctx->comm (called by mqs_destroy)
void mqs_comm_release(mqs_ctx *ctx);

// chol_nd.hip: banded SPD solve with the band cut into independent chunks; *done = false when it does not apply
int mqs_chol_nd_solve(double *S, double *x, int n, int hb, int *bad, hipStream_t stream, bool *done);

// Ensures ctx->dbuf holds at least `bytes`; returns MQS_OK or an error code.
int mqs_ctx_reserve(mqs_ctx *ctx, size_t bytes);

// Small-call path of the single-pass host-pointer entry points (the reference's real sizes: <= 300 points per call,
// slam2.py:1080-1082).  Below kZeroCopyMax bytes the "device" buffer IS a pinned host buffer the kernel reads and writes
// over the fabric: inputs are a CPU memcpy, outputs a CPU memcpy after the stream has drained -- no hipMemcpy calls at all
// (each costs ~8-10 us from pageable memory; five of them made a 300-point call 66 us).  Only for kernels that touch
// every byte once; iterative kernels (pose refinement) keep the device scratch.
constexpr size_t kZeroCopyMax = 256 * 1024;
struct mqs_stage {
    mqs_ctx *ctx;
    char *base;                 // where the call's layout lives (device scratch, or the pinned host buffer)
    bool zero_copy;
    int n_out;
    struct { void *user; const void *src; size_t bytes; } out[6];
};
int mqs_stage_begin(mqs_ctx *ctx, size_t bytes, mqs_stage *st);
inline hipError_t mqs_stage_in(mqs_stage *st, void *dst, const void *src, size_t bytes)
{
    if (st->zero_copy) { memcpy(dst, src, bytes); return hipSuccess; }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st->ctx->stream);
}
inline hipError_t mqs_stage_out(mqs_stage *st, void *user, const void *src, size_t bytes)
{
    if (st->zero_copy) { st->out[st->n_out].user = user; st->out[st->n_out].src = src; st->out[st->n_out].bytes = bytes; ++st->n_out; return hipSuccess; }
    return hipMemcpyAsync(user, src, bytes, hipMemcpyDeviceToHost, st->ctx->stream);
}
inline hipError_t mqs_stage_end(mqs_stage *st)
{
    const hipError_t e = hipStreamSynchronize(st->ctx->stream);
    if (e == hipSuccess && st->zero_copy)
        for (int k = 0; k < st->n_out; ++k) memcpy(st->out[k].user, st->out[k].src, st->out[k].bytes);
    return e;
}

// Wave-private LDS hand-off (one lane writes, other lanes of the SAME wavefront read later): the LDS pipe executes a
// wavefront's accesses in order, so all that is needed is that the compiler keeps the order (memory clobber) and the
// writes have been issued (lgkmcnt).  A C++ fence at wavefront scope also waits for every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)) -- which puts the latency of prefetches issued just before it on the serial chain.
__device__ __forceinline__ void mqs_wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Every global store this wavefront has issued so far has been acknowledged by the memory it went to.  A __syncthreads() does
// NOT bring this about on this target: outside threadgroup-split mode a workgroup-scope release needs no s_waitcnt vmcnt(0) (the
// waves of a workgroup share their CU's vector cache), and the compiler emits none -- 0 of the 146 barriers of ba.hip carry one.
// Whoever publishes data to ANOTHER workgroup, GPU or cache (the scalar cache included) with plain or relaxed-atomic stores and
// then raises a flag calls this between the two (then a barrier, if other waves' stores are covered by the same flag).
__device__ __forceinline__ void mqs_stores_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Dynamic LDS above 64 KiB needs hipFuncSetAttribute once per (kernel, device): `flags` is that kernel's per-device record.
struct mqs_lds_opt_in { bool done[64] = {}; };
static inline hipError_t mqs_lds_opt_in_once(mqs_lds_opt_in &flags, const void *kernel, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && flags.done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) flags.done[dev] = true;
    return e;
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
