// One Gauss-Newton / fixed-damping iteration of the dense-visibility bundle adjustment behind ONE call:
//   linearise + on-chip Schur elimination -> finalize -> [sum all-reduce of the reduced camera system over the ranks]
//   -> reduced solve + pose retraction -> landmark back-substitution,
// enqueued back to back on the caller's stream with no host round trip and no interpreter between the launches
// (the optimiser loop of bundle_adjust.cpp:323-324 with the graph of :268-298; kernels in ba.hip, transport in comm.hip).
// At 125 k landmarks per GPU (BASELINE configs[3], 1e6 landmarks sharded 8-way) the kernels of an iteration take ~50 us:
// five ctypes calls and a torch.distributed call per iteration would set the iteration time, this call does not.
#include "mqs_common.h"
#include "peer_dev.h"
#include <new>

struct mqs_ba_problem {
    mqs_ctx *ctx;           // may be null (single GPU, no collective)
    int C;
    int64_t N;
    double *poses[2], *points[2];
    const double *calib, *sigma, *obs, *prior_w, *prior_xyz, *prior_poses, *prior_sigmas;
    const uint8_t *mask, *prior_mask;
    double *lin, *dpose, *info;
    void *ws;
    int64_t ws_bytes;
    int cur;                // which of poses[] / points[] holds the current estimate
    int *status_host;       // pinned host word the kernels of an iteration set when a bounded wait gives up (sticky; 0 = healthy)
    int *status_dev;        // the same word as the device addresses it
    long long peer_reductions;   // reductions of this problem that went over the peer transport so far (the first one waits differently)
};

namespace {

const char *status_text(int s)
{
    switch (s) {
    case MQS_STATUS_FINALIZE_TIMEOUT: return "the tail of an iteration gave up waiting (2 s) for the finalizer workgroups of its own launch";
    case MQS_STATUS_PEER_TIMEOUT: return "a rank's row of the reduced camera system did not arrive within the wait's bound (peer transport: 30 s for a problem's first reduction, 2 s afterwards)";
    }
    return "unknown status";
}

// the sticky status of the problem and of its context's peer transport WITHOUT touching the stream: what has been seen so far
int status_so_far(const mqs_ba_problem *p)
{
    const int s = p->status_host ? __atomic_load_n(p->status_host, __ATOMIC_RELAXED) : 0;
    if (s == 0) return MQS_OK;
    mqs_set_error("bundle adjustment iteration failed: %s; the estimate of this problem is not valid any more", status_text(s));
    return MQS_E_TIMEOUT;
}

}  // namespace

extern "C" {

int mqs_ba_problem_create(mqs_ctx *ctx, int C, int64_t N, double *poses_a, double *poses_b, const double *calib,
                          const double *sigma, double *points_a, double *points_b, const double *obs, const uint8_t *mask,
                          const double *prior_w, const double *prior_xyz, const double *prior_poses,
                          const double *prior_sigmas, const uint8_t *prior_mask, double *lin, double *dpose, double *info,
                          void *workspace, int64_t workspace_bytes, mqs_ba_problem **out)
{
    MQS_ARG_CHECK(out != nullptr, "out must not be null");
    *out = nullptr;
    MQS_ARG_CHECK(C >= 1 && C <= MQS_MAX_CAMS, "1 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    MQS_ARG_CHECK(poses_a && poses_b && poses_a != poses_b && calib && sigma, "poses_a, poses_b (distinct), calib, sigma must not be null");
    MQS_ARG_CHECK(N == 0 || (points_a && points_b && points_a != points_b && obs), "points_a, points_b (distinct), obs must not be null");
    MQS_ARG_CHECK(lin && dpose && workspace, "lin, dpose, workspace must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_ba_workspace_bytes(C, N), "workspace too small (mqs_ba_workspace_bytes)");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    MQS_ARG_CHECK(!prior_mask || (prior_poses && prior_sigmas), "prior_poses/prior_sigmas required with prior_mask");
    MQS_ARG_CHECK(mqs_aligned16(points_a) && mqs_aligned16(points_b) && mqs_aligned16(obs), "device pointers must be 16-byte aligned");
    mqs_ba_problem *p = new (std::nothrow) mqs_ba_problem();
    if (!p) {
        mqs_set_error("out of host memory");
        return MQS_E_NOMEM;
    }
    p->ctx = ctx; p->C = C; p->N = N;
    p->poses[0] = poses_a; p->poses[1] = poses_b; p->points[0] = points_a; p->points[1] = points_b;
    p->calib = calib; p->sigma = sigma; p->obs = obs; p->mask = mask; p->prior_w = prior_w; p->prior_xyz = prior_xyz;
    p->prior_poses = prior_poses; p->prior_sigmas = prior_sigmas; p->prior_mask = prior_mask;
    p->lin = lin; p->dpose = dpose; p->info = info; p->ws = workspace; p->ws_bytes = workspace_bytes;
    p->cur = 0;
    // the status word lives in pinned host memory the GPU writes over the fabric: the host reads it without a copy, and
    // mqs_ba_gn_iteration_dev checks it on entry without synchronising anything
    void *h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        mqs_set_error("the problem's status word could not be allocated (hipHostMalloc)");
        return MQS_E_NOMEM;
    }
    memset(h, 0, 64);
    p->status_host = static_cast<int *>(h);
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        delete p;
        mqs_set_error("the problem's status word could not be mapped (hipHostGetDevicePointer)");
        return MQS_E_HIP;
    }
    p->status_dev = static_cast<int *>(d);
    *out = p;
    return MQS_OK;
}

void mqs_ba_problem_destroy(mqs_ba_problem *p)
{
    if (p && p->status_host) (void)hipHostFree(p->status_host);
    delete p;
}

// MQS_OK, or MQS_E_TIMEOUT when a bounded wait inside an iteration of this problem gave up (sticky: the estimate is not valid
// any more).  Synchronises `stream` first, so that it speaks for everything enqueued on it.
int mqs_ba_problem_status(mqs_ba_problem *p, void *stream)
{
    MQS_ARG_CHECK(p != nullptr, "problem must not be null");
    MQS_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return status_so_far(p);
}

int mqs_ba_problem_current(const mqs_ba_problem *p) { return p ? p->cur : -1; }

int mqs_ba_problem_set_current(mqs_ba_problem *p, int which)
{
    MQS_ARG_CHECK(p != nullptr && (which == 0 || which == 1), "which must be 0 or 1");
    p->cur = which;
    return MQS_OK;
}

// first half: linearise the current estimate into `lin` (this rank's partial reduced system)
int mqs_ba_gn_begin_dev(mqs_ba_problem *p, double lambda, void *stream)
{
    MQS_ARG_CHECK(p != nullptr, "problem must not be null");
    return mqs_ba_linearize_dev(p->poses[p->cur], p->calib, p->sigma, p->C, p->points[p->cur], p->obs, p->mask, p->prior_w,
                                p->prior_xyz, p->N, lambda, p->lin, p->ws, p->ws_bytes, stream);
}

// second half (after `lin` holds the sum over ranks): solve + retract into the other pose buffer, back-substitute into the
// other point buffer.  accept != 0 makes the new estimate current (Gauss-Newton); 0 leaves it as a trial (LM decides).
int mqs_ba_gn_finish_dev(mqs_ba_problem *p, double lambda, int accept, void *stream)
{
    MQS_ARG_CHECK(p != nullptr, "problem must not be null");
    const int c = p->cur, o = 1 - c;
    // solve + retract + back-substitute: one launch for C <= 4 (ba_tail_kernel), two beyond
    int rc = mqs_ba_solve_backsub_dev(p->lin, p->C, p->poses[c], p->calib, p->sigma, p->points[c], p->obs, p->mask, p->prior_w,
                                      p->prior_xyz, p->N, lambda, p->prior_poses, p->prior_sigmas, p->prior_mask, p->dpose,
                                      p->poses[o], p->info, p->points[o], stream);
    if (rc != MQS_OK) return rc;
    if (accept) p->cur = o;
    return MQS_OK;
}

int mqs_ba_gn_iteration_dev(mqs_ba_problem *p, double lambda, void *stream_)
{
    MQS_ARG_CHECK(p != nullptr, "problem must not be null");
    {
        const int rc = status_so_far(p);      // an earlier iteration's wait gave up: do not build on its output
        if (rc != MQS_OK) return rc;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t nlin = (int64_t)36 * p->C * p->C + 6 * p->C + 2;
    const int c = p->cur, o = 1 - c;
    const bool has_comm = p->ctx && mqs_comm_world_size(p->ctx) >= 1;
    mqs_peer_push push;
    mqs_peer_recv recv;
    int fused_wait = 0;
    // TWO launches (C in 2..4): the wave lineariser, then the tail -- whose first workgroups add the lineariser's partial rows
    // (the finalize, inside the launch), and which then solves, retracts and back-substitutes.  Without a communicator, or over
    // the peer transport with ranks on separate GPUs: there the finalizer pieces also store their quarter sums into every rank's
    // receive buffer and every workgroup waits for all ranks' pieces -- the all-reduce without any launch of its own.
    if (mqs_ba_wave_path(p->C) && mqs_ba_fused_finalize_enabled()) {
        const int pieces = MQS_FIN_PIECES * mqs_ba_finalize_groups(p->C);
        // (the first reduction of a problem over the peer transport takes the path below: a one-workgroup wait with a long bound)
        const bool peer = has_comm && mqs_comm_peer_fused(p->ctx) && p->peer_reductions > 0 && mqs_comm_peer_next(p->ctx, MQS_FIN_PIECES * MQS_PEER_QUARTER_STRIDE, pieces, &push, &recv, &fused_wait);
        if (peer) ++p->peer_reductions;
        if (!has_comm || peer) {
            mqs_ba_fin fin;
            int rc = mqs_ba_linearize_for_fused_tail(p->poses[c], p->calib, p->sigma, p->C, p->points[c], p->obs, p->mask, p->prior_w,
                                                     p->prior_xyz, p->N, lambda, p->ws, p->ws_bytes, stream, &fin);
            if (rc != MQS_OK) return rc;
            if (peer) { fin.push = &push; recv.status = p->status_dev; }
            fin.status = p->status_dev;
            rc = mqs_ba_tail_launch(nullptr, peer ? &recv : nullptr, &fin, p->C, p->poses[c], p->calib, p->sigma, p->points[c], p->obs,
                                    p->mask, p->prior_w, p->prior_xyz, p->N, lambda, p->prior_poses, p->prior_sigmas, p->prior_mask, p->lin,
                                    p->dpose, p->poses[o], p->info, p->points[o], stream);
            if (rc != MQS_OK) return rc;
            p->cur = o;
            return MQS_OK;
        }
    }
    // Peer transport with ranks that share a GPU (tests), or the finalize kept as a launch: the finalize kernel stores this rank's
    // reduced system into every rank's receive buffer, a one-workgroup kernel (or, MQS_PEER_FUSED=1, the tail) waits and adds.
    if (p->ctx && mqs_comm_peer_next(p->ctx, nlin, mqs_ba_finalize_groups(p->C), &push, &recv, &fused_wait)) {
        recv.status = p->status_dev;
        const bool first = p->peer_reductions++ == 0;
        if (first) { recv.spin_ticks = mqs::peer::kSpinTicksFirst; fused_wait = 0; }
        int rc = mqs_ba_linearize_push(p->poses[c], p->calib, p->sigma, p->C, p->points[c], p->obs, p->mask, p->prior_w,
                                       p->prior_xyz, p->N, lambda, p->lin, p->ws, p->ws_bytes, stream, &push);
        if (rc != MQS_OK) return rc;
        if (fused_wait && p->C <= 4) {
            rc = mqs_ba_tail_launch(nullptr, &recv, nullptr, p->C, p->poses[c], p->calib, p->sigma, p->points[c], p->obs, p->mask, p->prior_w,
                                    p->prior_xyz, p->N, lambda, p->prior_poses, p->prior_sigmas, p->prior_mask, p->lin, p->dpose,
                                    p->poses[o], p->info, p->points[o], stream);
            if (rc != MQS_OK) return rc;
            p->cur = o;
            return MQS_OK;
        }
        rc = mqs_comm_peer_gather(&recv, p->lin, nlin, stream);
        if (rc != MQS_OK) return rc;
        return mqs_ba_gn_finish_dev(p, lambda, 1, stream_);
    }
    int rc = mqs_ba_gn_begin_dev(p, lambda, stream_);
    if (rc != MQS_OK) return rc;
    // issued whenever the context holds a communicator -- also a one-rank one, where the sum is the identity: the single-GPU
    // tests then run the very call sequence an N-GPU iteration runs (lineariser, ncclAllReduce on the same stream, solve)
    if (has_comm) {
        rc = mqs_comm_all_reduce_sum_f64_dev(p->ctx, p->lin, nlin, stream_);
        if (rc != MQS_OK) return rc;
    }
    return mqs_ba_gn_finish_dev(p, lambda, 1, stream_);
}

// `iters` iterations back to back (the benchmark's and the GN driver's inner loop): still no host synchronisation.  Where the
// problem allows (mqs_ba_iterate_eligible: a shard below 400 k landmarks, 2..4 cameras; a single GPU, or the peer transport with
// the wait inside the kernels) the run is lineariser, iters - 1 launches that each finish one iteration and linearise the next
// (ba_iterate_kernel), tail: iters + 1 launches instead of 2 iters, bit-identical estimates.
int mqs_ba_gn_iterations_dev(mqs_ba_problem *p, int iters, double lambda, void *stream_)
{
    MQS_ARG_CHECK(p != nullptr && iters >= 0, "problem must not be null, iters >= 0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int done = 0;
    const bool has_comm = p->ctx && mqs_comm_world_size(p->ctx) >= 1;
    const bool peer_fused = has_comm && mqs_comm_peer_fused(p->ctx);
    if (iters >= 2 && mqs_ba_iterate_eligible(p->C, p->N) && (!has_comm || peer_fused)) {
        {
            const int rc = status_so_far(p);
            if (rc != MQS_OK) return rc;
        }
        // over the peer transport the first reduction of a problem waits in a kernel of its own (mqs_ba_gn_iteration_dev)
        if (peer_fused && p->peer_reductions == 0) {
            const int rc = mqs_ba_gn_iteration_dev(p, lambda, stream_);
            if (rc != MQS_OK) return rc;
            done = 1;
        }
        if (iters - done >= 2) {
            const int pieces = MQS_FIN_PIECES * mqs_ba_finalize_groups(p->C);
            mqs_ba_fin fin;
            int rc = mqs_ba_linearize_for_fused_tail(p->poses[p->cur], p->calib, p->sigma, p->C, p->points[p->cur], p->obs, p->mask, p->prior_w,
                                                     p->prior_xyz, p->N, lambda, p->ws, p->ws_bytes, stream, &fin);
            if (rc != MQS_OK) return rc;
            for (int k = done; k < iters; ++k) {
                const int c = p->cur, o = 1 - c;
                mqs_peer_push push;
                mqs_peer_recv recv;
                int fused_wait = 0;
                const bool peer = peer_fused && mqs_comm_peer_next(p->ctx, MQS_FIN_PIECES * MQS_PEER_QUARTER_STRIDE, pieces, &push, &recv, &fused_wait);
                if (peer_fused && !peer) { mqs_set_error("the peer transport refused a reduction"); return MQS_E_RCCL; }
                if (peer) { ++p->peer_reductions; fin.push = &push; recv.status = p->status_dev; }
                fin.status = p->status_dev;
                if (k + 1 < iters) {
                    mqs_ba_fin next;
                    rc = mqs_ba_iterate_launch(peer ? &recv : nullptr, &fin, p->C, p->poses[c], p->calib, p->sigma, p->points[c], p->obs, p->mask,
                                               p->prior_w, p->prior_xyz, p->N, lambda, p->prior_poses, p->prior_sigmas, p->prior_mask, p->lin,
                                               p->dpose, p->poses[o], p->info, p->points[o], p->ws, stream, &next);
                    if (rc != MQS_OK) return rc;
                    fin = next;
                } else {
                    rc = mqs_ba_tail_launch(nullptr, peer ? &recv : nullptr, &fin, p->C, p->poses[c], p->calib, p->sigma, p->points[c], p->obs,
                                            p->mask, p->prior_w, p->prior_xyz, p->N, lambda, p->prior_poses, p->prior_sigmas, p->prior_mask, p->lin,
                                            p->dpose, p->poses[o], p->info, p->points[o], stream);
                    if (rc != MQS_OK) return rc;
                }
                p->cur = o;
            }
            return MQS_OK;
        }
    }
    for (int k = done; k < iters; ++k) {
        const int rc = mqs_ba_gn_iteration_dev(p, lambda, stream_);
        if (rc != MQS_OK) return rc;
    }
    return MQS_OK;
}

}  // extern "C"
