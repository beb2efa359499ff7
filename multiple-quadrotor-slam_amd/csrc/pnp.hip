// Camera pose from 3D-2D correspondences on gfx950 (MI355X): the pose step of the reference's
// per-frame loop,
//   cv2.solvePnPRansac  Work/SLAM/application/own/slam2.py:453-454
//   cv2.solvePnP        Work/SLAM/application/own/slam2.py:489-490, 576-577, 1156
// Arithmetic: pnp_math.h.
//
// Mapping to the machine.  A frame has at most a few hundred correspondences (slam2.py:1080-1082), so the
// parallel axis is the PROBLEM, not the point: one wavefront per problem (a frame's pose, or one RANSAC
// hypothesis), lanes strided over its correspondences, the 28 sums of the normal equations reduced over
// the wave with lane exchanges, and the whole Levenberg-Marquardt loop (6x6 Cholesky in
// registers, redundantly in every lane) run on-chip: one launch per solvePnP, no host round trips.
// RANSAC is three launches: all hypotheses at once (direct linear transform of 6 sampled points, a few
// LM iterations on them, inlier count over all points), selection + inlier compaction, final refinement.
// Every sum has a fixed order: results are bitwise reproducible.
#include "mqs_common.h"
#include "pnp_math.h"
#include "wave_reduce.h"
#include "pnp_block.h"
#include "cam_math.h"

namespace {

using namespace mqs::pnp;

constexpr int kWave = 64;
using namespace mqs::pnpblk;          // wave_sum_all, Problem, WaveEval, wave_dlt, hypothesis_wave, block_sum_acc, BlockEval, select_refine_block (pnp_block.h)

// info per problem: [sum of squared residuals, LM iterations, number of correspondences, flags]
//   flags bit 0: LM stopped on its convergence test; bit 1: DLT start failed (pose_in used instead)
__global__ __launch_bounds__(kWave) void pnp_refine_kernel(const double *__restrict__ objp, const double *__restrict__ imgp,
                                                          int N, const int32_t *__restrict__ idx,
                                                          const int32_t *__restrict__ ptr, const double *__restrict__ intr,
                                                          const double *__restrict__ poses_in, int use_guess, int max_iter,
                                                          double eps, double *__restrict__ poses_out, double *__restrict__ info)
{
    __shared__ double sI[9];
    __shared__ double sA[132];
    const int lane = threadIdx.x, b = blockIdx.x;
    if (lane < 9) sI[lane] = intr[lane];
    __syncthreads();
    Problem pr = {objp, imgp, idx, ptr ? ptr[b] : 0, ptr ? ptr[b + 1] : N};
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = poses_in ? poses_in[12 * b + k] : ((k % 5 == 0) ? 1.0 : 0.0);
    if (pr.end - pr.begin < 3) {                    // no problem (a rejected frame of the device-resident loop): the start pose
        if (lane < 12) poses_out[12 * b + lane] = P[lane];
        if (info && lane < 4) info[4 * b + lane] = lane == 3 ? 2.0 : 0.0;
        return;
    }
    int flags = 0;
    if (!use_guess) {
        double Pd[12];
        if (pr.end - pr.begin >= 6 && wave_dlt(pr, sI, lane, sA, Pd)) {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = Pd[k];
        } else {
            flags |= 2;
        }
    }
    WaveEval ev = {pr, sI, lane, sA};
    const LmResult r = lm_refine(ev, P, max_iter, eps);
    if (r.converged) flags |= 1;
    if (lane < 12) poses_out[12 * b + lane] = P[lane];
    if (info && lane == 0) {
        info[4 * b + 0] = r.sqerr;
        info[4 * b + 1] = (double)r.iters;
        info[4 * b + 2] = (double)(pr.end - pr.begin);
        info[4 * b + 3] = (double)flags;
    }
}

// One RANSAC hypothesis per wave: DLT of its sample, LM on the sample, inlier count over all N points.
__global__ __launch_bounds__(kWave) void pnp_hypothesis_kernel(const double *__restrict__ objp, const double *__restrict__ imgp,
                                                              int N, const int32_t *__restrict__ n_dev,
                                                              const double *__restrict__ intr,
                                                              const int32_t *__restrict__ samples, int sample_size,
                                                              int sample_iters, double thr2, double *__restrict__ poses,
                                                              int32_t *__restrict__ counts)
{
    __shared__ double sI[9];
    __shared__ double sA[132];
    const int lane = threadIdx.x, h = blockIdx.x;
    if (n_dev) {                                     // the device-resident loop: the number of correspondences is device state
        N = *n_dev;
        if (N < sample_size) {                       // the frame was rejected before the pose step: nothing to do
            if (lane == 0) counts[h] = -1;
            return;
        }
    }
    if (lane < 9) sI[lane] = intr[lane];
    // the sample's correspondences into LDS once: the DLT's four passes and every evaluation of the refinement then read LDS instead
    // of going to the vector cache for the same six points (a round trip on the hypothesis' chain each time)
    __shared__ double sObj[3 * kWave], sImg[2 * kWave];
    const bool staged = sample_size <= kWave;
    if (staged && lane < sample_size) {
        const int i = samples[h * sample_size + lane];
        sObj[3 * lane] = objp[3 * i]; sObj[3 * lane + 1] = objp[3 * i + 1]; sObj[3 * lane + 2] = objp[3 * i + 2];
        sImg[2 * lane] = imgp[2 * i]; sImg[2 * lane + 1] = imgp[2 * i + 1];
    }
    __syncthreads();
    const Problem pr = staged ? Problem{sObj, sImg, nullptr, 0, sample_size} : Problem{objp, imgp, samples, h * sample_size, (h + 1) * sample_size};
    double P[12];
    const int count = hypothesis_wave(pr, objp, imgp, N, sI, sample_iters, thr2, sA, lane, P);
    if (lane < 12) poses[12 * h + lane] = P[lane];
    if (lane == 0) counts[h] = count;
}

__global__ __launch_bounds__(kSelBlock) void pnp_select_refine_kernel(const double *__restrict__ objp, const double *__restrict__ imgp,
                                                                     int N, const int32_t *__restrict__ n_dev,
                                                                     const double *__restrict__ intr,
                                                                     const double *__restrict__ poses, const int32_t *__restrict__ counts,
                                                                     int B, double thr2, int max_iter, double eps, int lds_ok,
                                                                     double *__restrict__ pose_out, int32_t *__restrict__ out_sel,
                                                                     int32_t *__restrict__ inlier_idx, uint8_t *__restrict__ mask,
                                                                     double *__restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double sel_lds[];     // the inliers' coordinates when they fit (lds_ok)
    select_refine_block(objp, imgp, N, n_dev, intr, poses, counts, B, thr2, max_iter, eps, lds_ok, pose_out, out_sel, inlier_idx, mask, info, sel_lds);
}

int check_points(const double *objp, const double *imgp, int64_t N, const double *intr)
{
    MQS_ARG_CHECK(N >= 0 && N <= 0x7fffffff, "0 <= N < 2^31");
    MQS_ARG_CHECK(intr != nullptr, "intr must not be null");
    MQS_ARG_CHECK(N == 0 || (objp && imgp), "objp, imgp must not be null");
    return MQS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// One keyframe step of slam2.py's handle_new_frame (:453-490, 541-590) in ONE launch (one workgroup of four waves):
//   pose P1 = solvePnP(tracked landmarks, start = previous pose)                                      (:489-490)
//   triangulate the not-yet-triangulated tracks against the base keyframe with (P0, P1): undistort both pixel sets
//   (cv2.undistortPoints :551-552), iterative-LS (:553-555), keep status == 1 (:556), cast to float32 (:19)
//   refined pose P2 = solvePnP(old + new points, start = P1)                                            (:576-577)
//   re-triangulate the kept points with (P0, P2) (:582-584); the caller keeps status >= 0 (:589).
// The host-pointer path needs eight calls for this (two pose solves, four undistortions, two triangulations) at <= 300
// points each: launch and copy latency, not arithmetic.  Same device functions as the single kernels (pnp_math.h lm_refine,
// cam_math.h undistort_pixel, tri_math.h iterative_ls_point<2>); the kept points are compacted in index order behind the
// old ones.  The pose sums run over four waves instead of one (a different, still fixed, order): the results equal the
// eight-call path's to rounding (1e-12 asserted, tests/test_replay.py).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int32_t kKfDropped = -128;              // status of a point the first pass did not keep

// eval over two segments (the tracked landmarks, then the compacted new points), summed over the WORKGROUP in a fixed order:
// every thread ends with the same sums, so the Levenberg-Marquardt loop around it runs redundantly and in step in all of them
struct KfEval {
    const double *objp, *imgp;
    int n_old;
    const double *cx, *cuv;                       // compacted new points: [n_ok][3] (float32-rounded), [n_ok][2]
    int n_ok;
    const double *intr;
    double *red;                                  // LDS [kKfWaves][kAcc]
    int tid;
    __device__ __forceinline__ void operator()(const double *P, double *acc) const
    {
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        const int n = n_old + n_ok;
        for (int k = tid; k < n; k += kKfThreads) {
            const bool old = k < n_old;
            const double *X = old ? objp + 3 * k : cx + 3 * (k - n_old);
            const double *U = old ? imgp + 2 * k : cuv + 2 * (k - n_old);
            accumulate_point(P, intr, X[0], X[1], X[2], U[0], U[1], acc);
        }
        block_sum_acc(acc, red, tid);
    }
};

__global__ __launch_bounds__(kKfThreads) void keyframe_step_kernel(
    const double *objp, const double *imgp, int n_old, const double *__restrict__ p0,
    const double *__restrict__ p1, int n_new, const double *__restrict__ intr, const double *__restrict__ P_prev,
    const double *__restrict__ P0, double tol, int max_iter, double eps, double screen_px, int lds_ok, double *scratch /* 9 n_new + 1 doubles */,
    double *__restrict__ pose_out /* 24: first pose, final pose */, double *__restrict__ x_out, int32_t *__restrict__ status_out,
    double *__restrict__ info /* 8: the two solves' info */)
{
    __shared__ double sI[9];
    __shared__ double sP[24];                     // P0 | the frame's pose, as the 2-view triangulation wants them
    __shared__ double sRed[kKfWaves * kAcc];
    __shared__ int sCnt[kKfWaves];
    extern __shared__ __attribute__((aligned(16))) double kf_lds[];      // correspondences + scratch when the step fits (lds_ok)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 9) sI[tid] = intr[tid];
    if (tid < 12) sP[tid] = P0[tid];
    if (lds_ok) {
        // Real frames (<= 300 correspondences, slam2.py:1080-1082) fit the LDS: the correspondences are read once and every
        // LM iteration then sweeps LDS instead of global memory; the scratch arrays live there too.
        double *so = kf_lds, *si = so + 3 * n_old;
        for (int k = tid; k < 3 * n_old; k += kKfThreads) so[k] = objp[k];
        for (int k = tid; k < 2 * n_old; k += kKfThreads) si[k] = imgp[k];
        objp = so; imgp = si;
        scratch = si + 2 * n_old;
    }
    __syncthreads();
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = P_prev[k];
    KfEval ev = {objp, imgp, n_old, nullptr, nullptr, 0, sI, sRed, tid};
    LmResult r = lm_refine(ev, P, max_iter, eps);
    if (tid < 12) { pose_out[tid] = P[tid]; pose_out[12 + tid] = P[tid]; }
    if (tid == 0) { info[0] = r.sqerr; info[1] = (double)r.iters; info[2] = (double)n_old; info[3] = r.converged ? 1.0 : 0.0;
                    info[4] = 0.0; info[5] = 0.0; info[6] = 0.0; info[7] = 0.0; }
    if (n_new <= 0) return;

    double *un = scratch;                         // [n_new][4] undistorted (x0, y0, x1, y1)
    double *cx = scratch + 4 * (size_t)n_new;     // [n_ok][3]
    double *cuv = cx + 3 * (size_t)n_new;         // [n_ok][2]
    int32_t *cidx = reinterpret_cast<int32_t *>(cuv + 2 * (size_t)n_new);   // [n_ok]
    if (tid < 12) sP[12 + tid] = P[tid];
    __syncthreads();
    // first triangulation; the kept points are compacted in index order (ballot ranks per wave, wave offsets through LDS),
    // float32-rounded
    int n_ok = 0;
    for (int base = 0; base < n_new; base += kKfThreads) {
        const int k = base + tid;
        bool ok = false;
        mqs::Vec3 x = {0, 0, 0};
        if (k < n_new) {
            double uv[2][2];
            mqs::cam::undistort_pixel(sI, p0[2 * k], p0[2 * k + 1], uv[0][0], uv[0][1]);
            mqs::cam::undistort_pixel(sI, p1[2 * k], p1[2 * k + 1], uv[1][0], uv[1][1]);
            un[4 * k + 0] = uv[0][0]; un[4 * k + 1] = uv[0][1]; un[4 * k + 2] = uv[1][0]; un[4 * k + 3] = uv[1][1];
            int32_t st;
            x = mqs::iterative_ls_point<2>(uv, sP, tol, MQS_TRI_MAX_ITER_DEFAULT, st);
            ok = st == 1;
            if (ok && screen_px > 0.0) {
                // OPTIONAL (off in the reference's flow: slam2.py:1092 defines max_2nd_solvePnP_reproj_error "used in 2nd iteration,
                // after 1st pass of triangulation" and never uses it): a fresh point that misses its own measurement in THIS frame by
                // more than the bound, under the pose it was triangulated with, is not handed to the second solvePnP -- status 1 says
                // converged and in front of both cameras, not small residuals (a point at a camera centre has both and any residual)
                double a28[kAcc];
#pragma unroll
                for (int q = 0; q < kAcc; ++q) a28[q] = 0.0;
                accumulate_point(sP + 12, sI, (double)(float)x.x, (double)(float)x.y, (double)(float)x.z, p1[2 * k], p1[2 * k + 1], a28);
                ok = a28[27] <= screen_px * screen_px;             // (NaN: dropped)
            }
            if (!ok) {
                status_out[k] = kKfDropped;
                x_out[3 * k] = x_out[3 * k + 1] = x_out[3 * k + 2] = __builtin_nan("");
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) sCnt[wave] = __popcll(m);
        __syncthreads();
        int before = n_ok, total = 0;
#pragma unroll
        for (int w = 0; w < kKfWaves; ++w) {
            if (w < wave) before += sCnt[w];
            total += sCnt[w];
        }
        const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
        if (ok) {
            cx[3 * rank + 0] = (double)(float)x.x; cx[3 * rank + 1] = (double)(float)x.y; cx[3 * rank + 2] = (double)(float)x.z;
            cuv[2 * rank + 0] = p1[2 * k]; cuv[2 * rank + 1] = p1[2 * k + 1];
            cidx[rank] = k;
        }
        n_ok += total;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // refined pose on old + kept points
    KfEval ev2 = {objp, imgp, n_old, cx, cuv, n_ok, sI, sRed, tid};
    r = lm_refine(ev2, P, max_iter, eps);
    __syncthreads();
    if (tid < 12) { pose_out[12 + tid] = P[tid]; sP[12 + tid] = P[tid]; }
    if (tid == 0) { info[4] = r.sqerr; info[5] = (double)r.iters; info[6] = (double)(n_old + n_ok); info[7] = r.converged ? 1.0 : 0.0; }
    __syncthreads();
    // second triangulation of the kept points with the refined pose
    for (int j = tid; j < n_ok; j += kKfThreads) {
        const int k = cidx[j];
        double uv[2][2] = {{un[4 * k + 0], un[4 * k + 1]}, {un[4 * k + 2], un[4 * k + 3]}};
        int32_t st;
        const mqs::Vec3 x = mqs::iterative_ls_point<2>(uv, sP, tol, MQS_TRI_MAX_ITER_DEFAULT, st);
        x_out[3 * k + 0] = x.x; x_out[3 * k + 1] = x.y; x_out[3 * k + 2] = x.z;
        status_out[k] = st;
    }
}

}  // namespace

extern "C" {

int mqs_pnp_refine_dev(const double *objp, const double *imgp, int64_t N, const int32_t *idx, const int32_t *ptr, int B,
                       const double *intr, const double *poses_in, int use_guess, int max_iter, double eps,
                       double *poses_out, double *info, void *stream_)
{
    int rc = check_points(objp, imgp, N, intr);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(B >= 1, "B >= 1");
    MQS_ARG_CHECK(ptr != nullptr || B == 1, "ptr required when B > 1");
    MQS_ARG_CHECK(poses_out != nullptr, "poses_out must not be null");
    MQS_ARG_CHECK(!use_guess || poses_in, "poses_in required with use_guess");
    MQS_ARG_CHECK(max_iter >= 0 && eps >= 0.0, "max_iter >= 0, eps >= 0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(pnp_refine_kernel, dim3(B), dim3(kWave), 0, stream, objp, imgp, (int)N, idx, ptr, intr, poses_in,
                       use_guess, max_iter, eps, poses_out, info);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int64_t mqs_pnp_workspace_bytes(int64_t N, int B)
{
    if (N < 0 || B < 1) return 0;
    // hypothesis poses, inlier counts, {0, inlier count}, inlier list (each 256-byte aligned)
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    return up((int64_t)B * 96) + up((int64_t)B * 4) + up(8) + up(N * 4);
}

int mqs_pnp_ransac_dev(const double *objp, const double *imgp, int64_t N, const double *intr, const int32_t *samples,
                       int B, int sample_size, double reproj_error, int sample_iters, int max_iter, double eps,
                       double *pose_out, int32_t *sel_out, uint8_t *mask, double *info, void *workspace,
                       int64_t workspace_bytes, void *stream_)
{
    int rc = check_points(objp, imgp, N, intr);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(B >= 1 && samples != nullptr, "B >= 1 hypotheses with samples");
    MQS_ARG_CHECK(sample_size >= 6 && sample_size <= N, "6 <= sample_size <= N");
    MQS_ARG_CHECK(reproj_error >= 0.0 && sample_iters >= 0 && max_iter >= 0 && eps >= 0.0, "non-negative parameters");
    MQS_ARG_CHECK(pose_out && sel_out, "pose_out, sel_out must not be null");
    MQS_ARG_CHECK(workspace && workspace_bytes >= mqs_pnp_workspace_bytes(N, B), "workspace too small (mqs_pnp_workspace_bytes)");
    return mqs_pnp_ransac_launch(objp, imgp, (int)N, nullptr, intr, samples, B, sample_size, reproj_error, sample_iters, max_iter, eps,
                                 pose_out, sel_out, mask, info, workspace, static_cast<hipStream_t>(stream_), 0);
}

}  // extern "C"

// N = capacity (workspace sized for it); n_dev (device, may be null): the live number of correspondences
void mqs_pnp_workspace_layout(void *workspace, int B, double **poses, int32_t **counts, int32_t **inlier_idx)
{
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    char *w = static_cast<char *>(workspace);
    *poses = reinterpret_cast<double *>(w); w += up((int64_t)B * 96);
    *counts = reinterpret_cast<int32_t *>(w); w += up((int64_t)B * 4);
    w += up(8);
    *inlier_idx = reinterpret_cast<int32_t *>(w);
}

// end_in_caller: only the hypotheses are launched; the caller runs select_refine_block (pnp_block.h) in a kernel of its own (the
// device-resident loop: inside its decision kernel)
int mqs_pnp_ransac_launch(const double *objp, const double *imgp, int N, const int32_t *n_dev, const double *intr,
                          const int32_t *samples, int B, int sample_size, double reproj_error, int sample_iters, int max_iter,
                          double eps, double *pose_out, int32_t *sel_out, uint8_t *mask, double *info, void *workspace,
                          hipStream_t stream, int end_in_caller)
{
    double *poses;
    int32_t *counts, *inl;
    mqs_pnp_workspace_layout(workspace, B, &poses, &counts, &inl);
    const double thr2 = reproj_error * reproj_error;
    hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3(B), dim3(kWave), 0, stream, objp, imgp, N, n_dev, intr, samples,
                       sample_size, sample_iters, thr2, poses, counts);
    if (!end_in_caller) {
        // best hypothesis, its inliers (-> mask), and OpenCV 2.4 solvePnPRansac's end -- solvePnP on the inliers, started from the best model --
        // in one launch; the inliers' coordinates in LDS when N correspondences fit (40 bytes each)
        const size_t lds_bytes = (size_t)N * 40;
        const bool lds_ok = lds_bytes <= 40 * 1024;
        hipLaunchKernelGGL(pnp_select_refine_kernel, dim3(1), dim3(kSelBlock), lds_ok ? lds_bytes : 0, stream, objp, imgp, N, n_dev, intr, poses,
                           counts, B, thr2, max_iter, eps, (int)lds_ok, pose_out, sel_out, inl, mask, info);
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// the keyframe step on device-resident inputs (scratch: 9 n_new + 1 doubles + n_new ints when the step does not fit the LDS)
int mqs_keyframe_step_launch(const double *objp, const double *imgp, int n_old, const double *p0, const double *p1, int n_new,
                             const double *intr, const double *P_prev, const double *P0, double tolerance, int max_iter, double eps,
                             double second_pass_screen_px, double *scratch, double *pose_out, double *x_out, int32_t *status_out, double *info,
                             hipStream_t stream)
{
    const size_t lds_bytes = ((size_t)5 * n_old + (size_t)9 * n_new + 2) * 8 + (size_t)n_new * 4;
    const bool lds_ok = lds_bytes <= 56 * 1024;
    hipLaunchKernelGGL(keyframe_step_kernel, dim3(1), dim3(kKfThreads), lds_ok ? lds_bytes : 0, stream, objp, imgp, n_old, p0, p1, n_new,
                       intr, P_prev, P0, tolerance, max_iter, eps, second_pass_screen_px, (int)lds_ok, scratch, pose_out, x_out, status_out, info);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// ---------------------------------------------------------------------------------------
// Host-pointer wrappers
// ---------------------------------------------------------------------------------------
extern "C" {

int mqs_solve_pnp(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t N, const double *intr, double *pose,
                  int use_guess, int max_iter, double eps, double *info)
{
    MQS_ARG_CHECK(ctx != nullptr && pose != nullptr, "ctx, pose must not be null");
    int rc = check_points(objp, imgp, N, intr);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(N >= (use_guess ? 3 : 6), "solvePnP needs >= 6 points (>= 3 with a starting pose)");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_obj = 0, o_img = up((size_t)N * 24), o_intr = o_img + up((size_t)N * 16), o_pose = o_intr + 256,
                 o_info = o_pose + 256, total = o_info + 256;
    // 2 * total up front: should the pinned buffer be unavailable, mqs_stage_begin falls back to the device scratch with that
    // size -- reserving it here means the scratch can no longer move after `d` has been read
    rc = mqs_ctx_reserve(ctx, 2 * total);
    if (rc != MQS_OK) return rc;
    hipStream_t s = ctx->stream;
    // Small problems (every real frame: <= 300 correspondences): the inputs are packed into the pinned host buffer by the
    // CPU and travel as ONE copy; the kernel keeps them in device memory (it re-reads them in every LM iteration) and
    // writes the 16 doubles of its result straight into the pinned buffer -- one hipMemcpy per call instead of six.
    mqs_stage st;
    st.zero_copy = false;
    if (2 * total <= kZeroCopyMax) {
        rc = mqs_stage_begin(ctx, 2 * total, &st);
        if (rc != MQS_OK) return rc;
    }
    char *d = static_cast<char *>(ctx->dbuf);          // read AFTER every call that may (re)allocate the scratch
    if (st.zero_copy) {
        char *h = st.base;
        memcpy(h + o_obj, objp, (size_t)N * 24);
        memcpy(h + o_img, imgp, (size_t)N * 16);
        memcpy(h + o_intr, intr, 72);
        memcpy(h + o_pose, pose, 96);
        MQS_HIP_CHECK(hipMemcpyAsync(d, h, o_pose + 96, hipMemcpyHostToDevice, s));
        double *h_pose = reinterpret_cast<double *>(h + total + o_pose), *h_info = reinterpret_cast<double *>(h + total + o_info);
        rc = mqs_pnp_refine_dev((double *)(d + o_obj), (double *)(d + o_img), N, nullptr, nullptr, 1, (double *)(d + o_intr),
                                (double *)(d + o_pose), use_guess, max_iter, eps, h_pose, h_info, s);
        if (rc != MQS_OK) return rc;
        MQS_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(pose, h_pose, 96);
        if (info) memcpy(info, h_info, 32);
        return MQS_OK;
    }
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_obj, objp, (size_t)N * 24, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_img, imgp, (size_t)N * 16, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_intr, intr, 72, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_pose, pose, 96, hipMemcpyHostToDevice, s));
    rc = mqs_pnp_refine_dev((double *)(d + o_obj), (double *)(d + o_img), N, nullptr, nullptr, 1, (double *)(d + o_intr),
                            (double *)(d + o_pose), use_guess, max_iter, eps, (double *)(d + o_pose), (double *)(d + o_info), s);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(pose, d + o_pose, 96, hipMemcpyDeviceToHost, s));
    if (info) MQS_HIP_CHECK(hipMemcpyAsync(info, d + o_info, 32, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

int mqs_keyframe_step(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t n_old, const double *p0, const double *p1,
                      int64_t n_new, const double *intr, const double *P_prev, const double *P0, double tolerance, int max_iter,
                      double eps, double *poses /* [2][12]: first, refined */, double *x, int32_t *status, double *info)
{
    MQS_ARG_CHECK(ctx != nullptr && poses != nullptr && intr != nullptr && P_prev != nullptr, "ctx, poses, intr, P_prev must not be null");
    MQS_ARG_CHECK(n_old >= 3 && objp && imgp, "a pose needs >= 3 tracked landmarks (objp, imgp)");
    MQS_ARG_CHECK(n_new >= 0 && n_old < (1 << 24) && n_new < (1 << 24), "sizes");
    MQS_ARG_CHECK(n_new == 0 || (p0 && p1 && P0 && x && status), "p0, p1, P0, x, status must not be null when there are new points");
    MQS_ARG_CHECK(max_iter >= 1, "max_iter >= 1");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    // inputs (one copy) | scratch | outputs (read back, or written straight into the pinned buffer)
    const size_t o_obj = 0, o_img = o_obj + up((size_t)n_old * 24), o_p0 = o_img + up((size_t)n_old * 16),
                 o_p1 = o_p0 + up((size_t)n_new * 16), o_intr = o_p1 + up((size_t)n_new * 16), o_pp = o_intr + 256, o_pb = o_pp + 256,
                 in_bytes = o_pb + 256;
    const size_t o_scr = in_bytes, scr_bytes = up((size_t)n_new * (9 * 8 + 4) + 64);
    const size_t o_pose = o_scr + scr_bytes, o_x = o_pose + 256, o_st = o_x + up((size_t)n_new * 24), o_info = o_st + up((size_t)n_new * 4),
                 total = o_info + 256;
    int rc = mqs_ctx_reserve(ctx, 2 * total);
    if (rc != MQS_OK) return rc;
    hipStream_t s = ctx->stream;
    mqs_stage st;
    st.zero_copy = false;
    if (2 * total <= kZeroCopyMax) {
        rc = mqs_stage_begin(ctx, 2 * total, &st);
        if (rc != MQS_OK) return rc;
    }
    char *d = static_cast<char *>(ctx->dbuf);
    const double dummyP[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const size_t lds_bytes = ((size_t)5 * n_old + (size_t)9 * n_new + 2) * 8 + (size_t)n_new * 4;   // correspondences | un, cx, cuv | cidx
    const bool lds_ok = lds_bytes <= 56 * 1024;
    auto launch = [&](const char *inb, char *outb) {
        hipLaunchKernelGGL(keyframe_step_kernel, dim3(1), dim3(kKfThreads), lds_ok ? lds_bytes : 0, s, (const double *)(inb + o_obj),
                           (const double *)(inb + o_img), (int)n_old, (const double *)(inb + o_p0), (const double *)(inb + o_p1), (int)n_new,
                           (const double *)(inb + o_intr), (const double *)(inb + o_pp), (const double *)(inb + o_pb), tolerance, max_iter,
                           eps, 0.0 /* no screen: slam2.py's flow */, (int)lds_ok, (double *)(d + o_scr), (double *)(outb + o_pose), (double *)(outb + o_x),
                           (int32_t *)(outb + o_st), (double *)(outb + o_info));
    };
    if (st.zero_copy) {
        // the inputs are packed in the pinned buffer by the CPU and travel as ONE copy (the kernel re-reads them in every LM
        // iteration: they stay in device memory); the outputs are written straight into the pinned buffer's second half
        char *h = st.base;
        memcpy(h + o_obj, objp, (size_t)n_old * 24);
        memcpy(h + o_img, imgp, (size_t)n_old * 16);
        if (n_new) { memcpy(h + o_p0, p0, (size_t)n_new * 16); memcpy(h + o_p1, p1, (size_t)n_new * 16); }
        memcpy(h + o_intr, intr, 72);
        memcpy(h + o_pp, P_prev, 96);
        memcpy(h + o_pb, P0 ? P0 : dummyP, 96);
        // (letting the single wave pull the inputs over the fabric itself, straight into LDS, measured slower than this
        // copy: 113 vs 104 us per frame)
        MQS_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
        launch(d, h + total);
        MQS_HIP_CHECK(hipGetLastError());
        MQS_HIP_CHECK(hipStreamSynchronize(s));
        const char *o = h + total;
        memcpy(poses, o + o_pose, 192);
        if (n_new) { memcpy(x, o + o_x, (size_t)n_new * 24); memcpy(status, o + o_st, (size_t)n_new * 4); }
        if (info) memcpy(info, o + o_info, 64);
        return MQS_OK;
    }
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_obj, objp, (size_t)n_old * 24, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_img, imgp, (size_t)n_old * 16, hipMemcpyHostToDevice, s));
    if (n_new) {
        MQS_HIP_CHECK(hipMemcpyAsync(d + o_p0, p0, (size_t)n_new * 16, hipMemcpyHostToDevice, s));
        MQS_HIP_CHECK(hipMemcpyAsync(d + o_p1, p1, (size_t)n_new * 16, hipMemcpyHostToDevice, s));
    }
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_intr, intr, 72, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_pp, P_prev, 96, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_pb, P0 ? P0 : dummyP, 96, hipMemcpyHostToDevice, s));
    launch(d, d);
    MQS_HIP_CHECK(hipGetLastError());
    MQS_HIP_CHECK(hipMemcpyAsync(poses, d + o_pose, 192, hipMemcpyDeviceToHost, s));
    if (n_new) {
        MQS_HIP_CHECK(hipMemcpyAsync(x, d + o_x, (size_t)n_new * 24, hipMemcpyDeviceToHost, s));
        MQS_HIP_CHECK(hipMemcpyAsync(status, d + o_st, (size_t)n_new * 4, hipMemcpyDeviceToHost, s));
    }
    if (info) MQS_HIP_CHECK(hipMemcpyAsync(info, d + o_info, 64, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

int mqs_solve_pnp_ransac(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t N, const double *intr,
                         const int32_t *samples, int B, int sample_size, double reproj_error, int sample_iters,
                         int max_iter, double eps, double *pose, int32_t *sel, uint8_t *mask, double *info)
{
    MQS_ARG_CHECK(ctx != nullptr && pose != nullptr && sel != nullptr, "ctx, pose, sel must not be null");
    int rc = check_points(objp, imgp, N, intr);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(B >= 1 && samples != nullptr, "B >= 1 hypotheses with samples");
    MQS_ARG_CHECK(sample_size >= 6 && sample_size <= N, "6 <= sample_size <= N");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t wsb = (size_t)mqs_pnp_workspace_bytes(N, B);
    const size_t o_obj = 0, o_img = up((size_t)N * 24), o_intr = o_img + up((size_t)N * 16), o_pose = o_intr + 256,
                 o_info = o_pose + 256, o_sel = o_info + 256, o_mask = o_sel + 256, o_smp = o_mask + up((size_t)N),
                 o_ws = o_smp + up((size_t)B * sample_size * 4), total = o_ws + wsb;
    rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    hipStream_t s = ctx->stream;
    // small problems: inputs packed by the CPU into the pinned buffer and sent as one copy, the four small results fetched
    // as one copy (two hipMemcpy per call instead of eight; the kernels themselves stay on device memory)
    const size_t io_bytes = o_smp + (size_t)B * sample_size * 4;
    mqs_stage st;
    st.zero_copy = false;
    if (io_bytes <= kZeroCopyMax) {
        rc = mqs_stage_begin(ctx, io_bytes, &st);
        if (rc != MQS_OK) return rc;
    }
    if (st.zero_copy) {
        char *h = st.base;
        memcpy(h + o_obj, objp, (size_t)N * 24);
        memcpy(h + o_img, imgp, (size_t)N * 16);
        memcpy(h + o_intr, intr, 72);
        memcpy(h + o_smp, samples, (size_t)B * sample_size * 4);
        MQS_HIP_CHECK(hipMemcpyAsync(d, h, io_bytes, hipMemcpyHostToDevice, s));
        rc = mqs_pnp_ransac_dev((double *)(d + o_obj), (double *)(d + o_img), N, (double *)(d + o_intr), (int32_t *)(d + o_smp), B,
                                sample_size, reproj_error, sample_iters, max_iter, eps, (double *)(d + o_pose),
                                (int32_t *)(d + o_sel), (uint8_t *)(d + o_mask), (double *)(d + o_info), d + o_ws, (int64_t)wsb, s);
        if (rc != MQS_OK) return rc;
        MQS_HIP_CHECK(hipMemcpyAsync(h + o_pose, d + o_pose, o_smp - o_pose, hipMemcpyDeviceToHost, s));
        MQS_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(pose, h + o_pose, 96);
        memcpy(sel, h + o_sel, 8);
        if (mask) memcpy(mask, h + o_mask, (size_t)N);
        if (info) memcpy(info, h + o_info, 32);
        return MQS_OK;
    }
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_obj, objp, (size_t)N * 24, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_img, imgp, (size_t)N * 16, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_intr, intr, 72, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_smp, samples, (size_t)B * sample_size * 4, hipMemcpyHostToDevice, s));
    rc = mqs_pnp_ransac_dev((double *)(d + o_obj), (double *)(d + o_img), N, (double *)(d + o_intr), (int32_t *)(d + o_smp), B,
                            sample_size, reproj_error, sample_iters, max_iter, eps, (double *)(d + o_pose),
                            (int32_t *)(d + o_sel), (uint8_t *)(d + o_mask), (double *)(d + o_info), d + o_ws, (int64_t)wsb, s);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(pose, d + o_pose, 96, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipMemcpyAsync(sel, d + o_sel, 8, hipMemcpyDeviceToHost, s));
    if (mask) MQS_HIP_CHECK(hipMemcpyAsync(mask, d + o_mask, (size_t)N, hipMemcpyDeviceToHost, s));
    if (info) MQS_HIP_CHECK(hipMemcpyAsync(info, d + o_info, 32, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

}  // extern "C"
