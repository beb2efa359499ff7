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

namespace {

using namespace mqs::pnp;

constexpr int kWave = 64;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) v += __shfl_xor(v, h);
    return v;
}

// One problem's correspondences: points idx[begin..end) of (objp, imgp), or begin..end directly.
struct Problem {
    const double *objp, *imgp;
    const int32_t *idx;
    int begin, end;
    __device__ __forceinline__ int point(int k) const { return idx ? idx[k] : k; }
};

// eval(P, acc): sums of pnp_math.h accumulate_point over the problem, identical in every lane.
struct WaveEval {
    Problem pr;
    const double *intr;     // LDS
    int lane;
    __device__ __forceinline__ void operator()(const double *P, double *acc) const
    {
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        for (int k = pr.begin + lane; k < pr.end; k += kWave) {
            const int i = pr.point(k);
            accumulate_point(P, intr, pr.objp[3 * i], pr.objp[3 * i + 1], pr.objp[3 * i + 2], pr.imgp[2 * i],
                             pr.imgp[2 * i + 1], acc);
        }
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = wave_sum(acc[k]);
    }
};

// Direct linear transform start over the problem's points (>= 6); sA: 121 + 11 doubles of LDS per wave.
// Returns false (in every lane) for a degenerate configuration.
__device__ bool wave_dlt(const Problem &pr, const double *intr, int lane, double *sA, double *P)
{
    const int n = pr.end - pr.begin;
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kWave) {
        const int i = pr.point(k);
        cx += pr.objp[3 * i]; cy += pr.objp[3 * i + 1]; cz += pr.objp[3 * i + 2];
    }
    const double inv_n = 1.0 / (double)n;
    const double c[3] = {wave_sum(cx) * inv_n, wave_sum(cy) * inv_n, wave_sum(cz) * inv_n};
    double dist = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kWave) {
        const int i = pr.point(k);
        const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
        dist += sqrt(fma(dx, dx, fma(dy, dy, dz * dz)));
    }
    double cov[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = pr.begin + lane; k < pr.end; k += kWave) {
        const int i = pr.point(k);
        const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
        cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz; cov[3] += dy * dy; cov[4] += dy * dz; cov[5] += dz * dz;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = wave_sum(cov[k]);
    double sigma = wave_sum(dist) * inv_n;
    if (!(sigma > 0.0)) sigma = 1.0;
    const double is = 1.0 / sigma;
    double ew[3], E[9];
    sym3_eigen(cov, ew, E);
    if (ew[2] < 1e-3 * ew[1]) {
        // planar (OpenCV: W[2] / W[1] < 1e-3): start from the plane-to-image homography (needs >= 4 points)
        double hacc[kHomAcc];
#pragma unroll
        for (int k = 0; k < kHomAcc; ++k) hacc[k] = 0.0;
        for (int k = pr.begin + lane; k < pr.end; k += kWave) {
            const int i = pr.point(k);
            const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
            double x, y;
            mqs::cam::undistort_pixel(intr, pr.imgp[2 * i], pr.imgp[2 * i + 1], x, y);
            hom_accumulate((E[0] * dx + E[1] * dy + E[2] * dz) * is, (E[3] * dx + E[4] * dy + E[5] * dz) * is, x, y, hacc);
        }
#pragma unroll
        for (int k = 0; k < kHomAcc; ++k) hacc[k] = wave_sum(hacc[k]);
        double *sb = sA + 64;
        if (lane == 0) {
            hom_assemble(hacc, sA, sb);
            const bool ok = chol_solve_small(sA, sb, 8);
            sA[0] = ok ? 1.0 : 0.0;
        }
        mqs_wave_lds_sync();
        double hv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) hv[k] = sb[k];
        const bool solved = sA[0] != 0.0;
        __builtin_amdgcn_wave_barrier();
        const bool posed = pose_from_homography(hv, E, c, sigma, P);
        return solved && posed;
    }
    double acc[kDltAcc];
#pragma unroll
    for (int k = 0; k < kDltAcc; ++k) acc[k] = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kWave) {
        const int i = pr.point(k);
        double x, y;
        mqs::cam::undistort_pixel(intr, pr.imgp[2 * i], pr.imgp[2 * i + 1], x, y);
        dlt_accumulate((pr.objp[3 * i] - c[0]) * is, (pr.objp[3 * i + 1] - c[1]) * is, (pr.objp[3 * i + 2] - c[2]) * is, x, y, acc);
    }
#pragma unroll
    for (int k = 0; k < kDltAcc; ++k) acc[k] = wave_sum(acc[k]);
    // 11 x 11 solve: serial, one lane, matrix in LDS (dynamic indexing), result broadcast through LDS
    double *sb = sA + 121;
    if (lane == 0) {
        dlt_assemble(acc, sA, sb);
        const bool ok = chol_solve_small(sA, sb, 11);
        sA[0] = ok ? 1.0 : 0.0;
    }
    mqs_wave_lds_sync();
    double p[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) p[k] = sb[k];
    const bool solved = sA[0] != 0.0;
    __builtin_amdgcn_wave_barrier();
    const bool posed = pose_from_dlt(p, c, sigma, P);
    return solved && posed;
}

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
    WaveEval ev = {pr, sI, lane};
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
                                                              int N, const double *__restrict__ intr,
                                                              const int32_t *__restrict__ samples, int sample_size,
                                                              int sample_iters, double thr2, double *__restrict__ poses,
                                                              int32_t *__restrict__ counts)
{
    __shared__ double sI[9];
    __shared__ double sA[132];
    const int lane = threadIdx.x, h = blockIdx.x;
    if (lane < 9) sI[lane] = intr[lane];
    __syncthreads();
    const Problem pr = {objp, imgp, samples, h * sample_size, (h + 1) * sample_size};
    double P[12];
    int count = -1;
    if (wave_dlt(pr, sI, lane, sA, P)) {
        WaveEval ev = {pr, sI, lane};
        lm_refine(ev, P, sample_iters, 1e-10);
        int c = 0;
        for (int i = lane; i < N; i += kWave) {
            // behind the camera: never an inlier
            const double Zc = fma(P[8], objp[3 * i], fma(P[9], objp[3 * i + 1], fma(P[10], objp[3 * i + 2], P[11])));
            const double e2 = reproj_sqerr(P, sI, objp[3 * i], objp[3 * i + 1], objp[3 * i + 2], imgp[2 * i], imgp[2 * i + 1]);
            c += (Zc > 0.0 && e2 <= thr2) ? 1 : 0;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s);
        count = c;
    }
    if (lane < 12) poses[12 * h + lane] = P[lane];
    if (lane == 0) counts[h] = count;
}

// Picks the hypothesis with the most inliers (lowest index on ties), marks and compacts its inliers.
// out_sel: [0] best hypothesis (-1: none valid), [1] inlier count; ptr2 = {0, inlier count}.
constexpr int kSelBlock = 256;
__global__ __launch_bounds__(kSelBlock) void pnp_select_kernel(const double *__restrict__ objp, const double *__restrict__ imgp,
                                                              int N, const double *__restrict__ intr,
                                                              const double *__restrict__ poses, const int32_t *__restrict__ counts,
                                                              int B, double thr2, double *__restrict__ best_pose,
                                                              int32_t *__restrict__ out_sel, int32_t *__restrict__ ptr2,
                                                              int32_t *__restrict__ inlier_idx, uint8_t *__restrict__ mask)
{
    __shared__ double sI[9], sP[12];
    __shared__ int sBestC[kSelBlock], sBestH[kSelBlock], sWave[kSelBlock / 64], sBase;
    const int tid = threadIdx.x;
    if (tid < 9) sI[tid] = intr[tid];
    int bc = -1, bh = -1;
    for (int h = tid; h < B; h += kSelBlock)
        if (counts[h] > bc) { bc = counts[h]; bh = h; }
    sBestC[tid] = bc; sBestH[tid] = bh;
    __syncthreads();
    for (int s = kSelBlock / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            const int c2 = sBestC[tid + s], h2 = sBestH[tid + s];
            if (c2 > sBestC[tid] || (c2 == sBestC[tid] && h2 >= 0 && (sBestH[tid] < 0 || h2 < sBestH[tid]))) {
                sBestC[tid] = c2; sBestH[tid] = h2;
            }
        }
        __syncthreads();
    }
    const int best = sBestH[0];
    if (tid < 12) sP[tid] = best >= 0 ? poses[12 * best + tid] : ((tid % 5 == 0) ? 1.0 : 0.0);
    if (tid == 0) sBase = 0;
    __syncthreads();
    for (int base = 0; base < N; base += kSelBlock) {
        const int i = base + tid;
        bool in = false;
        if (i < N && best >= 0) {
            const double Zc = fma(sP[8], objp[3 * i], fma(sP[9], objp[3 * i + 1], fma(sP[10], objp[3 * i + 2], sP[11])));
            const double e2 = reproj_sqerr(sP, sI, objp[3 * i], objp[3 * i + 1], objp[3 * i + 2], imgp[2 * i], imgp[2 * i + 1]);
            in = Zc > 0.0 && e2 <= thr2;
        }
        if (i < N && mask) mask[i] = in ? 1 : 0;
        const unsigned long long bal = __ballot(in);
        const int lane = tid & 63, wave = tid >> 6;
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) sWave[wave] = __popcll(bal);
        __syncthreads();
        int off = sBase;
        for (int w = 0; w < wave; ++w) off += sWave[w];
        if (in) inlier_idx[off + before] = i;
        __syncthreads();
        if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
        __syncthreads();
    }
    if (tid < 12) best_pose[tid] = sP[tid];
    if (tid == 0) { out_sel[0] = best; out_sel[1] = sBase; ptr2[0] = 0; ptr2[1] = sBase; }
}

int check_points(const double *objp, const double *imgp, int64_t N, const double *intr)
{
    MQS_ARG_CHECK(N >= 0 && N <= 0x7fffffff, "0 <= N < 2^31");
    MQS_ARG_CHECK(intr != nullptr, "intr must not be null");
    MQS_ARG_CHECK(N == 0 || (objp && imgp), "objp, imgp must not be null");
    return MQS_OK;
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
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    char *w = static_cast<char *>(workspace);
    double *poses = reinterpret_cast<double *>(w); w += up((int64_t)B * 96);
    int32_t *counts = reinterpret_cast<int32_t *>(w); w += up((int64_t)B * 4);
    int32_t *ptr2 = reinterpret_cast<int32_t *>(w); w += up(8);
    int32_t *inl = reinterpret_cast<int32_t *>(w);
    const double thr2 = reproj_error * reproj_error;
    hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3(B), dim3(kWave), 0, stream, objp, imgp, (int)N, intr, samples,
                       sample_size, sample_iters, thr2, poses, counts);
    // best hypothesis -> pose_out (used as the start of the final refinement), inliers -> inl / mask
    hipLaunchKernelGGL(pnp_select_kernel, dim3(1), dim3(kSelBlock), 0, stream, objp, imgp, (int)N, intr, poses, counts, B,
                       thr2, pose_out, sel_out, ptr2, inl, mask);
    // OpenCV 2.4 solvePnPRansac ends with solvePnP on the inliers, started from the best model
    hipLaunchKernelGGL(pnp_refine_kernel, dim3(1), dim3(kWave), 0, stream, objp, imgp, (int)N, inl, ptr2, intr, pose_out, 1,
                       max_iter, eps, pose_out, info);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // extern "C"

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
