// Workgroup-level pieces of the pose step shared by pnp.hip's kernels and the device-resident loop's decision kernel
// (slam_frame.hip): sums of an evaluation over four wavefronts, and the end of solvePnPRansac as a device function.
#pragma once
#include "mqs_common.h"
#include "pnp_math.h"
#include "wave_reduce.h"
#include "cam_math.h"

namespace mqs {
namespace pnpblk {

using namespace mqs::pnp;

constexpr int kBlkWave = 64;
// ---- one wavefront per problem ----
__device__ __forceinline__ double wave_sum(double v) { return mqs::wave::sum1(v); }     // the butterfly's pairs, no ds_bpermute

// K sums over the wavefront, every lane ending with all of them: chunks of 32 through the transposed reduction of
// wave_reduce.h (32 exchange-and-add steps per chunk; lane 2 e ends with entry e), then through 32 doubles of LDS (`scr`, the
// wave's own) back to every lane as broadcast reads -- against K butterflies of six exchange steps each.  (Broadcasting with
// v_readlane pairs instead kept up to 102 scalar registers alive: 177 spilled SGPRs in the refinement kernel.)
template <int K>
__device__ __forceinline__ void wave_sum_all(double (&acc)[K], int lane, double *scr)
{
#pragma unroll
    for (int c0 = 0; c0 < K; c0 += 32) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = (c0 + k < K) ? acc[c0 + k] : 0.0;
        const double tot = mqs::wave::wave_reduce32(v, lane);
        mqs_wave_lds_sync();                              // earlier readers of scr are done
        if (!(lane & 1)) scr[lane >> 1] = tot;
        mqs_wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (c0 + k < K) acc[c0 + k] = scr[k];
    }
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
    double *scr;            // LDS, 32 doubles of the wave's own (wave_sum_all)
    __device__ __forceinline__ void operator()(const double *P, double *acc) const
    {
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
            const int i = pr.point(k);
            accumulate_point(P, intr, pr.objp[3 * i], pr.objp[3 * i + 1], pr.objp[3 * i + 2], pr.imgp[2 * i],
                             pr.imgp[2 * i + 1], acc);
        }
        // (on a six-point RANSAC hypothesis 28 butterflies were most of an evaluation)
        wave_sum_all(*reinterpret_cast<double (*)[kAcc]>(acc), lane, scr);
    }
};

// Direct linear transform start over the problem's points (>= 6); sA: 121 + 11 doubles of LDS per wave.
// Returns false (in every lane) for a degenerate configuration.
__device__ bool wave_dlt(const Problem &pr, const double *intr, int lane, double *sA, double *P)
{
    const int n = pr.end - pr.begin;
    double cx = 0.0, cy = 0.0, cz = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
        const int i = pr.point(k);
        cx += pr.objp[3 * i]; cy += pr.objp[3 * i + 1]; cz += pr.objp[3 * i + 2];
    }
    const double inv_n = mqs::rcp((double)n);
    const double c[3] = {wave_sum(cx) * inv_n, wave_sum(cy) * inv_n, wave_sum(cz) * inv_n};
    double dist = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
        const int i = pr.point(k);
        const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
        const double d2 = fma(dx, dx, fma(dy, dy, dz * dz));
        dist += d2 > 0.0 ? d2 * mqs::rsqrt_d(d2) : 0.0;
    }
    double cov[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
        const int i = pr.point(k);
        const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
        cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz; cov[3] += dy * dy; cov[4] += dy * dz; cov[5] += dz * dz;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = wave_sum(cov[k]);
    double sigma = wave_sum(dist) * inv_n;
    if (!(sigma > 0.0)) sigma = 1.0;
    const double is = mqs::rcp(sigma);
    double ew[3], E[9];
    sym3_eigen(cov, ew, E);
    if (ew[2] < 1e-3 * ew[1]) {
        // planar (OpenCV: W[2] / W[1] < 1e-3): start from the plane-to-image homography (needs >= 4 points)
        double hacc[kHomAcc];
#pragma unroll
        for (int k = 0; k < kHomAcc; ++k) hacc[k] = 0.0;
        for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
            const int i = pr.point(k);
            const double dx = pr.objp[3 * i] - c[0], dy = pr.objp[3 * i + 1] - c[1], dz = pr.objp[3 * i + 2] - c[2];
            double x, y;
            mqs::cam::undistort_pixel(intr, pr.imgp[2 * i], pr.imgp[2 * i + 1], x, y);
            hom_accumulate((E[0] * dx + E[1] * dy + E[2] * dz) * is, (E[3] * dx + E[4] * dy + E[5] * dz) * is, x, y, hacc);
        }
        wave_sum_all(hacc, lane, sA);
        // 8 x 8 solve: every lane for itself, the system in registers (chol_solve_fixed)
        double A8[64], hv[8];
        hom_assemble(hacc, A8, hv);
        const bool solved = chol_solve_fixed<8>(A8, hv);
        const bool posed = pose_from_homography(hv, E, c, sigma, P);
        return solved && posed;
    }
    double acc[kDltAcc];
#pragma unroll
    for (int k = 0; k < kDltAcc; ++k) acc[k] = 0.0;
    for (int k = pr.begin + lane; k < pr.end; k += kBlkWave) {
        const int i = pr.point(k);
        double x, y;
        mqs::cam::undistort_pixel(intr, pr.imgp[2 * i], pr.imgp[2 * i + 1], x, y);
        dlt_accumulate((pr.objp[3 * i] - c[0]) * is, (pr.objp[3 * i + 1] - c[1]) * is, (pr.objp[3 * i + 2] - c[2]) * is, x, y, acc);
    }
    wave_sum_all(acc, lane, sA);
    // 11 x 11 solve: every lane for itself, the system in registers (chol_solve_fixed)
    double A11[121], p[11];
    dlt_assemble(acc, A11, p);
    const bool solved = chol_solve_fixed<11>(A11, p);
    const bool posed = pose_from_dlt(p, c, sigma, P);
    return solved && posed;
}

// One RANSAC hypothesis by one wavefront: DLT of its sample `pr`, Levenberg-Marquardt on the sample, then the inlier count over the
// N correspondences (objp, imgp) -- global memory or the caller's LDS copy.  Returns the count (-1: degenerate sample); P: the pose.
__device__ __forceinline__ int hypothesis_wave(const Problem &pr, const double *objp, const double *imgp, int N, const double *sI,
                                               int sample_iters, double thr2, double *sA, int lane, double *P)
{
    int count = -1;
    if (wave_dlt(pr, sI, lane, sA, P)) {
        WaveEval ev = {pr, sI, lane, sA};
        lm_refine(ev, P, sample_iters, 1e-10);
        int c = 0;
        for (int i = lane; i < N; i += kBlkWave) {
            // behind the camera: never an inlier
            const double Zc = fma(P[8], objp[3 * i], fma(P[9], objp[3 * i + 1], fma(P[10], objp[3 * i + 2], P[11])));
            const double e2 = reproj_sqerr(P, sI, objp[3 * i], objp[3 * i + 1], objp[3 * i + 2], imgp[2 * i], imgp[2 * i + 1]);
            c += (Zc > 0.0 && e2 <= thr2) ? 1 : 0;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s);
        count = c;
    }
    return count;
}

// ---- one workgroup (four wavefronts) per problem ----
constexpr int kKfThreads = 256;                   // four waves share a frame: <= 300 correspondences are one or two sweeps
constexpr int kKfWaves = kKfThreads / kBlkWave;

// The 28 sums of an evaluation over the WORKGROUP (kKfThreads threads), in a fixed order; every thread ends with the same sums, so the
// Levenberg-Marquardt loop around it runs redundantly and in step in all of them.  Per wave the transposed reduction of wave_reduce.h
// (32 exchange-and-add steps, lane 2 e ends with entry e) instead of 28 butterflies of six steps each: with one or two points per
// thread the butterflies were three quarters of an evaluation's instructions.  red: LDS [kKfWaves][kAcc].
__device__ __forceinline__ void block_sum_acc(double *acc, double *red, int tid)
{
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = k < kAcc ? acc[k] : 0.0;
    const double tot = mqs::wave::wave_reduce32(v, tid & 63);
    __syncthreads();                              // the previous call's sums have been read by everyone
    if (!(tid & 1) && ((tid & 63) >> 1) < kAcc) red[(tid >> 6) * kAcc + ((tid & 63) >> 1)] = tot;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kAcc; ++k) {
        double t = red[k];
#pragma unroll
        for (int w = 1; w < kKfWaves; ++w) t += red[w * kAcc + k];
        acc[k] = t;
    }
}

// eval over the points idx[0..n) of (objp, imgp) -- idx null: 0..n -- by the whole workgroup
struct BlockEval {
    const double *objp, *imgp;
    const int32_t *idx;
    int n;
    const double *intr;
    double *red;
    int tid;
    __device__ __forceinline__ void operator()(const double *P, double *acc) const
    {
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        for (int k = tid; k < n; k += kKfThreads) {
            const int i = idx ? idx[k] : k;
            accumulate_point(P, intr, objp[3 * i], objp[3 * i + 1], objp[3 * i + 2], imgp[2 * i], imgp[2 * i + 1], acc);
        }
        block_sum_acc(acc, red, tid);
    }
};

// The end of solvePnPRansac in ONE launch of one workgroup (round 5; a 256-thread selection kernel and a one-wave refinement kernel
// before: 4.8 + 20.8 us and a launch gap per frame): picks the hypothesis with the most inliers (lowest index on ties), marks and
// compacts its inliers in index order -- into LDS when they fit (lds_ok: 40 bytes per correspondence), else as an index list in
// global memory -- and refines the pose on them (OpenCV 2.4: solvePnP on the inliers, started from the best model) with the sums
// of an evaluation taken by four wavefronts (block_sum_acc; the keyframe step's form).
// out_sel: [0] best hypothesis (-1: none valid), [1] inlier count.  info: as pnp_refine_kernel's.
constexpr int kSelBlock = kKfThreads;
// (a device function so that the device-resident loop's decision kernel can run it in its own launch: slam_frame.hip)
__device__ __forceinline__ void select_refine_block(const double *__restrict__ objp, const double *__restrict__ imgp,
                                                    int N, const int32_t *__restrict__ n_dev, const double *__restrict__ intr,
                                                    const double *__restrict__ poses, const int32_t *__restrict__ counts,
                                                    int B, double thr2, int max_iter, double eps, int lds_ok,
                                                    double *__restrict__ pose_out, int32_t *__restrict__ out_sel,
                                                    int32_t *__restrict__ inlier_idx, uint8_t *__restrict__ mask,
                                                    double *__restrict__ info, double *sel_lds /* dynamic LDS: 5 N doubles when lds_ok */)
{
    __shared__ double sI[9], sP[12];
    __shared__ double sRed[kKfWaves * kAcc];
    __shared__ int sBestC[kSelBlock], sBestH[kSelBlock], sWave[kSelBlock / 64], sBase;
    const int tid = threadIdx.x;
    if (n_dev) N = *n_dev;
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
    double *so = sel_lds, *si = sel_lds + 3 * (size_t)(lds_ok ? N : 0);
    for (int base = 0; base < N; base += kSelBlock) {
        const int i = base + tid;
        bool in = false;
        double X = 0, Y = 0, Z = 0, u = 0, v = 0;
        if (i < N && best >= 0) {
            X = objp[3 * i]; Y = objp[3 * i + 1]; Z = objp[3 * i + 2]; u = imgp[2 * i]; v = imgp[2 * i + 1];
            const double Zc = fma(sP[8], X, fma(sP[9], Y, fma(sP[10], Z, sP[11])));
            const double e2 = reproj_sqerr(sP, sI, X, Y, Z, u, v);
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
        if (in) {
            const int r = off + before;
            if (lds_ok) { so[3 * r] = X; so[3 * r + 1] = Y; so[3 * r + 2] = Z; si[2 * r] = u; si[2 * r + 1] = v; }
            else inlier_idx[r] = i;
        }
        __syncthreads();
        if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
        __syncthreads();
    }
    const int n_in = sBase;
    if (tid == 0) { out_sel[0] = best; out_sel[1] = n_in; }
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = sP[k];
    if (n_in < 3) {                                  // no problem (no valid model, or a rejected frame of the device-resident loop): the start pose
        if (tid < 12) pose_out[tid] = P[tid];
        if (info && tid < 4) info[tid] = tid == 3 ? 2.0 : 0.0;
        return;
    }
    if (!lds_ok) __threadfence_block();              // the index list written above is read below by other threads of the workgroup
    __syncthreads();
    BlockEval ev = {lds_ok ? so : objp, lds_ok ? si : imgp, lds_ok ? nullptr : inlier_idx, n_in, sI, sRed, tid};
    const LmResult r = lm_refine(ev, P, max_iter, eps);
    if (tid < 12) pose_out[tid] = P[tid];
    if (info && tid == 0) {
        info[0] = r.sqerr;
        info[1] = (double)r.iters;
        info[2] = (double)n_in;
        info[3] = r.converged ? 1.0 : 0.0;
    }
}


}  // namespace pnpblk
}  // namespace mqs
