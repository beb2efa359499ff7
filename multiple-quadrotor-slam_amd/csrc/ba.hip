// Bundle-adjustment kernels for gfx950 (MI355X): projection-factor linearisation with on-chip
// landmark elimination (Schur complement), landmark back-substitution, cost evaluation, and the
// small reduced-camera-system solve + pose retraction.
//
// Replaces the GTSAM work behind Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:289-298
// (GenericProjectionFactor graph) and :323-324 (LevenbergMarquardtOptimizer::optimize):
// linearise all factors, build and solve the normal equations.  Arithmetic: ba_math.h.
//
// Mapping to the machine
//   * one thread per landmark, 256-thread workgroups, grid-stride over landmark batches with a
//     grid sized to the resident capacity (persistent waves keep their partial sums in registers);
//   * the C camera blocks (pose, calibration, 1/sigma: 24 doubles each) are staged once per
//     workgroup in LDS; landmarks (24 B) arrive as coalesced 16-byte pieces through an LDS
//     transpose; observations are camera-major, 16 contiguous bytes per lane;
//   * a landmark's contribution to the (6C)^2 reduced camera matrix is never stored: it is
//     produced slot by slot (ba_math.h Layout) into a 32-entry register window, and every full
//     window is summed over the wavefront with a TRANSPOSED shuffle reduction -- 32 values cost
//     32 exchange+add steps (v_permlane32_swap / v_permlane16_swap for the 32- and 16-lane
//     strides, DPP/ds_swizzle shuffles below) instead of 32 x 6 for independent butterflies --
//     leaving lane l with the wave total of window entry l >> 1, accumulated in one register per
//     window across all the batches the wave processes;
//   * waves -> workgroup through LDS, workgroups -> device through a [groups][slots] partial
//     array and a one-workgroup finalize kernel that also symmetrises S.  No atomics: the result
//     is bitwise reproducible run to run.
// Algorithmic HBM traffic per landmark per linearisation: 24 + 16*C bytes (+1*C with a mask,
// +8 with priors); the kernel is bound by fp64 VALU issue, not by HBM.
#include "mqs_common.h"
#include "ba_math.h"

namespace {

using namespace mqs::ba;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ void swap32(double &a, double &b)
{
    // v_permlane32_swap: a[lanes 32..63] <-> b[lanes 0..31]
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto rlo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto rhi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(rhi[0], rlo[0]);
    b = __hiloint2double(rhi[1], rlo[1]);
}

__device__ __forceinline__ void swap16(double &a, double &b)
{
    // v_permlane16_swap: odd 16-lane rows of a <-> even rows of b
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto rlo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto rhi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(rhi[0], rlo[0]);
    b = __hiloint2double(rhi[1], rlo[1]);
}

// Lane exchange v[lane ^ STRIDE] for STRIDE in {8, 4, 2, 1} with DPP moves (VALU latency; no LDS
// crossbar round trip as with ds_bpermute, which sat on the reduction's critical path).
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK_MASK, false);
}

template <int STRIDE>
__device__ __forceinline__ double xor_lane(double v)
{
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    if (STRIDE == 8) {                       // row_ror:8 within each row of 16 lanes
        lo = dpp_mov<0x128, 0xf>(lo, lo);
        hi = dpp_mov<0x128, 0xf>(hi, hi);
    } else if (STRIDE == 4) {                // banks {0,2} take lane+4 (row_shl:4), banks {1,3} lane-4 (row_shr:4)
        const unsigned l0 = lo, h0 = hi;
        lo = dpp_mov<0x104, 0x5>(l0, l0);
        lo = dpp_mov<0x114, 0xa>(lo, l0);
        hi = dpp_mov<0x104, 0x5>(h0, h0);
        hi = dpp_mov<0x114, 0xa>(hi, h0);
    } else if (STRIDE == 2) {                // quad_perm [2,3,0,1]
        lo = dpp_mov<0x4e, 0xf>(lo, lo);
        hi = dpp_mov<0x4e, 0xf>(hi, hi);
    } else {                                 // quad_perm [1,0,3,2]
        lo = dpp_mov<0xb1, 0xf>(lo, lo);
        hi = dpp_mov<0xb1, 0xf>(hi, hi);
    }
    return __hiloint2double(hi, lo);
}

// Transposed wavefront reduction of 32 values per lane: on return lane l holds the sum over
// all 64 lanes of v[l >> 1].  v is destroyed.
__device__ __forceinline__ double wave_reduce32(double (&v)[32], int lane)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap32(v[i], v[i + 16]); v[i] += v[i + 16]; }   // lane bit 5 selects i (+16)
#pragma unroll
    for (int i = 0; i < 8; ++i) { swap16(v[i], v[i + 8]); v[i] += v[i + 8]; }      // lane bit 4 selects i (+8)
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double send = b3 ? v[i] : v[i + 4], keep = b3 ? v[i + 4] : v[i];
        v[i] = keep + xor_lane<8>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double send = b2 ? v[i] : v[i + 2], keep = b2 ? v[i + 2] : v[i];
        v[i] = keep + xor_lane<4>(send);
    }
    {
        const double send = b1 ? v[0] : v[1], keep = b1 ? v[1] : v[0];
        v[0] = keep + xor_lane<2>(send);
    }
    return v[0] + xor_lane<1>(v[0]);
}

// Per-wave emitter: a 32-entry register window; every completed window is reduced over the wave
// and added to this lane's running total, kept in LDS (one double per window per thread) so that
// the window index may be anything and no accumulator registers are pinned.
struct WaveEmitter {
    double buf[32];
    double *acc;     // LDS, already offset by the thread index; stride kBlock per window
    int lane;
    __device__ __forceinline__ void put(int slot, double v)
    {
        buf[slot & 31] = v;
        if ((slot & 31) == 31) flush(slot >> 5);
    }
    __device__ __forceinline__ void flush(int window)
    {
#if defined(MQS_BA_ABLATE_REDUCE)      // timing experiment only: per-lane sum instead of the wave reduction
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += buf[i];
        acc[window * kBlock] += t;
#else
        acc[window * kBlock] += wave_reduce32(buf, lane);
#endif
    }
};

// Measurement access for ba_math.h: re-reads from global memory (L2 hits after the first touch).
struct DevObs {
    const double2 *o2;
    const uint8_t *mask;
    int64_t i, N;
    bool live;
    __device__ __forceinline__ void get(int c, double &u, double &v, bool &seen) const
    {
        u = 0.0; v = 0.0; seen = false;
        if (live) {
            const double2 t = o2[(int64_t)c * N + i];
            seen = mask ? (mask[(int64_t)c * N + i] != 0) : true;
            u = seen ? t.x : 0.0; v = seen ? t.y : 0.0;       // a masked slot may hold NaN
        }
    }
};

// F = E^T E and f = E^T e of every camera between the two passes of the lineariser: this thread's LDS column.
struct LdsFactorStash {
    static constexpr bool kEnabled = true;
    double *s;          // entry (c, k) at s[(5 * c + k) * kBlock]
    __device__ __forceinline__ void put(int c, double F00, double F01, double F11, double f0, double f1) const
    {
        double *p = s + 5 * c * kBlock;
        p[0] = F00; p[kBlock] = F01; p[2 * kBlock] = F11; p[3 * kBlock] = f0; p[4 * kBlock] = f1;
    }
    __device__ __forceinline__ void get(int c, double &F00, double &F01, double &F11, double &f0, double &f1) const
    {
        const double *p = s + 5 * c * kBlock;
        F00 = p[0]; F01 = p[kBlock]; F11 = p[2 * kBlock]; f0 = p[3 * kBlock]; f1 = p[4 * kBlock];
    }
};

// Cooperative load of this batch's landmarks: coalesced 16-byte pieces -> LDS -> 3 doubles per thread.
__device__ __forceinline__ void load_points(const double *__restrict__ points, int64_t base, int64_t N, double *sX,
                                            int tid, double &px, double &py, double &pz)
{
    const int64_t rem = N - base;
    const int npts = rem < kBlock ? (int)rem : kBlock;
    const int ndbl = npts * 3;
    const double2 *src = reinterpret_cast<const double2 *>(points + base * 3);   // 16-B aligned: base % 256 == 0
    double2 *dst = reinterpret_cast<double2 *>(sX);
    const int npair = ndbl >> 1;
    for (int p = tid; p < npair; p += kBlock) dst[p] = src[p];
    if ((ndbl & 1) && tid == 0) sX[ndbl - 1] = points[base * 3 + ndbl - 1];
    __syncthreads();
    const bool live = tid < npts;
    px = live ? sX[tid * 3 + 0] : 0.0;
    py = live ? sX[tid * 3 + 1] : 0.0;
    pz = live ? sX[tid * 3 + 2] : 0.0;
}

template <int C>
__device__ __forceinline__ void stage_cams(const double *poses, const double *calib, const double *sigma, double *sCam, int tid)
{
    if (tid < C) stage_camera(sCam + kCamStride * tid, poses + 12 * tid, calib + 9 * tid, sigma[tid]);
    __syncthreads();
}

__device__ __forceinline__ void load_prior(const double *__restrict__ prior_w, const double *__restrict__ prior_xyz,
                                           int64_t i, bool live, double px, double py, double pz, double &pw,
                                           double &dx, double &dy, double &dz)
{
    pw = 0.0; dx = dy = dz = 0.0;
    if (live && prior_w) {
        pw = prior_w[i];
        if (pw > 0.0) {
            dx = px - prior_xyz[3 * i + 0];
            dy = py - prior_xyz[3 * i + 1];
            dz = pz - prior_xyz[3 * i + 2];
        } else {
            pw = 0.0;
        }
    }
}

#ifndef MQS_BA_FACTOR_STASH
#define MQS_BA_FACTOR_STASH 1
#endif
template <int C>
__global__ __launch_bounds__(kBlock, 2) void ba_linearize_kernel(
    const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double lambda,
    double *__restrict__ partials)
{
    using L = Layout<C>;
    constexpr int NCH = L::kChunks;
    __shared__ double sCam[C * kCamStride];
    __shared__ double sX[kBlock * 3];
    __shared__ double sAcc[NCH * kBlock];
    constexpr bool kStash = MQS_BA_FACTOR_STASH && C <= 4;           // 5 C doubles per thread: fits beside sAcc up to 4 cameras
    __shared__ double sF[kStash ? 5 * C * kBlock : 1];

    const int tid = threadIdx.x;
    stage_cams<C>(poses, calib, sigma, sCam, tid);

    WaveEmitter em;
    em.lane = tid & 63;
    em.acc = sAcc + tid;
#pragma unroll
    for (int k = 0; k < NCH; ++k) sAcc[k * kBlock + tid] = 0.0;

    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t i = base + tid;
        const bool live = i < N;
        double px, py, pz;
        load_points(points, base, N, sX, tid, px, py, pz);
        const DevObs ob = {reinterpret_cast<const double2 *>(obs), mask, i, N, live};
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
        if (kStash) {
            const LdsFactorStash stash = {sF + tid};
            landmark_contribution<C, DevObs, WaveEmitter, LdsFactorStash>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, live, em, stash);
        } else {
            landmark_contribution<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, live, em);
        }
        __syncthreads();                                    // sX is reused by the next batch
    }

    // lanes -> workgroup: window k, entry j lives in lanes 2j (and 2j+1) of every wave
    __syncthreads();
    for (int s = tid; s < NCH * 32; s += kBlock) {
        const int k = s >> 5, j = s & 31;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += sAcc[k * kBlock + w * 64 + 2 * j];
        partials[(int64_t)blockIdx.x * (NCH * 32) + s] = t;
    }
}

// Sums the per-workgroup partials in a fixed order (reproducible) and scatters the slots into
// out = [S | g | cost | count], mirroring S.  One workgroup of 1024 threads per 64 slots: wave w
// adds rows w, w+16, ... (64 consecutive slots per row read = one 512-byte coalesced load, eight
// in flight), then the 16 waves combine through LDS.
constexpr int kFinThreads = 1024;

template <int C>
__global__ __launch_bounds__(kFinThreads) void ba_finalize_kernel(const double *__restrict__ partials, int nblocks,
                                                                  double *__restrict__ out)
{
    using L = Layout<C>;
    constexpr int NCH = L::kChunks;
    constexpr int kRow = NCH * 32;
    __shared__ double sSum[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    double t = 0.0;
    if (s < kRow) {
        int b = wave;
        for (; b + 16 * 7 < nblocks; b += 16 * 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partials[(int64_t)(b + 16 * k) * kRow + s];
#pragma unroll
            for (int k = 0; k < 8; ++k) t += v[k];
        }
        for (; b < nblocks; b += 16) t += partials[(int64_t)b * kRow + s];
    }
    sSum[wave][lane] = t;
    __syncthreads();
    if (wave == 0 && s < L::kSlots) {
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) r += sSum[w][lane];
        int o1, o2;
        slot_to_out<C>(s, o1, o2);
        if (o1 >= 0) out[o1] = r;
        if (o2 >= 0) out[o2] = r;
    }
}

template <int C>
__global__ __launch_bounds__(kBlock, 4) void ba_backsub_kernel(
    const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double lambda,
    const double *__restrict__ dpose, double *__restrict__ points_out)
{
    __shared__ double sCam[C * kCamStride];
    __shared__ double sX[kBlock * 3];
    __shared__ double sD[6 * C];
    const int tid = threadIdx.x;
    if (tid < 6 * C) sD[tid] = dpose[tid];
    stage_cams<C>(poses, calib, sigma, sCam, tid);

    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t i = base + tid;
        const bool live = i < N;
        double px, py, pz;
        load_points(points, base, N, sX, tid, px, py, pz);
        const DevObs ob = {reinterpret_cast<const double2 *>(obs), mask, i, N, live};
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
        const mqs::Vec3 dp = landmark_backsub<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, sD);
        __syncthreads();                                    // every thread has read its point from sX
        sX[tid * 3 + 0] = px + dp.x;
        sX[tid * 3 + 1] = py + dp.y;
        sX[tid * 3 + 2] = pz + dp.z;
        __syncthreads();
        const int64_t rem = N - base;
        const int npts = rem < kBlock ? (int)rem : kBlock;
        const int ndbl = npts * 3;
        double2 *dst = reinterpret_cast<double2 *>(points_out + base * 3);
        const double2 *src = reinterpret_cast<const double2 *>(sX);
        for (int p = tid; p < (ndbl >> 1); p += kBlock) dst[p] = src[p];
        if ((ndbl & 1) && tid == 0) points_out[base * 3 + ndbl - 1] = sX[ndbl - 1];
        __syncthreads();
    }
}

template <int C>
__global__ __launch_bounds__(kBlock) void ba_cost_kernel(
    const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double *__restrict__ partials)
{
    __shared__ double sCam[C * kCamStride];
    __shared__ double sX[kBlock * 3];
    __shared__ double sRed[2 * kWaves];
    const int tid = threadIdx.x;
    stage_cams<C>(poses, calib, sigma, sCam, tid);
    double cost = 0.0, count = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t i = base + tid;
        const bool live = i < N;
        double px, py, pz;
        load_points(points, base, N, sX, tid, px, py, pz);
        const DevObs ob = {reinterpret_cast<const double2 *>(obs), mask, i, N, live};
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
        double c1, n1;
        landmark_cost<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, c1, n1);
        if (live) { cost += c1; count += n1; }
        __syncthreads();
    }
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) { cost += __shfl_xor(cost, h); count += __shfl_xor(count, h); }
    if ((tid & 63) == 0) { sRed[2 * (tid >> 6)] = cost; sRed[2 * (tid >> 6) + 1] = count; }
    __syncthreads();
    if (tid == 0) {
        double c = 0, n = 0;
        for (int w = 0; w < kWaves; ++w) { c += sRed[2 * w]; n += sRed[2 * w + 1]; }
        partials[2 * blockIdx.x] = c;
        partials[2 * blockIdx.x + 1] = n;
    }
}

__global__ __launch_bounds__(kBlock) void ba_cost_finalize_kernel(const double *__restrict__ partials, int nblocks,
                                                                  double *__restrict__ out)
{
    __shared__ double sC[kBlock], sN[kBlock];
    const int tid = threadIdx.x;
    double c = 0.0, n = 0.0;
    for (int b = tid; b < nblocks; b += kBlock) { c += partials[2 * b]; n += partials[2 * b + 1]; }
    sC[tid] = c; sN[tid] = n;
    __syncthreads();
    for (int h = kBlock / 2; h >= 1; h >>= 1) {
        if (tid < h) { sC[tid] += sC[tid + h]; sN[tid] += sN[tid + h]; }
        __syncthreads();
    }
    if (tid == 0) { out[0] = sC[0]; out[1] = sN[0]; }
}

// ---------------------------------------------------------------------------------------
// Reduced camera system: add pose priors (bundle_adjust.cpp:273) and damping, Cholesky solve,
// retract the poses.  n = 6C <= 48: one wavefront, matrix in LDS.
// ---------------------------------------------------------------------------------------

__device__ void so3_log_dev(const double *R /*3x3 row-major*/, double w[3])
{
    const double tr = R[0] + R[4] + R[8];
    double c = 0.5 * (tr - 1.0);
    c = fmin(1.0, fmax(-1.0, c));
    const double th = acos(c);
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    const double k = (th < 1e-10) ? 0.5 : th / (2.0 * sin(th));
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
}

__device__ void so3_exp_dev(const double w[3], double E[9])
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double a, b;
    if (th < 1e-10) { a = 1.0; b = 0.5; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    for (int i = 0; i < 9; ++i) E[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

__device__ __forceinline__ double read_lane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// One wavefront; lane r keeps row r of the (6C x 6C) system in registers.  Right-looking Cholesky
// with v_readlane broadcasts (all register indices static), forward substitution the same way,
// backward substitution through an LDS copy of L (needs columns).
template <int C>
__global__ __launch_bounds__(64) void ba_solve_kernel(const double *__restrict__ lin, const double *__restrict__ poses,
                                                      const double *__restrict__ prior_poses,
                                                      const double *__restrict__ prior_sigmas,
                                                      const uint8_t *__restrict__ prior_mask, double lambda,
                                                      double *__restrict__ dpose, double *__restrict__ poses_out,
                                                      double *__restrict__ info)
{
    constexpr int n = 6 * C, ld = n + 1;
    __shared__ double sL[n * ld];
    __shared__ double sE[n];          // pose-prior: weighted residual added to g
    __shared__ double sW[n];          // pose-prior: weight added to the diagonal
    __shared__ double sInfo[2];
    const int lane = threadIdx.x;
    if (lane < n) { sE[lane] = 0.0; sW[lane] = 0.0; }
    if (lane == 0) { sInfo[0] = 0.0; sInfo[1] = 0.0; }
    __syncthreads();
    // pose priors: e = (Log(R0^T R), R0^T (t - t0)) / sigma, J ~ I  (bundle_adjust.cpp:273)
    if (prior_mask && lane < C && prior_mask[lane]) {
        const double *T0 = prior_poses + 12 * lane, *T = poses + 12 * lane, *sg = prior_sigmas + 6 * lane;
        double Rr[9], w[3], e[6];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rr[3 * i + j] = T0[i] * T[j] + T0[3 + i] * T[3 + j] + T0[6 + i] * T[6 + j];
        so3_log_dev(Rr, w);
        const double dt[3] = {T[9] - T0[9], T[10] - T0[10], T[11] - T0[11]};
        for (int i = 0; i < 3; ++i) {
            e[i] = w[i];
            e[3 + i] = T0[i] * dt[0] + T0[3 + i] * dt[1] + T0[6 + i] * dt[2];
        }
        double cst = 0.0;
        for (int i = 0; i < 6; ++i) {
            const double wi = 1.0 / (sg[i] * sg[i]);
            sW[6 * lane + i] = wi;
            sE[6 * lane + i] = wi * e[i];
            cst += 0.5 * wi * e[i] * e[i];
        }
        atomicAdd(&sInfo[0], cst);
    }
    __syncthreads();
    const bool rowlane = lane < n;
    const int r = rowlane ? lane : 0;
    double row[n];
#pragma unroll
    for (int j = 0; j < n; ++j) row[j] = lin[r * n + j];
    double b = lin[n * n + r] - sE[r];
    // diagonal: prior weight, then damping
#pragma unroll
    for (int j = 0; j < n; ++j)
        if (j == r) row[j] = (row[j] + sW[r]) * (1.0 + lambda);
    if (!rowlane) {
        b = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) row[j] = 0.0;
    }
    bool bad = false;
    double dinv = 1.0;
    // Column k of L goes through LDS once per step and comes back as broadcast reads (one address for the whole wave,
    // two entries per ds_read_b128): the v_readlane pair per (k, j) of the first version put ~550 scalar round trips on
    // the serial chain.  One wave: the LDS pipe keeps its accesses in order, no barrier needed (mqs_wave_lds_sync).
    __shared__ __attribute__((aligned(16))) double sCol[64];
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double akk = read_lane(row[k], k);
        bad = bad || !(akk > 0.0);
        const double inv = mqs::rsqrt_d(akk > 0.0 ? akk : 1.0);
        const double lik = row[k] * inv;             // L[lane][k] for lane >= k (lane k: the pivot)
        row[k] = lik;
        if (lane == k) dinv = inv;
        if (k + 1 < n) {
            if (n <= 36) {
                sCol[lane] = lik;
                mqs_wave_lds_sync();
#pragma unroll
                for (int j = k + 1; j < n; ++j) row[j] = fma(-lik, sCol[j], row[j]);   // only entries j <= lane are used later
                mqs_wave_lds_sync();
            } else {
                // 7 and 8 cameras: with 42 / 48 row registers the scheduler runs ahead on the pivot chain, parks every
                // step's loaded column and spills -- the v_readlane form has nothing to park
#pragma unroll
                for (int j = k + 1; j < n; ++j) row[j] = fma(-lik, read_lane(lik, j), row[j]);
            }
        }
    }
    // forward substitution L y = b
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double t = b * dinv;
        const double yk = read_lane(t, k);
        b = (lane == k) ? t : ((lane > k) ? fma(-row[k], yk, b) : b);
    }
    // backward substitution L^T x = y through LDS (column access)
    if (rowlane) {
#pragma unroll
        for (int j = 0; j < n; ++j) sL[lane * ld + j] = row[j];
    }
    __syncthreads();
#pragma unroll 1
    for (int k = n - 1; k >= 0; --k) {
        const double t = b * dinv;
        const double xk = read_lane(t, k);
        const double lki = (lane < k) ? sL[k * ld + lane] : 0.0;
        b = (lane == k) ? t : fma(-lki, xk, b);
    }
    __shared__ double sX[n];
    if (rowlane) { dpose[lane] = b; sX[lane] = b; }
    __syncthreads();
    if (poses_out && lane < C) {
        const double *T = poses + 12 * lane;
        double E[9], w[3] = {sX[6 * lane], sX[6 * lane + 1], sX[6 * lane + 2]};
        so3_exp_dev(w, E);
        double *O = poses_out + 12 * lane;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) O[3 * i + j] = T[3 * i] * E[j] + T[3 * i + 1] * E[3 + j] + T[3 * i + 2] * E[6 + j];
        for (int i = 0; i < 3; ++i)
            O[9 + i] = T[9 + i] + T[3 * i] * sX[6 * lane + 3] + T[3 * i + 1] * sX[6 * lane + 4] + T[3 * i + 2] * sX[6 * lane + 5];
    }
    if (info && lane == 0) { info[0] = sInfo[0]; info[1] = bad ? 1.0 : 0.0; }
}

// Grid: persistent workgroups, 2 per CU at <= 256 VGPRs (each keeps its partial sums in registers).
int ba_grid(int64_t N)
{
    int64_t g = (N + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 512) g = 512;
    return (int)g;
}

template <int C>
int64_t ws_doubles() { return (int64_t)512 * Layout<C>::kChunks * 32; }

int64_t ws_doubles_rt(int C)
{
    switch (C) {
#define MQS_CASE(c) case c: return ws_doubles<c>();
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    return 0;
}

int check_common(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                 const double *obs, int64_t N)
{
    MQS_ARG_CHECK(C >= 1 && C <= MQS_MAX_CAMS, "1 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    MQS_ARG_CHECK(poses && calib && sigma, "poses, calib, sigma must not be null");
    MQS_ARG_CHECK(N == 0 || (points && obs), "points, obs must not be null");
    MQS_ARG_CHECK(mqs_aligned16(points) && mqs_aligned16(obs), "device pointers must be 16-byte aligned");
    return MQS_OK;
}

}  // namespace

extern "C" {

int64_t mqs_ba_workspace_bytes(int C, int64_t N)
{
    (void)N;
    if (C < 1 || C > MQS_MAX_CAMS) return 0;
    return ws_doubles_rt(C) * 8;
}

int mqs_ba_linearize_dev(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                         const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                         int64_t N, double lambda, double *out, void *workspace, int64_t workspace_bytes, void *stream_)
{
    int rc = check_common(poses, calib, sigma, C, points, obs, N);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(out != nullptr && workspace != nullptr, "out and workspace must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_ba_workspace_bytes(C, N), "workspace too small (mqs_ba_workspace_bytes)");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int grid = ba_grid(N);
    double *partials = static_cast<double *>(workspace);
    switch (C) {
#define MQS_CASE(c)                                                                                        \
    case c:                                                                                                \
        hipLaunchKernelGGL((ba_linearize_kernel<c>), dim3(grid), dim3(kBlock), 0, stream, poses, calib, sigma, \
                           points, obs, mask, prior_w, prior_xyz, N, lambda, partials);                    \
        hipLaunchKernelGGL((ba_finalize_kernel<c>), dim3((Layout<c>::kChunks * 32 + 63) / 64), dim3(kFinThreads), 0, stream, partials, grid, out);   \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_ba_backsub_dev(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                       const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                       int64_t N, double lambda, const double *dpose, double *points_out, void *stream_)
{
    int rc = check_common(poses, calib, sigma, C, points, obs, N);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(dpose && (N == 0 || points_out), "dpose, points_out must not be null");
    MQS_ARG_CHECK(mqs_aligned16(points_out), "points_out must be 16-byte aligned");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    if (N == 0) return MQS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned grid = mqs_stream_grid(N, kBlock);
    switch (C) {
#define MQS_CASE(c)                                                                                       \
    case c:                                                                                               \
        hipLaunchKernelGGL((ba_backsub_kernel<c>), dim3(grid), dim3(kBlock), 0, stream, poses, calib, sigma, \
                           points, obs, mask, prior_w, prior_xyz, N, lambda, dpose, points_out);          \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_ba_cost_dev(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                    const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                    double *out, void *workspace, int64_t workspace_bytes, void *stream_)
{
    int rc = check_common(poses, calib, sigma, C, points, obs, N);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(out != nullptr && workspace != nullptr, "out and workspace must not be null");
    MQS_ARG_CHECK(workspace_bytes >= 2 * 4096 * 8, "workspace too small (64 KiB)");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int64_t g64 = (N + kBlock - 1) / kBlock;          // latency-bound streaming kernel: fill the machine
    const int grid = (int)(g64 < 1 ? 1 : (g64 > 4096 ? 4096 : g64));
    double *partials = static_cast<double *>(workspace);
    switch (C) {
#define MQS_CASE(c)                                                                                    \
    case c:                                                                                            \
        hipLaunchKernelGGL((ba_cost_kernel<c>), dim3(grid), dim3(kBlock), 0, stream, poses, calib, sigma, \
                           points, obs, mask, prior_w, prior_xyz, N, partials);                        \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    hipLaunchKernelGGL(ba_cost_finalize_kernel, dim3(1), dim3(kBlock), 0, stream, partials, grid, out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_ba_solve_dev(const double *lin, int C, const double *poses, const double *prior_poses,
                     const double *prior_sigmas, const uint8_t *prior_mask, double lambda, double *dpose,
                     double *poses_out, double *info, void *stream_)
{
    MQS_ARG_CHECK(C >= 1 && C <= MQS_MAX_CAMS, "1 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(lin && poses && dpose, "lin, poses, dpose must not be null");
    MQS_ARG_CHECK(!prior_mask || (prior_poses && prior_sigmas), "prior_poses/prior_sigmas required with prior_mask");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    switch (C) {
#define MQS_CASE(c)                                                                                     \
    case c:                                                                                             \
        hipLaunchKernelGGL((ba_solve_kernel<c>), dim3(1), dim3(64), 0, stream, lin, poses, prior_poses, \
                           prior_sigmas, prior_mask, lambda, dpose, poses_out, info);                   \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Host-pointer wrappers: stage everything in the ctx scratch, run, copy the result back.
// ---------------------------------------------------------------------------------------
namespace {

struct BaStage {
    double *poses, *calib, *sigma, *points, *obs, *prior_w, *prior_xyz, *dpose, *out, *points_out;
    uint8_t *mask;
    void *ws;
    int64_t ws_bytes;
};

int ba_stage(mqs_ctx *ctx, const double *poses, const double *calib, const double *sigma, int C, const double *points,
             const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
             const double *dpose, BaStage &st)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(C >= 1 && C <= MQS_MAX_CAMS, "1 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    MQS_ARG_CHECK(poses && calib && sigma && (N == 0 || (points && obs)), "inputs must not be null");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const int n6 = 6 * C;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
    const size_t o_poses = take(C * 96), o_calib = take(C * 72), o_sigma = take(C * 8);
    const size_t o_points = take((size_t)N * 24), o_obs = take((size_t)C * N * 16), o_mask = take((size_t)C * N);
    const size_t o_pw = take((size_t)N * 8), o_px = take((size_t)N * 24), o_dpose = take(n6 * 8);
    const size_t o_out = take(((size_t)n6 * n6 + n6 + 2) * 8), o_pout = take((size_t)N * 24);
    const int64_t wsb = mqs_ba_workspace_bytes(C, N);
    const size_t o_ws = take((size_t)wsb);
    int rc = mqs_ctx_reserve(ctx, off);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    st.poses = (double *)(d + o_poses); st.calib = (double *)(d + o_calib); st.sigma = (double *)(d + o_sigma);
    st.points = (double *)(d + o_points); st.obs = (double *)(d + o_obs);
    st.mask = mask ? (uint8_t *)(d + o_mask) : nullptr;
    st.prior_w = prior_w ? (double *)(d + o_pw) : nullptr;
    st.prior_xyz = prior_w ? (double *)(d + o_px) : nullptr;
    st.dpose = (double *)(d + o_dpose); st.out = (double *)(d + o_out); st.points_out = (double *)(d + o_pout);
    st.ws = d + o_ws; st.ws_bytes = wsb;
    hipStream_t s = ctx->stream;
    MQS_HIP_CHECK(hipMemcpyAsync(st.poses, poses, C * 96, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(st.calib, calib, C * 72, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(st.sigma, sigma, C * 8, hipMemcpyHostToDevice, s));
    if (N > 0) {
        MQS_HIP_CHECK(hipMemcpyAsync(st.points, points, (size_t)N * 24, hipMemcpyHostToDevice, s));
        MQS_HIP_CHECK(hipMemcpyAsync(st.obs, obs, (size_t)C * N * 16, hipMemcpyHostToDevice, s));
        if (mask) MQS_HIP_CHECK(hipMemcpyAsync(st.mask, mask, (size_t)C * N, hipMemcpyHostToDevice, s));
        if (prior_w) {
            MQS_HIP_CHECK(hipMemcpyAsync(st.prior_w, prior_w, (size_t)N * 8, hipMemcpyHostToDevice, s));
            MQS_HIP_CHECK(hipMemcpyAsync(st.prior_xyz, prior_xyz, (size_t)N * 24, hipMemcpyHostToDevice, s));
        }
    }
    if (dpose) MQS_HIP_CHECK(hipMemcpyAsync(st.dpose, dpose, n6 * 8, hipMemcpyHostToDevice, s));
    return MQS_OK;
}

}  // namespace

extern "C" {

int mqs_ba_linearize(mqs_ctx *ctx, const double *poses, const double *calib, const double *sigma, int C,
                     const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                     const double *prior_xyz, int64_t N, double lambda, double *out)
{
    MQS_ARG_CHECK(out != nullptr, "out must not be null");
    BaStage st;
    int rc = ba_stage(ctx, poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, nullptr, st);
    if (rc != MQS_OK) return rc;
    rc = mqs_ba_linearize_dev(st.poses, st.calib, st.sigma, C, st.points, st.obs, st.mask, st.prior_w, st.prior_xyz, N,
                              lambda, st.out, st.ws, st.ws_bytes, ctx->stream);
    if (rc != MQS_OK) return rc;
    const int n6 = 6 * C;
    MQS_HIP_CHECK(hipMemcpyAsync(out, st.out, ((size_t)n6 * n6 + n6 + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

int mqs_ba_backsub(mqs_ctx *ctx, const double *poses, const double *calib, const double *sigma, int C,
                   const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                   const double *prior_xyz, int64_t N, double lambda, const double *dpose, double *points_out)
{
    MQS_ARG_CHECK(dpose != nullptr && (N == 0 || points_out != nullptr), "dpose, points_out must not be null");
    BaStage st;
    int rc = ba_stage(ctx, poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, dpose, st);
    if (rc != MQS_OK) return rc;
    rc = mqs_ba_backsub_dev(st.poses, st.calib, st.sigma, C, st.points, st.obs, st.mask, st.prior_w, st.prior_xyz, N,
                            lambda, st.dpose, st.points_out, ctx->stream);
    if (rc != MQS_OK) return rc;
    if (N > 0) MQS_HIP_CHECK(hipMemcpyAsync(points_out, st.points_out, (size_t)N * 24, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

}  // extern "C"
