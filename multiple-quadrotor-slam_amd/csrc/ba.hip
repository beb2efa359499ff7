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
#include "wave_reduce.h"
#include "peer_dev.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

using namespace mqs::ba;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

using mqs::wave::swap32;
using mqs::wave::swap16;
using mqs::wave::xor_lane;
using mqs::wave::wave_reduce32;

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

// Measurement access for the fused tail's back-substitution: the C measurements of the thread's landmark were brought into LDS
// by LDS-DMA loads that were all in flight at once (ba_tail_kernel), the rolled camera loop indexes them there; the mask bytes
// ride in one register.
struct StagedObs {
    const double2 *s;          // this thread's column: camera c at s[c * kBlock]
    unsigned mbits;            // byte c = the mask byte of camera c (C <= 4)
    bool live, masked;
    __device__ __forceinline__ void get(int c, double &u, double &v, bool &seen) const
    {
        const double2 t = s[c * kBlock];
        seen = live && (!masked || ((mbits >> (8 * c)) & 0xffu) != 0u);
        u = seen ? t.x : 0.0; v = seen ? t.y : 0.0;           // a masked slot may hold NaN
    }
};

// the same with one wave's 64 measurements per camera side by side (ba_iterate_kernel: camera c at s[c * 64])
struct StagedObs64 {
    const double2 *s;
    unsigned mbits;
    bool live, masked;
    __device__ __forceinline__ void get(int c, double &u, double &v, bool &seen) const
    {
        const double2 t = s[c * 64];
        seen = live && (!masked || ((mbits >> (8 * c)) & 0xffu) != 0u);
        u = seen ? t.x : 0.0; v = seen ? t.y : 0.0;           // a masked slot may hold NaN
    }
};

// the same for the second form of the tail (kTail2Block-thread workgroups: camera c at s[c * kTail2Block]).  The ring is read with ds_read
// instructions the compiler cannot see through: it knows the ring is written by LDS-DMA and, unable to tell the two halves of the
// ring apart, would put an s_waitcnt vmcnt(0) in front of every read -- draining the NEXT batch's loads, which are in flight into
// the other half on purpose.  That the half read here has landed is the loop's own explicit wait (ba_tail2_kernel).
// 12 waves per workgroup and CU = 3 per SIMD = 168 registers each: with 16 (128 registers) the solve wave and the batch loop, which
// holds the next batch's loads in registers while it computes, both spilled -- and a scratch reload is a vector memory operation
// whose wait also drains the prefetch it stands behind
constexpr int kTail2Block = 768;
typedef double wl_d2 __attribute__((ext_vector_type(2)));
struct StagedObs2 {
    unsigned s;                // LDS byte address of this thread's measurement of camera 0
    unsigned mbits;
    bool live, masked;
    __device__ __forceinline__ void get(int c, double &u, double &v, bool &seen) const
    {
        wl_d2 t;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(s + (unsigned)c * (unsigned)(kTail2Block * 16)) : "memory");
        seen = live && (!masked || ((mbits >> (8 * c)) & 0xffu) != 0u);
        u = seen ? t.x : 0.0; v = seen ? t.y : 0.0;
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

// ---------------------------------------------------------------------------------------------------------------------
// Wave-level lineariser (C <= 4): up to four landmarks per lane share every window reduction.
//
// What the one-landmark-per-lane kernel above spends (ISA histogram, profiles/r02): of ~3 960 vector issue slots per landmark
// ~1 940 are the cross-lane reduction of its 11 windows (v_permlane*_swap issues in two passes), i.e. half of a kernel that
// is bound by vector issue.  A window is a sum over landmarks, so nothing forces one reduction per landmark: here a lane
// walks L landmarks through each window, ADDING their contributions in the window registers (the add rides in the last FMA
// of every entry, so it is free), and the wave reduces once -- 1 940 / L slots per landmark.
//   * one workgroup of four waves per CU, one wave per SIMD (256 VGPRs + accumulator registers), so that the workgroup
//     owns the CU's LDS: what a landmark needs again in a later window lives in its lane's LDS column (six doubles per
//     (landmark, camera): F = E^T E and f = E^T e between the two passes, then Uh = F PR L^-T for the off-diagonal windows);
//     only x, y, Z of each factor stay in registers;
//   * windows follow the blocks: one 32-entry window per camera (21 + 6 entries; cost and count ride in camera 0's), and
//     per camera pair a 32-entry window plus a 4-entry one (36 = 32 + 4), so that no block straddles a window and no
//     window is open while another one fills;
//   * a wave owns a contiguous range of 64-landmark rows (15 or 16 at 1e6 landmarks) and walks it in chunks of four rows:
//     four reductions per wave instead of sixteen; the LDS holds three landmarks' columns, the fourth landmark's six doubles
//     per camera live in accumulation registers (three rows per chunk: 113-119 us, four: 107-114 us at 1e6 x 4);
//     nothing is shared between waves until the end, so there is no barrier inside the loop.
// Results: same formulas as landmark_contribution (ba_math.h), sums in a different, still fixed, order -- bitwise
// reproducible run to run.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MQS_WL_ONLY_NODIST
#define MQS_WL_ONLY_NODIST 0
#endif
#ifndef MQS_WL_MAXL                        // A/B builds: MQS_WL_MAXL=2 MQS_WL_LDSL=1 MQS_WL_OCC=2 = two waves per SIMD, two landmarks per lane
#define MQS_WL_MAXL 4
#define MQS_WL_LDSL 3
#define MQS_WL_OCC 1
#endif
constexpr int kWaveLinMaxL = MQS_WL_MAXL;  // landmarks per lane and chunk
constexpr int kWaveLinLdsL = MQS_WL_LDSL;  // of which this many park their six doubles per camera in LDS; the last one's live in AGPRs
constexpr int kWaveLinOcc = MQS_WL_OCC;    // workgroups per CU (waves per SIMD)
static_assert(kWaveLinMaxL <= kWaveLinLdsL + 1, "one landmark's stash fits the accumulation registers");

// sum over the 64 lanes of 4 values per lane: lane l ends with the total of v[l >> 4]
__device__ __forceinline__ double wave_reduce4(double (&v)[4], int lane)
{
    (void)lane;
#pragma unroll
    for (int i = 0; i < 2; ++i) { swap32(v[i], v[i + 2]); v[i] += v[i + 2]; }
    swap16(v[0], v[1]);
    v[0] += v[1];
    double t = v[0];
    t += xor_lane<8>(t);
    t += xor_lane<4>(t);
    t += xor_lane<2>(t);
    t += xor_lane<1>(t);
    return t;
}

template <bool FIRST>
__device__ __forceinline__ void wl_acc2(double &dst, double a, double b, double c, double d)
{
    dst = FIRST ? fma(a, b, c * d) : fma(a, b, fma(c, d, dst));
}
template <bool FIRST>
__device__ __forceinline__ void wl_sub(double &dst, double t) { dst = FIRST ? -t : dst - t; }

struct WlJg { double a00, a01, a02, a10, a11, a12, x, y; };   // JgA of ba_math.h, five multiplies
__device__ __forceinline__ WlJg wl_make_jg(double x, double y, double Z)
{
    WlJg j;
    const double zx = Z * x, zy = Z * y;
    j.a00 = zx * y;  j.a01 = -fma(zx, x, Z); j.a02 = zy;
    j.a10 = fma(zy, y, Z); j.a11 = -j.a00;   j.a12 = -zx;
    j.x = x; j.y = y;
    return j;
}

// entry (i, jj) of Jg^T T added to dst; T = k Jg(d) given by its 2 x 6 entries (columns 3, 4 are -k)
template <bool FIRST, int I>
__device__ __forceinline__ void wl_entry(double &dst, const WlJg &j, double t0, double t1)
{
    if (I == 0) wl_acc2<FIRST>(dst, j.a00, t0, j.a10, t1);
    else if (I == 1) wl_acc2<FIRST>(dst, j.a01, t0, j.a11, t1);
    else if (I == 2) wl_acc2<FIRST>(dst, j.a02, t0, j.a12, t1);
    else if (I == 3) wl_sub<FIRST>(dst, t0);
    else if (I == 4) wl_sub<FIRST>(dst, t1);
    else wl_acc2<FIRST>(dst, j.x, t0, j.y, t1);
}

__device__ __forceinline__ void wl_k_times_jg(double k00, double k01, double k10, double k11, const WlJg &j, double (&T0)[6],
                                              double (&T1)[6])
{
    T0[0] = fma(k00, j.a00, k01 * j.a10); T0[1] = fma(k00, j.a01, k01 * j.a11); T0[2] = fma(k00, j.a02, k01 * j.a12);
    T1[0] = fma(k10, j.a00, k11 * j.a10); T1[1] = fma(k10, j.a01, k11 * j.a11); T1[2] = fma(k10, j.a02, k11 * j.a12);
    T0[3] = -k00; T0[4] = -k01; T0[5] = fma(k00, j.x, k01 * j.y);
    T1[3] = -k10; T1[4] = -k11; T1[5] = fma(k10, j.x, k11 * j.y);
}

// camera pair number p (order (0,1), (0,2), .., (1,2), ..) -> its cameras
template <int C> __device__ __forceinline__ constexpr int wl_pair_c(int p)
{
    return C == 4 ? (p < 3 ? 0 : (p < 5 ? 1 : 2)) : (C == 3 ? (p < 2 ? 0 : 1) : 0);
}
template <int C> __device__ __forceinline__ constexpr int wl_pair_d(int p)
{
    return C == 4 ? (p < 3 ? p + 1 : (p < 5 ? p - 1 : 3)) : (C == 3 ? (p < 2 ? p + 1 : 2) : 1);
}

// the lane's LDS column: element e (a double2) of this thread at st[e * kBlock]
struct WlStash {
    double2 *st;
    __device__ __forceinline__ void put(int slot, int k, double a, double b) const { st[(slot * 3 + k) * kBlock] = make_double2(a, b); }
    __device__ __forceinline__ double2 get(int slot, int k) const { return st[(slot * 3 + k) * kBlock]; }
};

// A double that lives in two accumulation registers (a0..a255: at one wave per SIMD a wave owns 256 of them beside its 256
// vector registers).  VALU instructions cannot read them, so a value costs one v_accvgpr_read/write per half and access:
// the place for what is touched rarely -- x, y, Z of every factor (read once per window) and the window totals.  The
// compiler allocates and tracks them through the "a" constraint; left to itself it spilled ~1 200 halves per chunk there.
struct AReg { int lo, hi; };
__device__ __forceinline__ void a_put(AReg &r, double v)
{
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.lo) : "v"(__double2loint(v)));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.hi) : "v"(__double2hiint(v)));
}
__device__ __forceinline__ double a_get(const AReg &r)
{
    int lo, hi;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(r.lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(r.hi));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void a_add(AReg &r, double v) { a_put(r, a_get(r) + v); }

// make_factor of ba_math.h with ONE select where that one has sixteen: an unused factor (masked, or behind its camera) gets
// the scale 0, which makes E, F and f exact zeros, and its x, y, Z -- finite by construction: a point behind the camera is
// divided by 1 -- then only ever multiply zeros downstream (Uh = F PR L^-T = 0, k = 0, T = 0).  u, v must be finite.
template <bool NODIST, class CamPtr>
__device__ __forceinline__ Factor wl_make_factor(const CamPtr &cam, double px, double py, double pz, double u, double v, bool seen)
{
    Factor o;
    const double dx = px - cam[9], dy = py - cam[10], dz = pz - cam[11];
    const double X = fma(cam[0], dx, fma(cam[3], dy, cam[6] * dz));
    const double Y = fma(cam[1], dx, fma(cam[4], dy, cam[7] * dz));
    const double Z = fma(cam[2], dx, fma(cam[5], dy, cam[8] * dz));
    const bool front = Z > 0.0;
    const double iz = mqs::rcp(front ? Z : 1.0);
    const double x = X * iz, y = Y * iz;
    const double fx = cam[12], fy = cam[13], sk = cam[14], u0 = cam[15], v0 = cam[16];
    const double isig = cam[21];
    const bool ok = seen && front;
    const double sc = ok ? isig * iz : 0.0;
    double eu, ev, E00, E01, E10, E11;
    if (NODIST) {                                  // no lens distortion on any camera (kernel-uniform): E = K / (sigma Z)
        eu = (fma(fx, x, fma(sk, y, u0)) - u) * isig;
        ev = (fma(fy, y, v0) - v) * isig;
        E00 = fx * sc; E01 = sk * sc; E10 = 0.0; E11 = fy * sc;
    } else {
        const double k1 = cam[17], k2 = cam[18], p1 = cam[19], p2 = cam[20];
        const double xx = x * x, yy = y * y, xy = x * y;
        const double r2 = xx + yy;
        const double g = fma(r2, fma(k2, r2, k1), 1.0);
        const double dg = fma(2.0 * k2, r2, k1);
        const double xd = fma(g, x, fma(2.0 * p1, xy, p2 * fma(2.0, xx, r2)));
        const double yd = fma(g, y, fma(2.0 * p2, xy, p1 * fma(2.0, yy, r2)));
        eu = (fma(fx, xd, fma(sk, yd, u0)) - u) * isig;
        ev = (fma(fy, yd, v0) - v) * isig;
        const double a = fma(2.0 * xx, dg, g) + 2.0 * p1 * y + 6.0 * p2 * x;
        const double b = fma(2.0 * xy, dg, 2.0 * p1 * x) + 2.0 * p2 * y;
        const double d = fma(2.0 * yy, dg, g) + 2.0 * p2 * x + 6.0 * p1 * y;
        E00 = fma(fx, a, sk * b) * sc; E01 = fma(fx, b, sk * d) * sc;
        E10 = fy * b * sc; E11 = fy * d * sc;
    }
    o.x = x; o.y = y; o.Z = front ? Z : 1.0;
    if (NODIST) {                                  // E10 = 0
        o.F00 = E00 * E00;
        o.F01 = E00 * E01;
        o.f0 = E00 * eu;
    } else {
        o.F00 = fma(E00, E00, E10 * E10);
        o.F01 = fma(E00, E01, E10 * E11);
        o.f0 = fma(E00, eu, E10 * ev);
    }
    o.F11 = fma(E01, E01, E11 * E11);
    o.f1 = fma(E01, eu, E11 * ev);
    // two selects (the values are finite on every path: u, v are zeros for an unseen factor, Z is 1 behind the camera); as
    // nested conditionals the compiler built two divergent regions of ~10 scalar instructions per factor around them
    const double ce = cam[22];
    double he = 0.5 * fma(eu, eu, ev * ev), ce2 = ce * ce;
    asm volatile("" : "+v"(he), "+v"(ce2));
    he = front ? he : ce2;
    o.half_e2 = seen ? he : 0.0;
    o.valid = ok;
    return o;
}

template <class CamPtr>
__device__ __forceinline__ void wl_make_PR(const CamPtr &cam, double x, double y, double PR[2][3])   // make_PR of ba_math.h
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        PR[0][k] = fma(-x, cam[3 * k + 2], cam[3 * k + 0]);
        PR[1][k] = fma(-y, cam[3 * k + 2], cam[3 * k + 1]);
    }
}

// Makes a value materialise HERE: without it the compiler sinks the landmark-block sums of pass A (and every product that
// feeds them: 4 cameras x 3 landmarks x 15 doubles) down to their first use after the loop and parks them in between.
__device__ __forceinline__ void wl_pin(double &x) { asm volatile("" : "+v"(x)); }

// The staged camera blocks as the wave lineariser reads them: through the CONSTANT address space at a wave-uniform address, i.e.
// with scalar loads into scalar registers (s_load_dwordx16), which fp64 instructions take as an operand for free.  From LDS a
// camera's 23 constants cost ten ds_read_b128 and as many waits per FACTOR (the compiler did not keep them over a camera's four
// landmarks at this register pressure): ~470 of the chunk's instructions.  The kernel writes the blocks to a slice of its
// workspace first (wl_publish_cams).
typedef const __attribute__((address_space(4))) double *WlCam;
// Both forms are built (the kernel's SCALAR parameter): the scalar one costs a prologue (publish, wait for the stores, empty the
// scalar cache) and a scalar-load round trip per camera and chunk, ~1.3 us that a wave with ONE short chunk does not earn back
// (125 k landmarks: 19.0 against 17.8 us); with four chunks it is 3.6 - 5 us ahead (1e6: 100.0 against 103.6 us on one box).
// The launcher picks by landmarks per wave (kWlScalarMinLandmarks).
#ifndef MQS_WL_SCALAR_CAMS
#define MQS_WL_SCALAR_CAMS -1               // A/B: 0 = always from LDS, 1 = always scalar, -1 = by size
#endif
constexpr int64_t kWlScalarMinLandmarks = 400000;      // ~1.5 chunks of 256 landmarks per wave at 1 024 waves

#if defined(MQS_WL_PROBE_WAIT)               // timing probe builds only (tools/probes/wl_wait_probe.py): shader cycles a wave spends in waits
__device__ unsigned long long g_wl_probe[4][1024];
#define MQS_WL_PROBE(slot, what)                                                                    \
    {                                                                                               \
        const unsigned long long t0_ = __builtin_readcyclecounter();                                \
        asm volatile(what ::: "memory");                                                            \
        const unsigned long long t1_ = __builtin_readcyclecounter();                                \
        if (lane == 0) g_wl_probe[slot][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023] += t1_ - t0_; \
    }
extern "C" int mqs_debug_wl_probe(unsigned long long *out, int clear)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wl_probe), sizeof(g_wl_probe)) != hipSuccess) return -1;
    if (clear) { static unsigned long long z[4][1024]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wl_probe), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define MQS_WL_PROBE(slot, what) {}
#endif

// the scalar loads of a camera's block are issued HERE (per chunk and camera; hoisted out of the chunk loop, four cameras' worth of
// scalar registers spilled into vector lanes and came back through v_readlane)
__device__ __forceinline__ void wl_cam_here(WlCam &cam) { asm volatile("" : "+s"(cam)); }
__device__ __forceinline__ void wl_cam_here(const double *&) {}

// A camera block as wl_chunk's two passes see it.  LDS form: the pointer.  Scalar form: the pointer again (the compiler issues the
// scalar loads where the first use is and waits on the spot, twice per camera where it is short of scalar registers) -- or, in the
// A/B build MQS_WL_EARLY_CAMS=1, the block IN scalar registers, brought there by s_load instructions this file issues itself and
// EARLY: the block of camera c + 1 is requested inside camera c's LAST landmark, right behind the last use of camera c's registers
// (the same registers serve both), and waited for at its first use, a landmark's worth of arithmetic later.  Round 4 built this
// because the round-3 verdict suspected those round trips (8 camera visits per chunk x 1-2 dependent scalar loads) behind the
// lineariser's SQ_WAIT_ANY quarter; measured on one device, interleaved (profiles/r04/03_lineariser_ab.json): 90.3 us with the
// early loads against 90.0 us without -- the scalar cache answers fast enough that the waits were never exposed.  Kept as the
// A/B form, off.  (The compiler does not know such loads are in flight; that is safe: its own counted lgkmcnt waits only become
// more conservative with one more operation in the counter, and every use of the registers is behind wait().)
#ifndef MQS_WL_EARLY_CAMS
#define MQS_WL_EARLY_CAMS 0                 // A/B: 1 = the scalar loads issued early by hand (below); 0 = the compiler's own, at the first use
#endif
#ifndef MQS_WL_TOT_ATOMIC
#define MQS_WL_TOT_ATOMIC 1                 // A/B: 0 = read + add + write of the window totals (round 3)
#endif
typedef int wl_i16 __attribute__((ext_vector_type(16)));
typedef int wl_i8 __attribute__((ext_vector_type(8)));
typedef int wl_i4 __attribute__((ext_vector_type(4)));
typedef int wl_i2 __attribute__((ext_vector_type(2)));
#define MQS_WL_D(v, k) __hiloint2double((v)[2 * (k) + 1], (v)[2 * (k)])

template <bool NODIST, class P> struct WlCamA;             // pass A: cam[0..16], cam[21], cam[22] (+ cam[17..20] with distortion)
template <bool NODIST> struct WlCamA<NODIST, const double *> {
    const double *p;
    __device__ __forceinline__ void issue(const double *b) { p = b; }
    __device__ __forceinline__ void wait() {}
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
#if !MQS_WL_EARLY_CAMS
template <bool NODIST> struct WlCamA<NODIST, WlCam> {
    WlCam p;
    __device__ __forceinline__ void issue(WlCam b) { p = b; wl_cam_here(p); }
    __device__ __forceinline__ void wait() {}
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
#else
template <bool NODIST> struct WlCamA<NODIST, WlCam> {
    wl_i16 q0, q1;          // cam[0..7], cam[8..15]
    wl_i2 q2;               // cam[16]
    wl_i4 q3;               // cam[21], cam[22]
    wl_i8 q4;               // cam[17..20] (distortion only)
    __device__ __forceinline__ void issue(WlCam b)
    {
        if (NODIST)
            asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx2 %2, %4, 0x80\n\ts_load_dwordx4 %3, %4, 0xa8"
                         : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3) : "s"(b) : "memory");
        else
            asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx2 %2, %5, 0x80\n\ts_load_dwordx4 %3, %5, 0xa8\n\t"
                         "s_load_dwordx8 %4, %5, 0x88"
                         : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3), "=&s"(q4) : "s"(b) : "memory");
    }
    __device__ __forceinline__ void wait()
    {
        if (NODIST) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3), "+s"(q4) : : "memory");
    }
    __device__ __forceinline__ double operator[](int k) const
    {
        if (k < 8) return MQS_WL_D(q0, k);
        if (k < 16) return MQS_WL_D(q1, k - 8);
        if (k == 16) return MQS_WL_D(q2, 0);
        if (k < 21) return MQS_WL_D(q4, k - 17);
        return MQS_WL_D(q3, k - 21);
    }
};
#endif
template <class P> struct WlCamR;                          // diagonal pass: the rotation, cam[0..8]
template <> struct WlCamR<const double *> {
    const double *p;
    __device__ __forceinline__ void issue(const double *b) { p = b; }
    __device__ __forceinline__ void wait() {}
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
#if !MQS_WL_EARLY_CAMS
template <> struct WlCamR<WlCam> {
    WlCam p;
    __device__ __forceinline__ void issue(WlCam b) { p = b; wl_cam_here(p); }
    __device__ __forceinline__ void wait() {}
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};
#else
template <> struct WlCamR<WlCam> {
    wl_i16 q0;
    wl_i2 q1;
    __device__ __forceinline__ void issue(WlCam b)
    {
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x40" : "=&s"(q0), "=&s"(q1) : "s"(b) : "memory");
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1) : : "memory"); }
    __device__ __forceinline__ double operator[](int k) const { return k < 8 ? MQS_WL_D(q0, k) : MQS_WL_D(q1, 0); }
};
#endif

// A window total into the wave's row of totals: ds_add_f64 without return -- the LDS does the read-modify-write, the wave neither
// reads the old total nor waits for anything (the read + wait + add + write it replaces sat ~100 cycles per window in s_waitcnt:
// the compiler sank the read of the old total down to its use).  One lane per entry and one wave per row, in program order: the
// same sums in the same order as before, bitwise reproducible.
__device__ __forceinline__ void wl_tot_add(double *p, double v)
{
#if MQS_WL_TOT_ATOMIC
    (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) double *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
    *p += v;
#endif
}

template <int C, int L, bool NODIST, class WlCamPtr>
__device__ __forceinline__ void wl_chunk(const WlCamPtr sCam, const WlStash stash, const double *__restrict__ points,
                                         const double2 *__restrict__ obs2, const uint8_t *__restrict__ mask,
                                         const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N,
                                         double lambda, int64_t row0, int lane, double *tot)
{
    using LT = Layout<C>;
#if defined(MQS_WL_PROBE_WAIT)
    const unsigned long long chunk_t0 = __builtin_readcyclecounter();
#endif
    // the stash of landmark l, camera c: LDS for the first kWaveLinLdsL landmarks, accumulation registers for the fourth (the
    // LDS holds three landmarks' worth; l, c, k are compile-time after unrolling, so the choice costs nothing)
    AReg a3[C][6];
    auto sput = [&](int l, int c, int k, double a, double b) {
        if (l < kWaveLinLdsL) stash.put(l * C + c, k, a, b);
        else { a_put(a3[c][2 * k], a); a_put(a3[c][2 * k + 1], b); }
    };
    auto sget = [&](int l, int c, int k) -> double2 {
        if (l < kWaveLinLdsL) return stash.get(l * C + c, k);
        return make_double2(a_get(a3[c][2 * k]), a_get(a3[c][2 * k + 1]));
    };
    double px[L], py[L], pz[L];
    bool live[L];
    int64_t idx[L];
    PointSystem ps[L];
    double cost = 0.0, count = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int64_t i = (row0 + l) * 64 + lane;
        live[l] = i < N;
        idx[l] = live[l] ? i : 0;
#if defined(MQS_WL_EXPERIMENT_NOLOAD)      // timing experiment only: what the exposed memory latency of a chunk costs (wrong results)
        px[l] = 0.01 * lane + (double)l; py[l] = 0.02 * lane - 0.5; pz[l] = 0.25 * (double)(i & 7);
#else
        px[l] = points[3 * idx[l] + 0]; py[l] = points[3 * idx[l] + 1]; pz[l] = points[3 * idx[l] + 2];
#endif
    }
    WlCamA<NODIST, WlCamPtr> camA;          // the camera of pass A: camera 0's block is requested with the chunk's head loads
    camA.issue(sCam);
    double2 ob[2][L];                       // this camera's and the next one's measurements
    uint8_t mk[2][L];                       // and their mask bytes: loaded with the measurements, never behind a wait of their own
    double pwl[L];                          // the prior weights: all loads of the chunk's head are in flight before the first wait
#pragma unroll
    for (int l = 0; l < L; ++l) {
#if defined(MQS_WL_EXPERIMENT_NOLOAD)
        ob[0][l] = make_double2(300.0 + lane, 200.0 + (double)l);
        mk[0][l] = 1;
        pwl[l] = 0.0;
#else
        ob[0][l] = obs2[idx[l]];
        mk[0][l] = mask ? mask[idx[l]] : (uint8_t)1;
        pwl[l] = prior_w ? prior_w[idx[l]] : 0.0;
#endif
    }
    MQS_WL_PROBE(0, "s_waitcnt vmcnt(0)")
#pragma unroll
    for (int l = 0; l < L; ++l) {
        // PriorFactor<Point3>: enters the landmark block at its start (H = w I, g = -w (p - p0)), so nothing of it stays live
        double pw = 0.0, ddx = 0.0, ddy = 0.0, ddz = 0.0;
        if (prior_w) {
            const double w = pwl[l];
            if (live[l] && w > 0.0) {
                pw = w;
                ddx = px[l] - prior_xyz[3 * idx[l] + 0];
                ddy = py[l] - prior_xyz[3 * idx[l] + 1];
                ddz = pz[l] - prior_xyz[3 * idx[l] + 2];
            }
        }
        ps[l].H = mqs::Sym3{pw, 0, 0, pw, 0, pw};
        ps[l].g = mqs::Vec3{-pw * ddx, -pw * ddy, -pw * ddz};
        cost += 0.5 * pw * fma(ddx, ddx, fma(ddy, ddy, ddz * ddz));
    }

    // ---- pass A: every factor once; landmark blocks; F, f parked in LDS; x, y, Z kept ----
    AReg X[L][C], Y[L][C], Z[L][C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (c + 1 < C) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
#if defined(MQS_WL_EXPERIMENT_NOLOAD)
                ob[(c + 1) & 1][l] = make_double2(300.0 + lane + c, 200.0 + (double)l);
                mk[(c + 1) & 1][l] = 1;
#else
                ob[(c + 1) & 1][l] = obs2[(int64_t)(c + 1) * N + idx[l]];
                mk[(c + 1) & 1][l] = mask ? mask[(int64_t)(c + 1) * N + idx[l]] : (uint8_t)1;
#endif
            }
        }
        if (c > 0 && c + 1 < C) MQS_WL_PROBE(1, "s_waitcnt vmcnt(4)")      // camera c's measurements (the 4 loads of camera c + 1 may stay out; L = 4, no mask)
        if (c > 0 && c + 1 == C) MQS_WL_PROBE(1, "s_waitcnt vmcnt(0)")
        camA.wait();                                                        // camera c's block (requested a landmark ago)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            unsigned mbyte = mk[c & 1][l];
            asm volatile("" : "+v"(mbyte));           // compared HERE: at the load the compare is a wait for one byte, per factor
            const bool seen = live[l] && mbyte != 0;
            // a masked slot may hold NaN: selects, not products
            const double u = seen ? ob[c & 1][l].x : 0.0, v = seen ? ob[c & 1][l].y : 0.0;
            const Factor fc = wl_make_factor<NODIST>(camA, px[l], py[l], pz[l], u, v, seen);
            double PR[2][3];
            wl_make_PR(camA, fc.x, fc.y, PR);
            // the stash stores first: the wait for the next camera's block (an lgkmcnt(0), a landmark further on) then finds them done
            sput(l, c, 0, fc.F00, fc.F01);
            sput(l, c, 1, fc.F11, fc.f0);
            sput(l, c, 2, fc.f1, 0.0);
            a_put(X[l][c], fc.x); a_put(Y[l][c], fc.y); a_put(Z[l][c], fc.Z);
            if (l == L - 1 && c + 1 < C) {
                // camera c's registers are dead from here: the next camera's block is requested into them now and lands while
                // this landmark's block sums issue
                double keep = PR[1][2];
                asm volatile("" : "+v"(keep));        // every use of camera c above this line
                PR[1][2] = keep;
                MQS_SCHED_FENCE();
                camA.issue(sCam + kCamStride * (c + 1));
            }
            point_add_factor(ps[l], fc, PR);
            cost += fc.half_e2;
            count += fc.valid ? 1.0 : 0.0;
            wl_pin(ps[l].H.xx); wl_pin(ps[l].H.xy); wl_pin(ps[l].H.xz); wl_pin(ps[l].H.yy); wl_pin(ps[l].H.yz); wl_pin(ps[l].H.zz);
            wl_pin(ps[l].g.x); wl_pin(ps[l].g.y); wl_pin(ps[l].g.z);
            wl_pin(cost); wl_pin(count);
            MQS_SCHED_FENCE();
        }
    }
    asm volatile("; MQS_MARK pass_a_done");
    WlCamR<WlCamPtr> camR;                  // the camera of the diagonal pass: camera 0's rotation lands under the landmark solves below
    camR.issue(sCam);
    double w0[L], w1[L], w2[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        point_finish(ps[l], 0.0, 0.0, 0.0, 0.0, lambda);
        if (!ps[l].ok) { ps[l].i00 = 0.0; ps[l].i11 = 0.0; ps[l].i22 = 0.0; }      // unconstrained: Uh = 0, w = 0
        w0[l] = ps[l].g.x * ps[l].i00;
        w1[l] = fma(-ps[l].l10, w0[l], ps[l].g.y) * ps[l].i11;
        w2[l] = fma(-ps[l].l21, w1[l], fma(-ps[l].l20, w0[l], ps[l].g.z)) * ps[l].i22;
    }

    // ---- diagonal blocks and gradient: one window per camera ----
    asm volatile("; MQS_MARK finish_done");
    // A wave alone on its SIMD hides no LDS latency by itself: every block's stash reads are issued one block ahead
    // (double-buffered in registers), so that they land while the previous block's arithmetic issues.
    double2 dq[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dq[0][k] = sget(0, 0, k);
    camR.wait();
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double buf[32];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int step = c * L + l, cur = step & 1;
            if (step + 1 < C * L) {
                const int nl = (l + 1 < L) ? l + 1 : 0, nc = (l + 1 < L) ? c : c + 1;
#pragma unroll
                for (int k = 0; k < 3; ++k) dq[cur ^ 1][k] = sget(nl, nc, k);
            }
            const double2 s0 = dq[cur][0], s1 = dq[cur][1], s2 = dq[cur][2];
            const double F00 = s0.x, F01 = s0.y, F11 = s1.x, f0 = s1.y, f1 = s2.x;
            const double xc = a_get(X[l][c]), yc = a_get(Y[l][c]), zc = a_get(Z[l][c]);
            double PR[2][3];
            wl_make_PR(camR, xc, yc, PR);
            if (l == L - 1 && c + 1 < C) {          // as in pass A: the next camera's rotation into the registers this one has just left
                double keep = PR[1][2];
                asm volatile("" : "+v"(keep));
                PR[1][2] = keep;
                MQS_SCHED_FENCE();
                camR.issue(sCam + kCamStride * (c + 1));
            }
            double U[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double Fa = r ? F01 : F00, Fb = r ? F11 : F01;
                double u0 = fma(Fa, PR[0][0], Fb * PR[1][0]);
                double u1 = fma(Fa, PR[0][1], Fb * PR[1][1]);
                double u2 = fma(Fa, PR[0][2], Fb * PR[1][2]);
                apply_LinvT(ps[l], u0, u1, u2);
                U[r][0] = u0; U[r][1] = u1; U[r][2] = u2;
            }
            sput(l, c, 0, U[0][0], U[0][1]);
            sput(l, c, 1, U[0][2], U[1][0]);
            sput(l, c, 2, U[1][1], U[1][2]);
            // F - U U^T and -(f + U w) as chains of three FMAs each (no separate product and subtraction)
            const double k00 = fma(-U[0][2], U[0][2], fma(-U[0][1], U[0][1], fma(-U[0][0], U[0][0], F00)));
            const double k01 = fma(-U[0][2], U[1][2], fma(-U[0][1], U[1][1], fma(-U[0][0], U[1][0], F01)));
            const double k11 = fma(-U[1][2], U[1][2], fma(-U[1][1], U[1][1], fma(-U[1][0], U[1][0], F11)));
            const double rh0 = fma(-U[0][2], w2[l], fma(-U[0][1], w1[l], fma(-U[0][0], w0[l], -f0)));
            const double rh1 = fma(-U[1][2], w2[l], fma(-U[1][1], w1[l], fma(-U[1][0], w0[l], -f1)));
            const WlJg jg = wl_make_jg(xc, yc, zc);
            double T0[6], T1[6];
            wl_k_times_jg(k00, k01, k01, k11, jg, T0, T1);
#define MQS_WL_DIAG(FIRST)                                                                                         \
    {                                                                                                              \
        wl_entry<FIRST, 0>(buf[0], jg, T0[0], T1[0]); wl_entry<FIRST, 0>(buf[1], jg, T0[1], T1[1]);               \
        wl_entry<FIRST, 0>(buf[2], jg, T0[2], T1[2]); wl_entry<FIRST, 0>(buf[3], jg, T0[3], T1[3]);               \
        wl_entry<FIRST, 0>(buf[4], jg, T0[4], T1[4]); wl_entry<FIRST, 0>(buf[5], jg, T0[5], T1[5]);               \
        wl_entry<FIRST, 1>(buf[6], jg, T0[1], T1[1]); wl_entry<FIRST, 1>(buf[7], jg, T0[2], T1[2]);               \
        wl_entry<FIRST, 1>(buf[8], jg, T0[3], T1[3]); wl_entry<FIRST, 1>(buf[9], jg, T0[4], T1[4]);               \
        wl_entry<FIRST, 1>(buf[10], jg, T0[5], T1[5]);                                                            \
        wl_entry<FIRST, 2>(buf[11], jg, T0[2], T1[2]); wl_entry<FIRST, 2>(buf[12], jg, T0[3], T1[3]);             \
        wl_entry<FIRST, 2>(buf[13], jg, T0[4], T1[4]); wl_entry<FIRST, 2>(buf[14], jg, T0[5], T1[5]);             \
        wl_entry<FIRST, 3>(buf[15], jg, T0[3], T1[3]); wl_entry<FIRST, 3>(buf[16], jg, T0[4], T1[4]);             \
        wl_entry<FIRST, 3>(buf[17], jg, T0[5], T1[5]);                                                            \
        wl_entry<FIRST, 4>(buf[18], jg, T0[4], T1[4]); wl_entry<FIRST, 4>(buf[19], jg, T0[5], T1[5]);             \
        wl_entry<FIRST, 5>(buf[20], jg, T0[5], T1[5]);                                                            \
        wl_entry<FIRST, 0>(buf[21], jg, rh0, rh1); wl_entry<FIRST, 1>(buf[22], jg, rh0, rh1);                     \
        wl_entry<FIRST, 2>(buf[23], jg, rh0, rh1); wl_entry<FIRST, 3>(buf[24], jg, rh0, rh1);                     \
        wl_entry<FIRST, 4>(buf[25], jg, rh0, rh1); wl_entry<FIRST, 5>(buf[26], jg, rh0, rh1);                     \
    }
            if (l == 0) MQS_WL_DIAG(true) else MQS_WL_DIAG(false)
#undef MQS_WL_DIAG
            MQS_SCHED_FENCE();
        }
        asm volatile("; MQS_MARK diag_entries_done");
        buf[27] = (c == 0) ? cost : 0.0;
        buf[28] = (c == 0) ? count : 0.0;
        buf[29] = 0.0; buf[30] = 0.0; buf[31] = 0.0;
        // the next camera's rotation (requested a landmark ago: landed) is collected HERE, before the window total goes to the LDS:
        // the wait is an lgkmcnt(0) and would sit out the LDS add behind it otherwise
        if (c + 1 < C) camR.wait();
        {
            // window total -> this wave's row of totals in LDS (Layout<C> numbering; lanes 2j and 2j+1 hold entry j), added
            // there by the LDS itself (wl_tot_add)
            const int j = lane >> 1;
            const int slot = (j < LT::kDiagUsed) ? LT::diag_off(c) + j : ((c == 0 && j == 27) ? LT::kCost : ((c == 0 && j == 28) ? LT::kCount : LT::kCount + 1));
            const double t = wave_reduce32(buf, lane);               // kCount + 1: a spare word of the row
            if ((lane & 1) == 0) wl_tot_add(tot + slot, t);
        }
        MQS_SCHED_FENCE();
    }

    // ---- off-diagonal blocks S_cd = -Jg_c^T (Uh_c Uh_d^T) Jg_d: a 32-entry and a 4-entry window per pair ----
    asm volatile("; MQS_MARK diag_done");
    constexpr int NP = C * (C - 1) / 2;
    double2 pq[2][6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { pq[0][k] = sget(0, 0, k); pq[0][3 + k] = sget(0, 1, k); }      // pair (0, 1), landmark 0
#pragma unroll
    for (int pair = 0; pair < NP; ++pair) {
        {
            const int c = wl_pair_c<C>(pair), d = wl_pair_d<C>(pair);
            double buf[32], b4[4];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int step = pair * L + l, cur = step & 1;
                if (step + 1 < NP * L) {
                    const int nl = (l + 1 < L) ? l + 1 : 0, np = (l + 1 < L) ? pair : pair + 1;
                    const int nc = wl_pair_c<C>(np), nd = wl_pair_d<C>(np);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { pq[cur ^ 1][k] = sget(nl, nc, k); pq[cur ^ 1][3 + k] = sget(nl, nd, k); }
                }
                const double2 a0 = pq[cur][0], a1 = pq[cur][1], a2 = pq[cur][2];
                const double2 e0 = pq[cur][3], e1 = pq[cur][4], e2 = pq[cur][5];
                // Uc = [[a0.x a0.y a1.x],[a1.y a2.x a2.y]], Ud likewise
                const double k00 = -fma(a0.x, e0.x, fma(a0.y, e0.y, a1.x * e1.x));
                const double k01 = -fma(a0.x, e1.y, fma(a0.y, e2.x, a1.x * e2.y));
                const double k10 = -fma(a1.y, e0.x, fma(a2.x, e0.y, a2.y * e1.x));
                const double k11 = -fma(a1.y, e1.y, fma(a2.x, e2.x, a2.y * e2.y));
                const WlJg jd = wl_make_jg(a_get(X[l][d]), a_get(Y[l][d]), a_get(Z[l][d]));
                double T0[6], T1[6];
                wl_k_times_jg(k00, k01, k10, k11, jd, T0, T1);
                const WlJg jc = wl_make_jg(a_get(X[l][c]), a_get(Y[l][c]), a_get(Z[l][c]));
#define MQS_WL_ROW(FIRST, I, B)                                                                                    \
    wl_entry<FIRST, I>(B[0], jc, T0[0], T1[0]); wl_entry<FIRST, I>(B[1], jc, T0[1], T1[1]);                       \
    wl_entry<FIRST, I>(B[2], jc, T0[2], T1[2]); wl_entry<FIRST, I>(B[3], jc, T0[3], T1[3]);                       \
    wl_entry<FIRST, I>(B[4], jc, T0[4], T1[4]); wl_entry<FIRST, I>(B[5], jc, T0[5], T1[5]);
#define MQS_WL_PAIR(FIRST)                                                                                         \
    {                                                                                                              \
        double *r0 = buf, *r1 = buf + 6, *r2 = buf + 12, *r3 = buf + 18, *r4 = buf + 24;                           \
        MQS_WL_ROW(FIRST, 0, r0) MQS_WL_ROW(FIRST, 1, r1) MQS_WL_ROW(FIRST, 2, r2) MQS_WL_ROW(FIRST, 3, r3)        \
        MQS_WL_ROW(FIRST, 4, r4)                                                                                   \
        wl_entry<FIRST, 5>(buf[30], jc, T0[0], T1[0]); wl_entry<FIRST, 5>(buf[31], jc, T0[1], T1[1]);             \
        wl_entry<FIRST, 5>(b4[0], jc, T0[2], T1[2]); wl_entry<FIRST, 5>(b4[1], jc, T0[3], T1[3]);                 \
        wl_entry<FIRST, 5>(b4[2], jc, T0[4], T1[4]); wl_entry<FIRST, 5>(b4[3], jc, T0[5], T1[5]);                 \
    }
                if (l == 0) MQS_WL_PAIR(true) else MQS_WL_PAIR(false)
#undef MQS_WL_PAIR
#undef MQS_WL_ROW
                MQS_SCHED_FENCE();
            }
            asm volatile("; MQS_MARK pair_entries_done");
            {
                const double t32 = wave_reduce32(buf, lane), t4 = wave_reduce4(b4, lane);
                if ((lane & 1) == 0) wl_tot_add(tot + LT::pair_off(c, d) + (lane >> 1), t32);
                if ((lane & 15) == 0) wl_tot_add(tot + LT::pair_off(c, d) + 32 + (lane >> 4), t4);
            }
            asm volatile("; MQS_MARK pair_reduced");
            MQS_SCHED_FENCE();
        }
    }
#if defined(MQS_WL_PROBE_WAIT)
    if (lane == 0) g_wl_probe[3][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023] += __builtin_readcyclecounter() - chunk_t0;
#endif
}

template <int C> constexpr int kWlCamSlice = (C * kCamStride + 15) / 16 * 16;      // doubles per workgroup: whole 128-byte lines
template <int C, bool SCALAR>
__global__ __launch_bounds__(kBlock, kWaveLinOcc) void ba_linearize_wave_kernel(
    const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double lambda,
    double *__restrict__ partials)
{
    using L = Layout<C>;
    constexpr int NCH = L::kChunks;
    constexpr int kRow = NCH * 32;
    // the camera blocks are a separate (static) LDS object: the compiler then knows that a stash write cannot change them
    // and keeps a camera's constants in registers over its three landmarks instead of re-reading them after every write
    __shared__ double sCam[C * kCamStride];
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_smem[];
    double2 *sStash = reinterpret_cast<double2 *>(wl_smem);                                // [kWaveLinLdsL * C * 3][kBlock]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // the row walk is scalar
    stage_cams<C>(poses, calib, sigma, sCam, tid);
    WlCam cams_scalar = nullptr;
    if (SCALAR) {
    // this workgroup's copy of the camera blocks in global memory (a 128-byte-aligned slice of the workspace behind the rows of
    // partials: no other workgroup's bytes share a cache line with it), for the scalar loads of wl_chunk.  Write-through stores,
    // acknowledged (mqs_stores_landed) before the barrier; the scalar cache is emptied of what an earlier launch may have left
    // for these addresses; the address reaches the loads through an opaque register, so that no load is scheduled above this.
    double *cam_slice = partials + (int64_t)256 * kRow + (int64_t)blockIdx.x * kWlCamSlice<C>;
    for (int k = tid; k < C * kCamStride; k += kBlock) __hip_atomic_store(cam_slice + k, sCam[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mqs_stores_landed();
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cams_scalar = (WlCam)(cam_slice);
    asm volatile("" : "+s"(cams_scalar) : : "memory");
    }
    typename std::conditional<SCALAR, WlCam, const double *>::type cams;
    if constexpr (SCALAR) cams = cams_scalar; else cams = sCam;

    // window totals of this wave: a row in Layout<C> numbering behind the stash (entry j of a window lives in lanes 2j, 2j+1)
    double *sTot = reinterpret_cast<double *>(wl_smem + sizeof(double2) * kWaveLinLdsL * C * 3 * kBlock);   // [kWaves][kRow]
    double *tot = sTot + wave * kRow;
    for (int k = lane; k < kRow; k += 64) tot[k] = 0.0;
    mqs_wave_lds_sync();

    // contiguous rows of 64 landmarks per wave, walked in chunks of up to kWaveLinMaxL rows
    const int64_t rows = (N + 63) / 64;
    const int64_t nw = (int64_t)gridDim.x * kWaves, gw = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t r_begin = rows * gw / nw, r_end = rows * (gw + 1) / nw;
    const WlStash stash = {sStash + tid};
    const double2 *obs2 = reinterpret_cast<const double2 *>(obs);
    bool nodist = true;
#pragma unroll
    for (int c = 0; c < C; ++c) nodist = nodist && camera_without_distortion(sCam + kCamStride * c);
    for (int64_t r = r_begin; r < r_end;) {
        const int64_t left = r_end - r;
        const int nl = left >= kWaveLinMaxL ? kWaveLinMaxL : (int)left;
        // cameras without lens distortion (all of them: decided once per kernel, wave-uniform) take bodies without the distortion
        // model; the bodies are whole chunks, so the instruction stream of a launch never contains the other variant
#define MQS_WL_CALL(LL)                                                                                                   \
    do {                                                                                                                  \
        if (nodist) wl_chunk<C, LL, true>(cams, stash, points, obs2, mask, prior_w, prior_xyz, N, lambda, r, lane, tot);  \
        else wl_chunk<C, LL, false>(cams, stash, points, obs2, mask, prior_w, prior_xyz, N, lambda, r, lane, tot);        \
    } while (0)
        // Which GPU test reaches which chunk body (tests/test_ba_gpu.py): wl_chunk<C, 1..4, true> (no lens distortion):
        // test_wave_lineariser_every_chunk_size_against_the_c_oracle, the four cases without `dist` (N = 70 000 / 150 000 / 180 001 /
        // 262 207: a wave owns 1 to 5 rows, i.e. chunks of 1, 2, 3, 4) and test_full_size_properties_1e6x4 (chunks of 4 and 3);
        // wl_chunk<C, 1..4, false> (k1 = 0.3, or the full Cal3DS2 coefficient set): the ten `-dist` cases of the same test, C = 2, 3, 4,
        // masked and unmasked; wl_chunk<C, 1, *> alone: every small case of test_linearize_and_backsub_parity.  All against the C
        // oracle at 1e-10 on S, g, cost, count and, from the same linearisation point, a back-substitution.  Those sizes run the
        // kernel's LDS form (SCALAR = false); the scalar-load form (N >= kWlScalarMinLandmarks) is walked by the six cases of the
        // same test at N = 400 001 .. 610 000 (chunks of 4 + 2, 4 + 3, 4 + 4 + 1, 4 + 4 + 2; C = 2, 3, 4; with and without
        // distortion and mask; their two shards take the LDS form, so shard additivity compares the forms) and by the 1e6 test.
#if defined(MQS_WL_ONLY_L4)      // ISA counting only (tools/isa_mix.py --define MQS_WL_ONLY_L4 [--define MQS_WL_ONLY_NODIST=1]): one body
        wl_chunk<C, 4, MQS_WL_ONLY_NODIST>(cams, stash, points, obs2, mask, prior_w, prior_xyz, N, lambda, r, lane, tot);
#else
#if MQS_WL_MAXL >= 4
        if (nl == 4) MQS_WL_CALL(4); else
#endif
#if MQS_WL_MAXL >= 3
        if (nl == 3) MQS_WL_CALL(3); else
#endif
        if (nl == 2) MQS_WL_CALL(2);
        else MQS_WL_CALL(1);
#endif
#undef MQS_WL_CALL
        r += nl;
    }
    // waves -> workgroup -> this workgroup's row of partials (Layout<C> numbering: ba_finalize_kernel serves both linearisers)
    __syncthreads();
    for (int s = tid; s < kRow; s += kBlock) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += sTot[w * kRow + s];
        partials[(int64_t)blockIdx.x * kRow + s] = t;
    }
}

// Sums the per-workgroup partials in a fixed order (reproducible: piece_part_sum below) and scatters the slots into
// out = [S | g | cost | count], mirroring S.  One workgroup of 1024 threads per 64 slots: sixteen waves over kFinPieces pieces x
// four parts (64 consecutive slots per row read = one 512-byte coalesced load, eight in flight), combined through LDS.
constexpr int kFinThreads = 1024;

// The order in which partial rows are added, shared by ba_finalize_kernel and the finalizer workgroups of the fused tail
// (ba_tail_kernel) so that both give the same bits: the rows are cut into kFinPieces PIECES of ceil(rows / kFinPieces)
// consecutive rows; inside a piece, part w (0..3) adds rows first + w, first + w + 4, ... in order; a piece's sum is
// ((p0 + p1) + p2) + p3 and the total the left fold (((P0 + P1) + P2) + ...) over the pieces.
constexpr int kFinPieces = MQS_FIN_PIECES;
__device__ __forceinline__ double piece_part_sum(const double *__restrict__ partials, int nrows, int row_stride, int piece, int w, int slot)
{
    const int rq = (nrows + kFinPieces - 1) / kFinPieces, r0 = piece * rq, r1 = min(nrows, r0 + rq);
    double t = 0.0;
    int b = r0 + w;
    for (; b + 4 * 7 < r1; b += 4 * 8) {                           // eight loads in flight, added in order
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partials[(int64_t)(b + 4 * k) * row_stride + slot];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; b < r1; b += 4) t += partials[(int64_t)b * row_stride + slot];
    return t;
}

template <int C>
__global__ __launch_bounds__(kFinThreads) void ba_finalize_kernel(const double *__restrict__ partials, int nblocks,
                                                                  double *__restrict__ out, mqs_peer_push push)
{
    using L = Layout<C>;
    constexpr int NCH = L::kChunks;
    constexpr int kRow = NCH * 32;
    __shared__ double sSum[4 * kFinPieces][64];                        // [4 * piece + part]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    for (int pw = wave; pw < 4 * kFinPieces; pw += kFinThreads / 64)
        sSum[pw][lane] = (s < kRow) ? piece_part_sum(partials, nblocks, kRow, pw >> 2, pw & 3, s) : 0.0;
    __syncthreads();
    if (wave == 0 && s < L::kSlots) {
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < kFinPieces; ++q) {
            const double ps = ((sSum[4 * q][lane] + sSum[4 * q + 1][lane]) + sSum[4 * q + 2][lane]) + sSum[4 * q + 3][lane];
            r = q == 0 ? ps : r + ps;
        }
        int o1, o2;
        slot_to_out<C>(s, o1, o2);
        if (o1 >= 0) out[o1] = r;
        if (o2 >= 0) out[o2] = r;
        // peer transport (comm.hip), ranks that share a GPU: the same entries into slot [rank] of every rank's receive buffer
        if (o1 >= 0) mqs::peer::push_entry(push, o1, r);
        if (o2 >= 0) mqs::peer::push_entry(push, o2, r);
    }
    if (push.world > 0) {
        mqs_stores_landed();
        __syncthreads();                              // every wave's stores above have landed, in this GPU and in the peers
        if (wave == 0) mqs::peer::publish_piece(push, blockIdx.x, lane);
    }
}

#ifndef MQS_BA_BACKSUB_DIRECT
#define MQS_BA_BACKSUB_DIRECT 1
#endif
template <int C>
__global__ __launch_bounds__(kBlock, 4) void ba_backsub_kernel(
    const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double lambda,
    const double *__restrict__ dpose, double *__restrict__ points_out)
{
    __shared__ double sCam[C * kCamStride];
#if !MQS_BA_BACKSUB_DIRECT
    __shared__ double sX[kBlock * 3];
#endif
    __shared__ double sD[6 * C];
    const int tid = threadIdx.x;
    if (tid < 6 * C) sD[tid] = dpose[tid];
    stage_cams<C>(poses, calib, sigma, sCam, tid);

    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t i = base + tid;
        const bool live = i < N;
        double px, py, pz;
#if MQS_BA_BACKSUB_DIRECT
        // Each lane reads and writes its own 24-byte landmark: three 8-byte accesses per lane that together cover whole cache
        // lines across the wave.  The LDS transpose of the first version made them 16-byte coalesced at the price of three
        // workgroup barriers per batch -- in a kernel that is short of work to hide latency, not of bandwidth.
        const int64_t ii = live ? i : 0;
        px = points[3 * ii + 0]; py = points[3 * ii + 1]; pz = points[3 * ii + 2];
        if (!live) { px = 0.0; py = 0.0; pz = 0.0; }
#else
        load_points(points, base, N, sX, tid, px, py, pz);
#endif
        const DevObs ob = {reinterpret_cast<const double2 *>(obs), mask, i, N, live};
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
        const mqs::Vec3 dp = landmark_backsub<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, sD);
#if MQS_BA_BACKSUB_DIRECT
        if (live) {
            points_out[3 * i + 0] = px + dp.x;
            points_out[3 * i + 1] = py + dp.y;
            points_out[3 * i + 2] = pz + dp.z;
        }
#else
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
#endif
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
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    // w = theta / (2 sin theta) * vee(R - R^T), |vee| = 2 sin theta.  A prior sits close to its pose: below 0.1 rad the factor
    // is asin(s) / (2 s) as a series in s^2 = sin^2 theta (seven terms: 1.4e-16 relative at the switch) -- more accurate there
    // than acos of a cosine next to 1, and seven FMAs instead of acos + sin (the log map is on the path of every workgroup of
    // the fused tail when a pose prior is present: ~1 us of a lone wave).
    const double s2 = 0.25 * fma(vx, vx, fma(vy, vy, vz * vz));
    double k;
    if (c > 0.0 && s2 < 0.01) {
        double p = 143.0 / 10240.0;
        p = fma(p, s2, 231.0 / 13312.0);
        p = fma(p, s2, 63.0 / 2816.0);
        p = fma(p, s2, 35.0 / 1152.0);
        p = fma(p, s2, 5.0 / 112.0);
        p = fma(p, s2, 3.0 / 40.0);
        p = fma(p, s2, 1.0 / 6.0);
        k = 0.5 * fma(p, s2, 1.0);
    } else {
        const double th = acos(c);
        k = (th < 1e-10) ? 0.5 : th / (2.0 * sin(th));
    }
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
        if (j == r) row[j] = (lambda >= 0.0) ? (row[j] + sW[r]) * (1.0 + lambda) : (row[j] + sW[r]) - lambda;
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

// ---------------------------------------------------------------------------------------------------------------------
// Reduced camera system for C <= 4 cameras (n = 6C <= 24): ONE wavefront, the whole solve inside the factorisation loop.
//
// The kernel above walks three dependent chains (24 pivots, 24 forward steps, 24 backward steps through LDS columns) and is
// a launch of its own: 10-12 us of an iteration whose other kernels take 30 us at 125 k landmarks.  Here
//   * lanes 0..n-1 hold the rows of S (they become the rows of L), lanes 32..32+n-1 the unit vectors e_c and lane 56 the
//     right-hand side: the forward substitution  v_k <- v_k / L_kk,  v_j <- v_j - v_k L_jk  IS the trailing update
//     row_j <- row_j - L_ik L_jk  with v_k in the place of L_ik, so every lane runs the same instruction stream and the loop
//     ends with L^-1 (one column per lane 32 + c) and y = L^-1 b (lane 56) in registers; x = L^-T y is then one dot
//     product per lane -- no substitution chain at all;
//   * the pivot chain does not go through LDS: step k needs column k of L broadcast to every lane, but the NEXT pivot and the
//     next column only need L[k+1][k], which comes back through a v_readlane pair; the other entries of the column take the
//     LDS round trip (double-buffered) behind the reciprocal square root of the next pivot.
// The function is executed by one full wavefront; every array argument is LDS.  Used by the stand-alone solve kernel and by
// the fused tail of an iteration below (same arithmetic, same bits).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSolveInvLane0 = 32, kSolveRhsLane = 56;

// pose priors of the cameras (bundle_adjust.cpp:273): lanes c < C; e = (Log(R0^T R), R0^T (t - t0)) / sigma, J ~ I
template <int C>
__device__ __forceinline__ void pose_prior_terms(const double *__restrict__ poses, const double *__restrict__ prior_poses,
                                                 const double *__restrict__ prior_sigmas, const uint8_t *__restrict__ prior_mask,
                                                 int lane, double *sE, double *sW, double *sInfo)
{
    constexpr int n = 6 * C;
    if (lane < n) { sE[lane] = 0.0; sW[lane] = 0.0; }
    if (lane == 0) { sInfo[0] = 0.0; sInfo[1] = 0.0; }
    mqs_wave_lds_sync();
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
    mqs_wave_lds_sync();
}

template <int C>
__device__ __forceinline__ void reduced_solve_wave(const double *sM /*[64][n]*/, double *sCol /*[2][64]*/, double *sY /*[n]*/,
                                                   double *sX /*[n]*/, int lane, bool &bad)
{
    constexpr int n = 6 * C;
    static_assert(n <= 24, "rows, inverse columns and the right-hand side share one wavefront");
    const bool isI = lane >= kSolveInvLane0 && lane < kSolveInvLane0 + n, isB = lane == kSolveRhsLane;
    const int ci = lane - kSolveInvLane0;
    // this lane's starting vector: a row of the damped system, a unit vector, or the right-hand side (build_solve_matrix)
    double row[n];
#pragma unroll
    for (int j = 0; j < n; j += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(sM + lane * n + j);
        row[j] = v.x; row[j + 1] = v.y;
    }
    // A lone wave issues one instruction per ~5 cycles and hides no latency by itself, so the loop is written as the machine
    // should run it: per pivot k a short CRITICAL part -- L[.][k] = row[k] / L_kk, the two entries of that column the next two
    // pivots wait for fetched by v_readlane (no LDS round trip), the next pivot, the seed of its reciprocal square root --
    // and, in the shadow of that seed's Newton steps, the part nobody waits for yet: the rest of the PREVIOUS pivot's column
    // (read back from LDS a whole step ago) applied to rows k + 2 and beyond.  sched_barriers keep the compiler from sinking
    // the critical part behind the LDS waits (it did: 6.1 us for the 24 pivots, tools/probes/tail_phases.hip).
    // A pivot that is not positive gives NaN / Inf from here on; `chk` collects that without a branch on the chain.
    double cb[n], lprev = 0.0;                       // the far entries of the previous pivot's column as read back from LDS
#pragma unroll
    for (int j = 0; j < n; ++j) cb[j] = 0.0;
    double inv = mqs::rsqrt_d(read_lane(row[0], 0));
    double chk = 0.0;                                // 0 * (1 / L_kk) summed over the pivots: NaN as soon as one pivot was not positive
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double lik = row[k] * inv;             // rows: L[lane][k] (lanes >= k);  inverse / rhs lanes: v_k
        row[k] = lik;
        chk = fma(inv, 0.0, chk);                    // rsq(0) = inf, rsq(< 0) = NaN: 0 * either is NaN
        if (k + 1 < n) {
            double *col = sCol + 64 * (k & 1);
            col[lane] = lik;
            asm volatile("" ::: "memory");           // the LDS pipe keeps a wavefront's accesses in order: no wait needed before the reads below
            row[k + 1] = fma(-lik, read_lane(lik, k + 1), row[k + 1]);
            if (k + 2 < n) row[k + 2] = fma(-lik, read_lane(lik, k + 2), row[k + 2]);
            const double akk = read_lane(row[k + 1], k + 1);
            double y = __builtin_amdgcn_rsq(akk);
            const double hd = 0.5 * akk;
            __builtin_amdgcn_sched_barrier(0);
            // Newton steps of the seed (six dependent instructions) with the previous column's updates between them
            constexpr int kStages = 6;
            double t = 0.0, e = 0.0;
#pragma unroll
            for (int st = 0; st < kStages; ++st) {
                if (st == 0) t = hd * y;
                else if (st == 1) e = fma(-t, y, 0.5);
                else if (st == 2) y = fma(y, e, y);
                else if (st == 3) t = hd * y;
                else if (st == 4) e = fma(-t, y, 0.5);
                else y = fma(y, e, y);
                if (k >= 1) {
#pragma unroll
                    for (int j = k + 2 + st; j < n; j += kStages) row[j] = fma(-lprev, cb[j], row[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            inv = y;
            lprev = lik;
            // this column's far entries: requested now that the previous column's registers are free (one buffer live at a time: the fused
            // tail runs at 128 registers), consumed in the next step's shadow
#pragma unroll
            for (int j = (k + 3) & ~1; j < n; j += 2) {
                // pairs (one ds_read_b128; also keeps the loop from being recognised as a memcpy, which lands in scratch memory)
                const double2 v = *reinterpret_cast<const double2 *>(col + j);
                if (j >= k + 3) cb[j] = v.x;
                cb[j + 1] = v.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    bad = !(chk == 0.0);                             // wave-uniform
    // lane 56 holds y = L^-1 b, lane 32 + c column c of L^-1:  x_c = sum_j (L^-1)[j][c] y_j
    if (isB) {
#pragma unroll
        for (int j = 0; j < n; ++j) sY[j] = row[j];
    }
    mqs_wave_lds_sync();
    double x = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) x = fma(row[j], sY[j], x);
    if (isI) sX[ci] = x;
    mqs_wave_lds_sync();
}

// retraction R <- R Exp(w), t <- t + R v of camera `c` (one lane per camera)
__device__ __forceinline__ void retract_pose_dev(const double *__restrict__ T, const double *d6, double *__restrict__ O)
{
    double E[9], w[3] = {d6[0], d6[1], d6[2]};
    so3_exp_dev(w, E);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O[3 * i + j] = T[3 * i] * E[j] + T[3 * i + 1] * E[3 + j] + T[3 * i + 2] * E[6 + j];
    for (int i = 0; i < 3; ++i) O[9 + i] = T[9 + i] + T[3 * i] * d6[3] + T[3 * i + 1] * d6[4] + T[3 * i + 2] * d6[5];
}

template <int C>
struct SolveLds {
    static constexpr int n = 6 * C;
    __attribute__((aligned(16))) double m[64 * n];          // the 64 lanes' starting vectors (build_solve_matrix)
    __attribute__((aligned(16))) double col[128];
    double lin[n * n + n + 2];
    double e[n], w[n], y[n], x[n];
    double info[2];
};

// diagonal of the solved system: the prior weight, then the damping (Marquardt scaling for lambda >= 0, |lambda| I below)
__device__ __forceinline__ double damped_diagonal(double v, double w, double lambda)
{
    return (lambda >= 0.0) ? (v + w) * (1.0 + lambda) : (v + w) - lambda;
}

// The starting vectors of the solve wave's 64 lanes, written by ALL threads of the workgroup (`nthreads`; the lone solve wave
// spent 240 of its ~1 700 instructions selecting them): lane r < n: row r of S with the prior weight and the damping on the
// diagonal; lane 32 + c: e_c; lane 56: g - prior residual; others 0.
template <int C>
__device__ __forceinline__ void build_solve_matrix(SolveLds<C> &sm, double lambda, int tid, int nthreads)
{
    constexpr int n = 6 * C;
    // four threads per lane vector, every fourth entry each (no division by n)
    for (int q = tid; q < 256; q += nthreads)
    for (int j = q & 3; j < n; j += 4) {
        const int lane = q >> 2, idx = lane * n + j;
        double v = 0.0;
        if (lane < n) {
            v = sm.lin[lane * n + j];
            if (j == lane) v = damped_diagonal(v, sm.w[lane], lambda);
        } else if (lane >= kSolveInvLane0 && lane < kSolveInvLane0 + n) {
            v = (j == lane - kSolveInvLane0) ? 1.0 : 0.0;
        } else if (lane == kSolveRhsLane) {
            v = sm.lin[n * n + j] - sm.e[j];
        }
        sm.m[idx] = v;
    }
}

// The same starting vectors straight from the registers the reduced system was loaded into (thread t holds entries t, t + 256,
// t + 512 of [S | g | cost | count]): the fused tail's path -- no LDS round trip of the system between the load and the solve.
// Same values as build_solve_matrix, entry for entry.
template <int C>
__device__ __forceinline__ void build_solve_matrix_from_registers(SolveLds<C> &sm, const double (&v)[3], double lambda, int tid)
{
    constexpr int n = 6 * C;
    static_assert(n * n + n + 2 <= 3 * kBlock, "three entries per thread hold the system");
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = tid + kBlock * q;
        if (k < n * n) {
            const int r = k / n, j = k - r * n;
            sm.m[k] = (j == r) ? damped_diagonal(v[q], sm.w[r], lambda) : v[q];
        } else if (k < n * n + n) {
            const int j = k - n * n;
            sm.m[kSolveRhsLane * n + j] = v[q] - sm.e[j];
        }
    }
    for (int idx = n * n + tid; idx < 64 * n; idx += kBlock) {          // lanes n .. 63 except the right-hand side's: constants
        const int lane = idx / n, j = idx - lane * n;
        if (lane != kSolveRhsLane) sm.m[idx] = (lane >= kSolveInvLane0 && lane < kSolveInvLane0 + n && j == lane - kSolveInvLane0) ? 1.0 : 0.0;
    }
}

// what the publishing workgroup's first wave does with the solution in sm.x
template <int C>
__device__ __forceinline__ void publish_solution(const SolveLds<C> &sm, const double *__restrict__ poses, bool bad, int lane,
                                                 double *__restrict__ dpose, double *__restrict__ poses_out,
                                                 double *__restrict__ info)
{
    constexpr int n = 6 * C;
    if (lane < n) dpose[lane] = sm.x[lane];
    if (poses_out && lane < C) retract_pose_dev(poses + 12 * lane, sm.x + 6 * lane, poses_out + 12 * lane);
    if (info && lane == 0) { info[0] = sm.info[0]; info[1] = bad ? 1.0 : 0.0; }
}

template <int C>
__global__ __launch_bounds__(64) void ba_solve_small_kernel(const double *__restrict__ lin, const double *__restrict__ poses,
                                                            const double *__restrict__ prior_poses,
                                                            const double *__restrict__ prior_sigmas,
                                                            const uint8_t *__restrict__ prior_mask, double lambda,
                                                            double *__restrict__ dpose, double *__restrict__ poses_out,
                                                            double *__restrict__ info)
{
    constexpr int n = 6 * C;
    __shared__ SolveLds<C> sm;
    const int lane = threadIdx.x;
    for (int k = lane; k < n * n + n + 2; k += 64) sm.lin[k] = lin[k];
    mqs_wave_lds_sync();
    pose_prior_terms<C>(poses, prior_poses, prior_sigmas, prior_mask, lane, sm.e, sm.w, sm.info);
    build_solve_matrix<C>(sm, lambda, lane, 64);
    mqs_wave_lds_sync();
    bool bad;
    reduced_solve_wave<C>(sm.m, sm.col, sm.y, sm.x, lane, bad);
    publish_solution<C>(sm, poses, bad, lane, dpose, poses_out, info);
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail of an iteration in ONE launch: reduced solve + pose retraction + landmark back-substitution.
// Every workgroup solves the 24 x 24 system itself (wave 0, ~2 us, from the reduced system every workgroup reads anyway)
// while its other waves have the camera blocks staged, then back-substitutes its contiguous range of landmarks; workgroup 0
// also publishes dpose, the retracted poses and info.  A persistent grid (<= 4 workgroups per CU) so that the solve is paid
// once per resident workgroup, ranges cut in rows of 64 landmarks so that the workgroups finish together.
// With a peer communicator (comm.hip) `lin` is this rank's receive buffer: `world` rows written by the ranks' finalize
// kernels over xGMI; the workgroup waits for the rows' flags and sums them in rank order -- the all-reduce of the iteration
// without a launch of its own.
// ---------------------------------------------------------------------------------------------------------------------
// The finalize inside the tail (one launch less per iteration: 4.3 us of 34 at 125 k landmarks).  The first kFinPieces * ceil(kRow / 64)
// workgroups of the tail first add one PIECE (an eighth) of the lineariser's partial rows for 64 slots each (piece_part_sum: the
// order of ba_finalize_kernel, the same bits), write the piece sums in [S | g | cost | count] numbering and raise a flag stamped
// with the launch's epoch; every workgroup then stages its cameras and prior terms, waits for the flags (bounded spin) and folds
// the pieces in order.  The finalizers have the lowest workgroup indices: they are resident before any workgroup that waits
// for them.  Quarter sums and flags are written and read with agent-scope atomic accesses (coherent per access across the
// XCDs' L2s; no fence: a fence costs a whole-L2 write-back / invalidate).  With a peer transport the finalizers also store
// their quarters into every rank's receive buffer and the wait is for every rank's flags: the all-reduce without any launch.
struct TailFin {
    const double *partials;           // null: no fused finalize
    int nrows;
    double *quarters;                 // the piece sums: [kFinPieces][kQuarterStride]
    unsigned long long *flags;        // [kFinPieces * slot groups]
    unsigned long long epoch;
    mqs_peer_push push;               // world = 0: single GPU
    int *status;                      // null, or the problem's host-visible status word: set when the wait for the finalizers gives up
    int withhold;                     // test hook: the finalizer piece that does not raise its flag (-1: none)
};
constexpr int kQuarterStride = MQS_PEER_QUARTER_STRIDE;

// How the wave lineariser and the fused tail share the caller's workspace (512 rows of kRow doubles, ws_doubles):
//   rows [0, grid)                                     the lineariser's partial rows, grid <= 256 * kWaveLinOcc
//   [256 rows, + 256 * kWlCamSlice<C>)                 the workgroups' camera blocks for the scalar loads (SCALAR form)
//   the last kFinPieces * kQuarterStride + 64 doubles  the fused finalize's piece sums and flags
// The three must not meet: checked per camera count; more than one workgroup per CU (the A/B build MQS_WL_OCC=2: 512 rows)
// would run the partial rows into the other two, so that build keeps to the LDS form and the finalize as a launch of its own.
template <int C>
struct WlWorkspaceCheck {
    static constexpr int64_t kRow = Layout<C>::kChunks * 32;
    static_assert((int64_t)256 * kWlCamSlice<C> + kFinPieces * kQuarterStride + 64 <= (int64_t)256 * kRow,
                  "camera slices + quarter sums + flags fit behind the 256 rows of partial sums");
    static_assert(kWaveLinOcc == 1 || MQS_WL_SCALAR_CAMS == 0, "two workgroups per CU: 512 partial rows leave no room for the camera slices");
    static constexpr bool ok = true;
};
static_assert(WlWorkspaceCheck<2>::ok && WlWorkspaceCheck<3>::ok && WlWorkspaceCheck<4>::ok, "workspace regions");

#ifndef MQS_TAIL_STAGED_OBS
#define MQS_TAIL_STAGED_OBS 1               // A/B: 0 = the measurements are loaded inside the rolled camera loop
#endif
template <int C>
__global__ __launch_bounds__(kBlock, 4) void ba_tail_kernel(
    const double *__restrict__ lin, mqs_peer_recv pr, TailFin fin, const double *__restrict__ poses, const double *__restrict__ calib,
    const double *__restrict__ sigma, const double *__restrict__ points, const double *__restrict__ obs,
    const uint8_t *__restrict__ mask, const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N,
    double lambda, const double *__restrict__ prior_poses, const double *__restrict__ prior_sigmas,
    const uint8_t *__restrict__ prior_mask, double *__restrict__ lin_out, double *__restrict__ dpose,
    double *__restrict__ poses_out, double *__restrict__ info, double *__restrict__ points_out)
{
    constexpr int n = 6 * C, nlin = n * n + n + 2;
    __shared__ double sCam[C * kCamStride];
    __shared__ SolveLds<C> sm;
    __shared__ int sTimedOut;          // a bounded wait of this workgroup gave up: nothing it waited for is read, nothing is published
    const int tid = threadIdx.x;
    if (tid == 0) sTimedOut = 0;
    __syncthreads();
    double v[3] = {0.0, 0.0, 0.0};
    constexpr int kRowT = Layout<C>::kChunks * 32, kSlotGroups = (kRowT + 63) / 64, kFinalizers = kFinPieces * kSlotGroups;
    if (fin.partials) {
        if ((int)blockIdx.x < kFinalizers) {
            __shared__ double sQ[4][64];
            const int sg = blockIdx.x % kSlotGroups, q = blockIdx.x / kSlotGroups, lane = tid & 63, wave = tid >> 6;
            const int slot = 64 * sg + lane;
            sQ[wave][lane] = (slot < kRowT) ? piece_part_sum(fin.partials, fin.nrows, kRowT, q, wave, slot) : 0.0;
            __syncthreads();
            if (wave == 0 && slot < Layout<C>::kSlots) {
                const double qs = ((sQ[0][lane] + sQ[1][lane]) + sQ[2][lane]) + sQ[3][lane];
                int o1, o2;
                slot_to_out<C>(slot, o1, o2);
                if (o1 >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o1, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o2 >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o2, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o1 >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o1, qs);
                if (o2 >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o2, qs);
            }
            mqs_stores_landed();
            __syncthreads();                          // the piece's entries have landed, here and in the peers
            if ((int)blockIdx.x != fin.withhold) {
                if (tid == 0) __hip_atomic_store(fin.flags + blockIdx.x, fin.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mqs::peer::publish_piece(fin.push, blockIdx.x, tid);  // lanes 0 .. world - 1 (none on a single GPU)
            }
        }
    } else if (pr.rows) {
        const bool arrived = mqs::peer::wait_and_sum(sm.lin, nlin, pr, tid, kBlock, &sTimedOut);
        if (arrived && blockIdx.x == gridDim.x - 1 && lin_out) {
            __syncthreads();
            for (int k = tid; k < nlin; k += kBlock) lin_out[k] = sm.lin[k];
        }
    } else {
        // the reduced system straight into registers: the loads are in flight while the camera blocks are staged and the
        // prior terms computed
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = (tid + kBlock * q < nlin) ? lin[tid + kBlock * q] : 0.0;
    }
    stage_cams<C>(poses, calib, sigma, sCam, tid);                  // ends in a workgroup barrier
    if (tid < 64) pose_prior_terms<C>(poses, prior_poses, prior_sigmas, prior_mask, tid, sm.e, sm.w, sm.info);
    __syncthreads();
    if (fin.partials) {
        // the quarters of this launch (and, over the peer transport, of every rank's): wait, then add in the fixed order
        if (fin.push.world > 0) {
            mqs::peer::wait_flags(pr, tid, kBlock, &sTimedOut);
        } else {
            // Forward progress rests on the finalizers (the lowest workgroup indices) being dispatched before the workgroups that
            // wait for them fill the chip: in-order dispatch, which the hardware does and HIP does not promise.  Hence the bound,
            // and a wait that gives up is an ERROR the host sees (status word, info[1] = 2), never a sum of stale quarters.
            if (tid < kFinalizers) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(fin.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fin.epoch) {
                    if (wall_clock64() - t0 > mqs::peer::kSpinTicks) {
                        sTimedOut = 1;
                        if (fin.status) __hip_atomic_store(fin.status, MQS_STATUS_FINALIZE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
    }
    if (sTimedOut) {                                              // workgroup-uniform (written before the barriers above)
        if (blockIdx.x == gridDim.x - 1 && tid == 0 && info) info[1] = 2.0;
        return;
    }
    if (fin.partials) {
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            const int k = tid + kBlock * q3;
            if (k < nlin) {
                if (fin.push.world > 0) {
                    double t = 0.0;
                    for (int rk = 0; rk < pr.world; ++rk) {            // rank order; a rank's pieces folded in the finalize's order
                        const double *row = pr.rows + (size_t)rk * pr.row_stride + k;
                        double pv[kFinPieces];
#pragma unroll
                        for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(row + q * kQuarterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        double r = pv[0];
#pragma unroll
                        for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                        t += r;
                    }
                    v[q3] = t;
                } else {
                    double pv[kFinPieces];
#pragma unroll
                    for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(fin.quarters + q * kQuarterStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    double r = pv[0];
#pragma unroll
                    for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                    v[q3] = r;
                }
                if (blockIdx.x == gridDim.x - 1 && lin_out) lin_out[k] = v[q3];
            }
        }
        build_solve_matrix_from_registers<C>(sm, v, lambda, tid);
    } else if (pr.rows) {
        build_solve_matrix<C>(sm, lambda, tid, kBlock);
    } else {
        build_solve_matrix_from_registers<C>(sm, v, lambda, tid);
    }
    __syncthreads();
    // the LAST workgroup of the grid has no landmarks: it publishes dpose, the retracted poses and info (the trigonometry of
    // the retraction, ~1 us on one wave, would otherwise sit on the path of a workgroup that also back-substitutes)
    const bool publisher = blockIdx.x == gridDim.x - 1;
    if (tid < 64) {
        bool bad;
        reduced_solve_wave<C>(sm.m, sm.col, sm.y, sm.x, tid, bad);
        if (publisher) publish_solution<C>(sm, poses, bad, tid, dpose, poses_out, info);
    }
    if (publisher) return;
    __syncthreads();

    // the staging area of the measurements: the solve's starting vectors, column buffer and system (dead behind the barrier above)
    static_assert(sizeof(double2) * C * kBlock <= sizeof(sm.m) + sizeof(sm.col) + sizeof(sm.lin), "staged measurements fit the solve's LDS");
    double2 *sObs = reinterpret_cast<double2 *>(sm.m);
    const double2 *o2 = reinterpret_cast<const double2 *>(obs);
    const int wave64 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int64_t rows64 = (N + 63) / 64;
    const int64_t nwg = gridDim.x - 1;
    const int64_t r_begin = rows64 * blockIdx.x / nwg, r_end = rows64 * (blockIdx.x + 1) / nwg;
    int64_t end = r_end * 64;
    if (end > N) end = N;
    for (int64_t base = r_begin * 64; base < end; base += kBlock) {
        const int64_t i = base + tid;
        const bool live = i < end;
        const int64_t ii = live ? i : 0;
        double px = points[3 * ii + 0], py = points[3 * ii + 1], pz = points[3 * ii + 2];
        if (!live) { px = 0.0; py = 0.0; pz = 0.0; }
#if MQS_TAIL_STAGED_OBS
        // the landmark's C measurements: one LDS-DMA load per camera, all in flight together with the point's loads above (a
        // measurement loaded inside the rolled camera loop costs a memory latency per camera: one 1 KB load in flight per wave)
#pragma unroll
        for (int c = 0; c < C; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(o2 + (int64_t)c * N + ii),
                                             (__attribute__((address_space(3))) void *)(sObs + c * kBlock + wave64), 16, 0, 0);
        unsigned mbits = 0u;
        if (mask) {
#pragma unroll
            for (int c = 0; c < C; ++c) mbits |= (unsigned)mask[(int64_t)c * N + ii] << (8 * c);
        }
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the DMA has landed in LDS (and everything else has arrived)
        const StagedObs ob = {sObs + tid, mbits, live, mask != nullptr};
#else
        const DevObs ob = {reinterpret_cast<const double2 *>(obs), mask, i, N, live};
        double pw, dx, dy, dz;
        load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
#endif
        const mqs::Vec3 dp = landmark_backsub<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, sm.x);
        if (live) {
            points_out[3 * i + 0] = px + dp.x;
            points_out[3 * i + 1] = py + dp.y;
            points_out[3 * i + 2] = pz + dp.z;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail, second form (round 4): ONE workgroup of 12 waves per CU instead of four of 4 waves.
//
// What the form above pays at 1e6 x 4 (profiles/r04/02: 13.6 M vector instructions per launch, the SIMDs' vector units 74 % busy
// summed over their four waves -- the launch is bound by vector issue as much as by latency): every one of its 1 024 workgroups
// builds and solves the 24 x 24 system (~2 500 wave-instructions each: 2.6 M of the 13.6 M, and 4-5 us at the head of every
// workgroup during which it streams nothing), and a batch's loads are issued, waited for and only then computed on.  Here
//   * 256 workgroups (one per CU) of 768 threads: the system is solved ONCE per CU, by wave 0, while every wave already has its
//     first batch's loads in flight;
//   * the measurements of batch b + 1 (LDS-DMA into the other half of a two-deep ring) and its points / prior weights / mask
//     bytes (registers) are requested BEFORE batch b's arithmetic, so a wave never sits out a memory latency after the first;
//   * the finalize pieces (fused finalize) are folded by the first 16 workgroups, three pieces each, in the order of
//     ba_finalize_kernel: the same bits.
// Same arithmetic per landmark (landmark_backsub), same solve (reduced_solve_wave): bit-identical outputs to the first form.
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kTail2Block) void ba_tail2_kernel(
    const double *__restrict__ lin, mqs_peer_recv pr, TailFin fin, const double *__restrict__ poses, const double *__restrict__ calib,
    const double *__restrict__ sigma, const double *__restrict__ points, const double *__restrict__ obs,
    const uint8_t *__restrict__ mask, const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N,
    double lambda, const double *__restrict__ prior_poses, const double *__restrict__ prior_sigmas,
    const uint8_t *__restrict__ prior_mask, double *__restrict__ lin_out, double *__restrict__ dpose,
    double *__restrict__ poses_out, double *__restrict__ info, double *__restrict__ points_out)
{
    constexpr int n = 6 * C, nlin = n * n + n + 2;
    constexpr int TB = kTail2Block;
    __shared__ double sCam[C * kCamStride];
    __shared__ SolveLds<C> sm;
    __shared__ int sTimedOut;
    extern __shared__ __attribute__((aligned(16))) unsigned char tail2_smem[];
    double2 *sRing = reinterpret_cast<double2 *>(tail2_smem);           // [2][C][TB] measurements: this batch's and the next one's
    const int tid = threadIdx.x;
    if (tid == 0) sTimedOut = 0;
    __syncthreads();

    // ---- this workgroup's landmarks, and the first batch's loads (in flight under everything up to the solve) ----
    // the last workgroup also publishes dpose, the retracted poses and info: its wave 0 alone, ~1 us behind the other fifteen
    // (the loop below has no workgroup barrier, nobody waits for it)
    const bool publisher = blockIdx.x == gridDim.x - 1;
    const double2 *o2 = reinterpret_cast<const double2 *>(obs);
    const int wave64 = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int64_t rows64 = (N + 63) / 64;
    const int64_t nwg = gridDim.x;
    const int64_t r_begin = rows64 * blockIdx.x / nwg, r_end = rows64 * (blockIdx.x + 1) / nwg;
    int64_t end = r_end * 64;
    if (end > N) end = N;
    struct Batch { double px, py, pz, pw; unsigned mb[C]; };          // mask bytes kept apart until used: combining them is a wait
    auto request = [&](int64_t base, int slot, Batch &b) {
        const int64_t i = base + tid;
        const int64_t ii = i < end ? i : (end > 0 ? end - 1 : 0);
        b.px = points[3 * ii + 0]; b.py = points[3 * ii + 1]; b.pz = points[3 * ii + 2];
#pragma unroll
        for (int c = 0; c < C; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(o2 + (int64_t)c * N + ii),
                                             (__attribute__((address_space(3))) void *)(sRing + (slot * C + c) * TB + wave64), 16, 0, 0);
#pragma unroll
        for (int c = 0; c < C; ++c) b.mb[c] = mask ? (unsigned)mask[(int64_t)c * N + ii] : 1u;
        b.pw = prior_w ? prior_w[ii] : 0.0;
    };
    Batch cur;
    cur.px = cur.py = cur.pz = cur.pw = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) cur.mb[c] = 0u;
    const int64_t base0 = r_begin * 64;
    if (base0 < end) request(base0, 0, cur);
    // The first batch's registers are PARKED in LDS across the solve (ring half 1 and beyond: unused before the loop): the solve
    // wave needs all 128 registers, and what is live across it would go to scratch memory -- on the serial chain of every CU.
    // Parking waits for the loads; that costs nothing: every wave has to meet the barrier in front of the solve anyway, and by
    // then (finalizer pieces, flags) the loads have long landed.
    double *sPark = reinterpret_cast<double *>(sRing + C * TB);          // [4][TB] doubles, then [C][TB] unsigned
    unsigned *sParkM = reinterpret_cast<unsigned *>(sPark + 4 * TB);
    auto park = [&]() {
        sPark[tid] = cur.px; sPark[TB + tid] = cur.py; sPark[2 * TB + tid] = cur.pz; sPark[3 * TB + tid] = cur.pw;
#pragma unroll
        for (int c = 0; c < C; ++c) sParkM[c * TB + tid] = cur.mb[c];
    };
    auto unpark = [&]() {
        cur.px = sPark[tid]; cur.py = sPark[TB + tid]; cur.pz = sPark[2 * TB + tid]; cur.pw = sPark[3 * TB + tid];
#pragma unroll
        for (int c = 0; c < C; ++c) cur.mb[c] = sParkM[c * TB + tid];
    };

    // ---- the reduced system: finalize pieces / the peers' rows / the caller's `lin` ----
    double v[3] = {0.0, 0.0, 0.0};
    constexpr int kRowT = Layout<C>::kChunks * 32, kSlotGroups = (kRowT + 63) / 64, kFinalizers = kFinPieces * kSlotGroups;
    constexpr int kVirt = TB / kBlock;                                    // finalizer pieces one workgroup folds (four waves each)
    if (fin.partials) {
        const int vb = (int)blockIdx.x * kVirt + tid / kBlock;           // the piece, numbered as the first form's workgroups
        if ((int)blockIdx.x * kVirt < kFinalizers) {                      // workgroup-uniform
            double (*sQ)[4][64] = reinterpret_cast<double (*)[4][64]>(sRing + C * TB);   // ring slot 1: not in use before the loop
            static_assert(sizeof(double) * kVirt * 4 * 64 <= sizeof(double2) * C * TB, "the piece sums fit a ring slot");
            const int t4 = tid % kBlock, lane = t4 & 63, wave = t4 >> 6, vq = tid / kBlock;
            const bool mine = vb < kFinalizers;
            const int sg = vb % kSlotGroups, q = vb / kSlotGroups;
            const int slot = 64 * sg + lane;
            sQ[vq][wave][lane] = (mine && slot < kRowT) ? piece_part_sum(fin.partials, fin.nrows, kRowT, q, wave, slot) : 0.0;
            __syncthreads();
            if (mine && wave == 0 && slot < Layout<C>::kSlots) {
                const double qs = ((sQ[vq][0][lane] + sQ[vq][1][lane]) + sQ[vq][2][lane]) + sQ[vq][3][lane];
                int o1, o2i;
                slot_to_out<C>(slot, o1, o2i);
                if (o1 >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o1, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o2i >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o2i, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o1 >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o1, qs);
                if (o2i >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o2i, qs);
            }
            mqs_stores_landed();          // (also drains this wave's first-batch loads: the finalizer workgroups are 12 of 256)
            __syncthreads();                          // the pieces' entries have landed, here and in the peers
            if (mine && vb != fin.withhold) {
                if (t4 == 0) __hip_atomic_store(fin.flags + vb, fin.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mqs::peer::publish_piece(fin.push, vb, t4);          // lanes 0 .. world - 1 of the piece's first wave
            }
        }
    } else if (pr.rows) {
        const bool arrived = mqs::peer::wait_and_sum(sm.lin, nlin, pr, tid, TB, &sTimedOut);
        if (arrived && publisher && lin_out) {
            __syncthreads();
            for (int k = tid; k < nlin; k += TB) lin_out[k] = sm.lin[k];
        }
    } else if (tid < kBlock) {
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = (tid + kBlock * q < nlin) ? lin[tid + kBlock * q] : 0.0;
    }
    stage_cams<C>(poses, calib, sigma, sCam, tid);                  // ends in a workgroup barrier
    if (tid < 64) pose_prior_terms<C>(poses, prior_poses, prior_sigmas, prior_mask, tid, sm.e, sm.w, sm.info);
    __syncthreads();
    if (fin.partials) {
        if (fin.push.world > 0) {
            mqs::peer::wait_flags(pr, tid, TB, &sTimedOut);
        } else {
            // (forward progress: the finalizers are the lowest workgroup indices of a grid of one workgroup per CU -- all resident)
            if (tid < kFinalizers) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(fin.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fin.epoch) {
                    if (wall_clock64() - t0 > mqs::peer::kSpinTicks) {
                        sTimedOut = 1;
                        if (fin.status) __hip_atomic_store(fin.status, MQS_STATUS_FINALIZE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
    }
    if (sTimedOut) {                                              // workgroup-uniform: nothing waited for is read, nothing is published
        if (publisher && tid == 0 && info) info[1] = 2.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first batch's LDS-DMA must not land in a released LDS
        return;
    }
    if (fin.partials) {
        if (tid < kBlock) {
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) {
                const int k = tid + kBlock * q3;
                if (k < nlin) {
                    if (fin.push.world > 0) {
                        double t = 0.0;
                        for (int rk = 0; rk < pr.world; ++rk) {            // rank order; a rank's pieces folded in the finalize's order
                            const double *row = pr.rows + (size_t)rk * pr.row_stride + k;
                            double pv[kFinPieces];
#pragma unroll
                            for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(row + q * kQuarterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            double r = pv[0];
#pragma unroll
                            for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                            t += r;
                        }
                        v[q3] = t;
                    } else {
                        double pv[kFinPieces];
#pragma unroll
                        for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(fin.quarters + q * kQuarterStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        double r = pv[0];
#pragma unroll
                        for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                        v[q3] = r;
                    }
                    if (publisher && lin_out) lin_out[k] = v[q3];
                }
            }
            build_solve_matrix_from_registers<C>(sm, v, lambda, tid);
        }
    } else if (pr.rows) {
        build_solve_matrix<C>(sm, lambda, tid, TB);
    } else if (tid < kBlock) {
        build_solve_matrix_from_registers<C>(sm, v, lambda, tid);
    }
    park();
    __syncthreads();
    bool bad = false;
    if (tid < 64) reduced_solve_wave<C>(sm.m, sm.col, sm.y, sm.x, tid, bad);
    __syncthreads();
    unpark();
    // Every wave has its parked batch back BEFORE any wave requests the next one: the parking area IS ring half 1, the half the
    // first request of the loop lands in, and one wave's slice of the ring is another wave's parked registers (the publisher's
    // wave 0, a microsecond late out of the retraction, used to find its points overwritten: tools/probes/tail_form_diff.py).
    __syncthreads();
    if (publisher && tid < 64) publish_solution<C>(sm, poses, bad, tid, dpose, poses_out, info);      // behind the barriers: nobody waits for the retraction

    // ---- back-substitution: batch b's arithmetic with batch b + 1's loads in flight ----
    int slot = 0;
    for (int64_t base = base0; base < end; base += TB, slot ^= 1) {
        Batch nxt = cur;
        const bool more = base + TB < end;                       // workgroup-uniform
        // this batch's loads have landed (DMA in LDS, registers); only then is the next batch requested: vmcnt counts in order,
        // and the ring half the next batch lands in was read by the batch before this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) request(base + TB, slot ^ 1, nxt);
        const int64_t i = base + tid;
        const bool live = i < end;
        double px = cur.px, py = cur.py, pz = cur.pz;
        if (!live) { px = 0.0; py = 0.0; pz = 0.0; }
        double pw = 0.0, dx = 0.0, dy = 0.0, dz = 0.0;
        if (live && prior_w && cur.pw > 0.0) {
            pw = cur.pw;
            dx = px - prior_xyz[3 * i + 0];
            dy = py - prior_xyz[3 * i + 1];
            dz = pz - prior_xyz[3 * i + 2];
        }
        unsigned mbits = 0u;
#pragma unroll
        for (int c = 0; c < C; ++c) mbits |= cur.mb[c] << (8 * c);
        const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) double2 *)(sRing + slot * C * TB + tid);
        const StagedObs2 ob = {ring_addr, mbits, live, mask != nullptr};
        const mqs::Vec3 dp = landmark_backsub<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, sm.x);
        if (live) {
            points_out[3 * i + 0] = px + dp.x;
            points_out[3 * i + 1] = py + dp.y;
            points_out[3 * i + 2] = pz + dp.z;
        }
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Back-to-back Gauss-Newton iterations (round 4): the TAIL of iteration k and the LINEARISER of iteration k + 1 in ONE launch.
//
// An iteration by itself is two launches -- lineariser, tail -- and on a shard the size one rank of BASELINE configs[3] holds
// (125 k landmarks) their boundaries and prologues are a fifth of it: 15.7 + 10.3 us of kernels, 31.9 us per iteration.  Inside a
// run of iterations (mqs_ba_gn_iterations_dev) nothing needs the host between the back-substitution of one iteration and the
// linearisation of the next, and both walk the same landmarks: here every wave
//   1. takes part in the tail's head exactly as ba_tail_kernel does (finalizer pieces of the previous launch's partial rows in
//      the first workgroups, flags, the pieces folded in order, the 6C x 6C solve in wave 0 of every workgroup; over the peer
//      transport the pieces also go to every rank and the wait is for all ranks' flags),
//   2. back-substitutes ITS rows of 64 landmarks (the lineariser's partition: the wave that linearises a row next is the wave
//      that has just written it) into the other landmark buffer,
//   3. retracts the poses itself (wave 0; the last workgroup also publishes dpose, poses, info), stages the new camera blocks,
//   4. linearises its rows at the new estimate (wl_chunk, LDS form) and leaves the workgroup's partial row for the next launch.
// A run of K iterations is lineariser, K - 1 of these, tail: K + 1 launches instead of 2 K.  Same arithmetic, same summation
// orders as the two-launch iteration: bit-identical estimates (tests/test_ba_gpu.py).  One workgroup per CU (the lineariser's
// geometry and LDS); shards below kWlScalarMinLandmarks landmarks (beyond, the lineariser's scalar-load form and the
// four-waves-per-SIMD tail are the faster pair, and the launch boundary is 2 % of an iteration).
// ---------------------------------------------------------------------------------------------------------------------
#if defined(MQS_ITERATE_PROBE)          // timing probe builds only (tools/probes/iterate_phases.py): 100 MHz wall-clock stamps per workgroup and phase
__device__ long long g_it_probe[8][256];
#define MQS_IT_STAMP(k) { if (threadIdx.x == 0) g_it_probe[k][blockIdx.x & 255] = wall_clock64(); }
extern "C" int mqs_debug_iterate_probe(long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_it_probe), sizeof(g_it_probe)) == hipSuccess ? 0 : -1;
}
#else
#define MQS_IT_STAMP(k) {}
#endif
template <int C>
__global__ __launch_bounds__(kBlock, 1) void ba_iterate_kernel(
    mqs_peer_recv pr, TailFin fin, const double *__restrict__ poses, const double *__restrict__ calib, const double *__restrict__ sigma,
    const double *__restrict__ points, const double *__restrict__ obs, const uint8_t *__restrict__ mask,
    const double *__restrict__ prior_w, const double *__restrict__ prior_xyz, int64_t N, double lambda,
    const double *__restrict__ prior_poses, const double *__restrict__ prior_sigmas, const uint8_t *__restrict__ prior_mask,
    double *__restrict__ lin_out, double *__restrict__ dpose, double *__restrict__ poses_out, double *__restrict__ info,
    double *__restrict__ points_out, double *__restrict__ partials_out)
{
    using L = Layout<C>;
    constexpr int n = 6 * C, nlin = n * n + n + 2;
    constexpr int kRow = L::kChunks * 32, kSlotGroups = (kRow + 63) / 64, kFinalizers = kFinPieces * kSlotGroups;
    __shared__ double sCam[C * kCamStride];
    __shared__ double sPoseNew[C * 12];
    __shared__ int sTimedOut;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_smem[];
    // the solve's LDS and the finalizers' piece sums lie over the lineariser's stash: dead before the first chunk is linearised
    SolveLds<C> &sm = *reinterpret_cast<SolveLds<C> *>(wl_smem);
    static_assert(sizeof(SolveLds<C>) + 4 * 64 * sizeof(double) <= sizeof(double2) * kWaveLinLdsL * C * 3 * kBlock, "the solve fits the stash");
    double (*sQ)[64] = reinterpret_cast<double (*)[64]>(wl_smem + ((sizeof(SolveLds<C>) + 255) & ~size_t(255)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sTimedOut = 0;
    __syncthreads();
    MQS_IT_STAMP(0)

    // ---- 1. the tail's head (ba_tail_kernel) ----
    if ((int)blockIdx.x < kFinalizers) {
        const int sg = blockIdx.x % kSlotGroups, q = blockIdx.x / kSlotGroups;
        const int slot = 64 * sg + lane;
        sQ[wave][lane] = (slot < kRow) ? piece_part_sum(fin.partials, fin.nrows, kRow, q, wave, slot) : 0.0;
        __syncthreads();
        if (wave == 0 && slot < L::kSlots) {
            const double qs = ((sQ[0][lane] + sQ[1][lane]) + sQ[2][lane]) + sQ[3][lane];
            int o1, o2;
            slot_to_out<C>(slot, o1, o2);
            if (o1 >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o1, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 >= 0) __hip_atomic_store(fin.quarters + q * kQuarterStride + o2, qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o1 >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o1, qs);
            if (o2 >= 0) mqs::peer::push_entry(fin.push, q * kQuarterStride + o2, qs);
        }
        mqs_stores_landed();
        __syncthreads();                          // the piece's entries have landed, here and in the peers
        if ((int)blockIdx.x != fin.withhold) {
            if (tid == 0) __hip_atomic_store(fin.flags + blockIdx.x, fin.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mqs::peer::publish_piece(fin.push, blockIdx.x, tid);
        }
    }
    MQS_IT_STAMP(1)
    stage_cams<C>(poses, calib, sigma, sCam, tid);                  // ends in a workgroup barrier
    if (tid < 64) pose_prior_terms<C>(poses, prior_poses, prior_sigmas, prior_mask, tid, sm.e, sm.w, sm.info);
    __syncthreads();
    if (fin.push.world > 0) {
        mqs::peer::wait_flags(pr, tid, kBlock, &sTimedOut);
    } else {
        if (tid < kFinalizers) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(fin.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fin.epoch) {
                if (wall_clock64() - t0 > mqs::peer::kSpinTicks) {
                    sTimedOut = 1;
                    if (fin.status) __hip_atomic_store(fin.status, MQS_STATUS_FINALIZE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    MQS_IT_STAMP(2)
    const bool publisher = blockIdx.x == gridDim.x - 1;
    if (sTimedOut) {                                              // nothing waited for is read, nothing is published
        if (publisher && tid == 0 && info) info[1] = 2.0;
        return;
    }
    {
        double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            const int k = tid + kBlock * q3;
            if (k < nlin) {
                if (fin.push.world > 0) {
                    double t = 0.0;
                    for (int rk = 0; rk < pr.world; ++rk) {            // rank order; a rank's pieces folded in the finalize's order
                        const double *row = pr.rows + (size_t)rk * pr.row_stride + k;
                        double pv[kFinPieces];
#pragma unroll
                        for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(row + q * kQuarterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        double r = pv[0];
#pragma unroll
                        for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                        t += r;
                    }
                    v[q3] = t;
                } else {
                    double pv[kFinPieces];
#pragma unroll
                    for (int q = 0; q < kFinPieces; ++q) pv[q] = __hip_atomic_load(fin.quarters + q * kQuarterStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    double r = pv[0];
#pragma unroll
                    for (int q = 1; q < kFinPieces; ++q) r += pv[q];
                    v[q3] = r;
                }
                if (publisher && lin_out) lin_out[k] = v[q3];
            }
        }
        build_solve_matrix_from_registers<C>(sm, v, lambda, tid);
    }
    __syncthreads();
    bool bad = false;
    if (tid < 64) reduced_solve_wave<C>(sm.m, sm.col, sm.y, sm.x, tid, bad);
    __syncthreads();
    MQS_IT_STAMP(3)
    // the retraction (wave 0, one lane per camera): into LDS for this workgroup's next linearisation; the last workgroup publishes
    if (tid < C) retract_pose_dev(poses + 12 * tid, sm.x + 6 * tid, sPoseNew + 12 * tid);
    if (publisher && tid < 64) publish_solution<C>(sm, poses, bad, tid, dpose, poses_out, info);

    // ---- 2. back-substitution of this wave's rows (the lineariser's partition) ----
    // A wave alone on its SIMD hides no latency: the measurements of up to kAhead of its rows are requested at once by LDS-DMA
    // (one 1 KB piece per camera and row into the wave's own slice of the stash area, beyond the solve's LDS), the next row's
    // point while the current row's arithmetic issues.  (With the measurements loaded inside the rolled camera loop the launch
    // took as long as the two it replaces: four dependent HBM latencies per row.)
    const int64_t rows = (N + 63) / 64;
    const int64_t nw = (int64_t)gridDim.x * kWaves, gw = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t r_begin = rows * gw / nw, r_end = rows * (gw + 1) / nw;
    {
        constexpr size_t kStageOff = 24 * 1024;
        constexpr size_t kStageRoom = sizeof(double2) * kWaveLinLdsL * C * 3 * kBlock + sizeof(double) * kWaves * 352 - kStageOff;
        constexpr int kAheadFit = (int)(kStageRoom / (sizeof(double2) * kWaves * C * 64));
        constexpr int kAhead = kAheadFit >= 7 ? 7 : (kAheadFit >= 1 ? kAheadFit : 1);    // seven rows ahead (A/B builds with a smaller stash: what fits)
        static_assert(sizeof(SolveLds<C>) + 4 * 64 * sizeof(double) + 256 <= kStageOff, "the staged measurements lie beyond the solve's LDS");
        static_assert(kStageOff + sizeof(double2) * kWaves * kAhead * C * 64 <= sizeof(double2) * kWaveLinLdsL * C * 3 * kBlock + sizeof(double) * kWaves * 352,
                      "and inside the launch's dynamic LDS (the stash and the rows of totals behind it: zeroed only when the linearisation starts)");
        double2 *sObs = reinterpret_cast<double2 *>(wl_smem + kStageOff) + (size_t)wave * kAhead * C * 64;      // [kAhead][C][64]
        const double2 *o2 = reinterpret_cast<const double2 *>(obs);
        for (int64_t rg = r_begin; rg < r_end; rg += kAhead) {
            const int nr = (int)((r_end - rg) < kAhead ? (r_end - rg) : kAhead);
            for (int k = 0; k < nr; ++k) {
                const int64_t i = (rg + k) * 64 + lane;
                const int64_t ii = i < N ? i : N - 1;
#pragma unroll
                for (int c = 0; c < C; ++c)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(o2 + (int64_t)c * N + ii),
                                                     (__attribute__((address_space(3))) void *)(sObs + (k * C + c) * 64), 16, 0, 0);
            }
            int64_t i0 = rg * 64 + lane;
            int64_t ii0 = i0 < N ? i0 : N - 1;
            double npx = points[3 * ii0 + 0], npy = points[3 * ii0 + 1], npz = points[3 * ii0 + 2];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the measurements of these rows are in LDS, the first point has arrived
            for (int k = 0; k < nr; ++k) {
                const int64_t i = (rg + k) * 64 + lane;
                const bool live = i < N;
                double px = npx, py = npy, pz = npz;
                if (k + 1 < nr) {                                    // the next row's point, in flight under this row's arithmetic
                    const int64_t j = (rg + k + 1) * 64 + lane;
                    const int64_t jj = j < N ? j : N - 1;
                    npx = points[3 * jj + 0]; npy = points[3 * jj + 1]; npz = points[3 * jj + 2];
                }
                if (!live) { px = 0.0; py = 0.0; pz = 0.0; }
                unsigned mbits = 0u;
                if (mask) {
                    const int64_t ii = live ? i : 0;
#pragma unroll
                    for (int c = 0; c < C; ++c) mbits |= (unsigned)mask[(int64_t)c * N + ii] << (8 * c);
                }
                double pw, dx, dy, dz;
                load_prior(prior_w, prior_xyz, i, live, px, py, pz, pw, dx, dy, dz);
                const StagedObs64 ob = {sObs + k * C * 64 + lane, mbits, live, mask != nullptr};
                const mqs::Vec3 dp = landmark_backsub<C>(sCam, ob, px, py, pz, pw, dx, dy, dz, lambda, sm.x);
                if (live) {
                    points_out[3 * i + 0] = px + dp.x;
                    points_out[3 * i + 1] = py + dp.y;
                    points_out[3 * i + 2] = pz + dp.z;
                }
            }
        }
    }
    __syncthreads();                                  // every wave is done with the old camera blocks and with dpose (sm.x lies in the stash)
    MQS_IT_STAMP(4)

    // ---- 3. the new camera blocks ----
    if (tid < C) stage_camera(sCam + kCamStride * tid, sPoseNew + 12 * tid, calib + 9 * tid, sigma[tid]);
    __syncthreads();

    MQS_IT_STAMP(5)
    // ---- 4. linearisation of the same rows at the new estimate (ba_linearize_wave_kernel, LDS form) ----
    double2 *sStash = reinterpret_cast<double2 *>(wl_smem);
    double *sTot = reinterpret_cast<double *>(wl_smem + sizeof(double2) * kWaveLinLdsL * C * 3 * kBlock);   // [kWaves][kRow]
    double *tot = sTot + wave * kRow;
    for (int k = lane; k < kRow; k += 64) tot[k] = 0.0;
    mqs_wave_lds_sync();
    const WlStash stash = {sStash + tid};
    const double2 *obs2 = reinterpret_cast<const double2 *>(obs);
    const double *cams = sCam;
    bool nodist = true;
#pragma unroll
    for (int c = 0; c < C; ++c) nodist = nodist && camera_without_distortion(sCam + kCamStride * c);
    for (int64_t r = r_begin; r < r_end;) {
        const int64_t left = r_end - r;
        const int nl = left >= kWaveLinMaxL ? kWaveLinMaxL : (int)left;
#define MQS_WL_CALL(LL)                                                                                                       \
    do {                                                                                                                      \
        if (nodist) wl_chunk<C, LL, true>(cams, stash, points_out, obs2, mask, prior_w, prior_xyz, N, lambda, r, lane, tot);  \
        else wl_chunk<C, LL, false>(cams, stash, points_out, obs2, mask, prior_w, prior_xyz, N, lambda, r, lane, tot);        \
    } while (0)
#if MQS_WL_MAXL >= 4
        if (nl == 4) MQS_WL_CALL(4); else
#endif
#if MQS_WL_MAXL >= 3
        if (nl == 3) MQS_WL_CALL(3); else
#endif
        if (nl == 2) MQS_WL_CALL(2);
        else MQS_WL_CALL(1);
#undef MQS_WL_CALL
        r += nl;
    }
    __syncthreads();
    MQS_IT_STAMP(6)
    for (int s = tid; s < kRow; s += kBlock) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += sTot[w * kRow + s];
        partials_out[(int64_t)blockIdx.x * kRow + s] = t;
    }
    MQS_IT_STAMP(7)
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

// A/B switch for measurements: MQS_BA_LINEARIZER=lane selects the one-landmark-per-lane kernel for every C
bool wave_lineariser_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MQS_BA_LINEARIZER");
        v = (e && strcmp(e, "lane") == 0) ? 0 : 1;
    }
    return v == 1;
}

// A/B switch for measurements: MQS_BA_TAIL_FORM=2 selects the second form of the fused tail (one twelve-wave workgroup per CU: built
// in round 4, bit-identical, and SLOWER -- 38.9 against 36.1 us at 1e6, 11.5 against 10.3 us at 125 k, profiles/r04/05 -- because
// 3 072 resident waves quantise 15 625 rows worse than 4 092 do and three waves per SIMD hide less than four)
int tail_form()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MQS_BA_TAIL_FORM");
        v = (e && e[0] == '2') ? 2 : 1;
    }
    return v;
}

// A/B switch for measurements: MQS_BA_TAIL=split issues solve and back-substitution as two launches for every C
bool tail_fusion_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MQS_BA_TAIL");
        v = (e && strcmp(e, "split") == 0) ? 0 : 1;
    }
    return v == 1;
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

// parts: bit 0 = the lineariser kernel, bit 1 = the finalize kernel (mqs_ba_time_dev times them one by one)
static int ba_linearize_parts(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                              const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                              int64_t N, double lambda, double *out, void *workspace, int64_t workspace_bytes, void *stream_,
                              int parts, const mqs_peer_push *push = nullptr);

int mqs_ba_linearize_dev(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                         const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                         int64_t N, double lambda, double *out, void *workspace, int64_t workspace_bytes, void *stream_)
{
    return ba_linearize_parts(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, out, workspace,
                              workspace_bytes, stream_, 3);
}

static int ba_linearize_parts(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                              const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                              int64_t N, double lambda, double *out, void *workspace, int64_t workspace_bytes, void *stream_,
                              int parts, const mqs_peer_push *push)
{
    int rc = check_common(poses, calib, sigma, C, points, obs, N);
    if (rc != MQS_OK) return rc;
    mqs_peer_push pp = {};
    if (push) pp = *push;
    MQS_ARG_CHECK(workspace != nullptr && ((parts & 2) == 0 || out != nullptr), "out and workspace must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_ba_workspace_bytes(C, N), "workspace too small (mqs_ba_workspace_bytes)");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *partials = static_cast<double *>(workspace);
    {
        static const bool skip_fin = getenv("MQS_EXPERIMENT_SKIP_FINALIZE") != nullptr;      // timing experiment only (wrong results)
        if (skip_fin) parts &= ~2;
    }
    if (C >= 2 && C <= 4 && wave_lineariser_enabled()) {
        // one workgroup per CU (its LDS holds the lanes' columns), as many as there are 64-landmark rows to hand out
        const int64_t rows = (N + 63) / 64;
        int grid = (int)((rows + kWaves - 1) / kWaves);
        if (grid < 1) grid = 1;
        if (grid > 256 * kWaveLinOcc) grid = 256 * kWaveLinOcc;
        const size_t lds = (size_t)kWaveLinLdsL * C * 3 * kBlock * sizeof(double2) + (size_t)kWaves * 352 * sizeof(double);   // 352 = Layout<4>'s row, the largest here
        const bool scalar_cams = MQS_WL_SCALAR_CAMS < 0 ? N >= kWlScalarMinLandmarks : MQS_WL_SCALAR_CAMS != 0;
        switch (C) {
#define MQS_CASE(c)                                                                                        \
    case c: {                                                                                              \
        static mqs_lds_opt_in opt, opt_s;               /* per device: dynamic LDS above 64 KiB needs the opt-in */ \
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(ba_linearize_wave_kernel<c, false>), lds));      \
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt_s, reinterpret_cast<const void *>(ba_linearize_wave_kernel<c, true>), lds));     \
        if ((parts & 1) && scalar_cams)                                                                    \
            hipLaunchKernelGGL((ba_linearize_wave_kernel<c, true>), dim3(grid), dim3(kBlock), lds, stream, poses, calib, sigma, \
                               points, obs, mask, prior_w, prior_xyz, N, lambda, partials);                \
        else if (parts & 1)                                                                                \
            hipLaunchKernelGGL((ba_linearize_wave_kernel<c, false>), dim3(grid), dim3(kBlock), lds, stream, poses, calib, sigma, \
                               points, obs, mask, prior_w, prior_xyz, N, lambda, partials);                \
        if (parts & 2)                                                                                     \
            hipLaunchKernelGGL((ba_finalize_kernel<c>), dim3((Layout<c>::kChunks * 32 + 63) / 64), dim3(kFinThreads), 0, stream, partials, grid, out, pp);   \
        break;                                                                                             \
    }
            MQS_CASE(2) MQS_CASE(3) MQS_CASE(4)
#undef MQS_CASE
        }
        MQS_HIP_CHECK(hipGetLastError());
        return MQS_OK;
    }
    const int grid = ba_grid(N);
    switch (C) {
#define MQS_CASE(c)                                                                                        \
    case c:                                                                                                \
        if (parts & 1)                                                                                     \
            hipLaunchKernelGGL((ba_linearize_kernel<c>), dim3(grid), dim3(kBlock), 0, stream, poses, calib, sigma, \
                               points, obs, mask, prior_w, prior_xyz, N, lambda, partials);                \
        if (parts & 2)                                                                                     \
            hipLaunchKernelGGL((ba_finalize_kernel<c>), dim3((Layout<c>::kChunks * 32 + 63) / 64), dim3(kFinThreads), 0, stream, partials, grid, out, pp);   \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // extern "C"

int mqs_ba_linearize_push(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                          const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                          double lambda, double *out, void *workspace, int64_t workspace_bytes, hipStream_t stream,
                          const mqs_peer_push *push)
{
    return ba_linearize_parts(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, out, workspace,
                              workspace_bytes, stream, 3, push);
}

// The wave lineariser alone (C in 2..4), and where the fused tail finds its partial rows, writes its quarters and raises its flags
// (all inside the caller's workspace; the epoch is unique per call, so nothing needs initialising).
std::atomic<unsigned long long> g_fin_epoch{1};
// test hook (tests/test_ba_gpu.py): the finalizer piece of the fused tail that does NOT raise its flag, so that the wait for it runs
// into its bound -- the only way to reach the time-out path of a protocol whose producers always arrive.  Per CALLING THREAD: only the
// launches the setting thread issues afterwards see it (a process-wide switch would make every other thread's problems time out too,
// and a problem's status word is sticky)
thread_local int g_withhold_flag = -1;
extern "C" int mqs_debug_ba_withhold_flag(int piece)
{
    g_withhold_flag = piece;
    return MQS_OK;
}

int mqs_ba_linearize_for_fused_tail(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                                    const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                                    double lambda, void *workspace, int64_t workspace_bytes, hipStream_t stream, mqs_ba_fin *fin)
{
    MQS_ARG_CHECK(mqs_ba_wave_path(C), "the fused finalize serves the wave lineariser (2..4 cameras)");
    MQS_ARG_CHECK(kWaveLinOcc == 1, "the fused finalize keeps its piece sums behind 256 partial rows: not with two workgroups per CU (MQS_WL_OCC)");
    int rc = ba_linearize_parts(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, nullptr, workspace,
                                workspace_bytes, stream, 1, nullptr);
    if (rc != MQS_OK) return rc;
    const int64_t kRow = ws_doubles_rt(C) / 512;
    const int64_t rows = (N + 63) / 64;
    int grid = (int)((rows + kWaves - 1) / kWaves);
    if (grid < 1) grid = 1;
    if (grid > 256 * kWaveLinOcc) grid = 256 * kWaveLinOcc;
    double *partials = static_cast<double *>(workspace);
    double *tail = partials + 512 * kRow - (kFinPieces * kQuarterStride + 64);      // the workspace's last 5 184 doubles: rows 256.. are unused here
    fin->partials = partials;
    fin->nrows = grid;
    fin->quarters = tail;
    fin->flags = reinterpret_cast<unsigned long long *>(tail + kFinPieces * kQuarterStride);
    fin->epoch = g_fin_epoch.fetch_add(1);
    fin->push = nullptr;
    fin->status = nullptr;
    fin->withhold = g_withhold_flag;
    return MQS_OK;
}

bool mqs_ba_wave_path(int C) { return C >= 2 && C <= 4 && wave_lineariser_enabled() && tail_fusion_enabled(); }
// (MQS_WL_OCC=2 builds: the fused finalize is off, WlWorkspaceCheck)

bool mqs_ba_fused_finalize_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MQS_BA_FINALIZE");
        v = ((e && strcmp(e, "kernel") == 0) || kWaveLinOcc != 1) ? 0 : 1;
    }
    return v == 1;
}

int mqs_ba_finalize_groups(int C)
{
    switch (C) {
#define MQS_CASE(c) case c: return (Layout<c>::kChunks * 32 + 63) / 64;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    return 0;
}

extern "C" {

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
        hipLaunchKernelGGL((ba_solve_small_kernel<c>), dim3(1), dim3(64), 0, stream, lin, poses, prior_poses, \
                           prior_sigmas, prior_mask, lambda, dpose, poses_out, info);                   \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4)
#undef MQS_CASE
#define MQS_CASE(c)                                                                                     \
    case c:                                                                                             \
        hipLaunchKernelGGL((ba_solve_kernel<c>), dim3(1), dim3(64), 0, stream, lin, poses, prior_poses, \
                           prior_sigmas, prior_mask, lambda, dpose, poses_out, info);                   \
        break;
        MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // extern "C"

// The fused tail; `peer` (comm.hip) non-null: the reduced system is the sum of the peers' rows in this rank's receive buffer.
int mqs_ba_tail_launch(const double *lin, const mqs_peer_recv *peer, const mqs_ba_fin *fin_, int C, const double *poses, const double *calib,
                       const double *sigma, const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                       const double *prior_xyz, int64_t N, double lambda, const double *prior_poses, const double *prior_sigmas,
                       const uint8_t *prior_mask, double *lin_out, double *dpose, double *poses_out, double *info,
                       double *points_out, hipStream_t stream)
{
    mqs_peer_recv pr = {};
    if (peer) pr = *peer;
    TailFin fin = {};
    fin.withhold = -1;
    if (fin_) {
        fin.partials = fin_->partials; fin.nrows = fin_->nrows; fin.quarters = fin_->quarters; fin.flags = fin_->flags;
        fin.epoch = fin_->epoch;
        if (fin_->push) fin.push = *fin_->push;
        fin.status = fin_->status;
        fin.withhold = fin_->withhold;
    }
    const int64_t rows64 = (N + 63) / 64;
    if (tail_form() == 2) {
        // second form: one workgroup of 12 waves per CU (the system is solved once per CU); below a chip's worth of landmarks one
        // workgroup per four rows; the finalizer pieces are the first ceil(kFinPieces * ceil(kRow / 64) / 3) <= 16 workgroups
        int64_t g = (rows64 + 3) / 4;
        if (g < 1) g = 1;
        const int64_t gfin = (kFinPieces * mqs_ba_finalize_groups(C) + kTail2Block / kBlock - 1) / (kTail2Block / kBlock);
        if (fin_ && g < gfin) g = gfin;
        if (g > 256) g = 256;
        switch (C) {
#define MQS_CASE(c)                                                                                                      \
    case c: {                                                                                                            \
        static mqs_lds_opt_in opt;                                                                                       \
        const size_t ring = (size_t)c * kTail2Block * sizeof(double2), park = (size_t)kTail2Block * (32 + 4 * c);       \
        const size_t lds = ring + (ring > park ? ring : park);      /* two ring halves; the parked batch lies over the second */ \
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(ba_tail2_kernel<c>), lds));                \
        hipLaunchKernelGGL((ba_tail2_kernel<c>), dim3((unsigned)g), dim3(kTail2Block), lds, stream, lin, pr, fin, poses, calib, sigma, points, obs, \
                           mask, prior_w, prior_xyz, N, lambda, prior_poses, prior_sigmas, prior_mask, lin_out, dpose, poses_out, \
                           info, points_out);                                                                            \
        break;                                                                                                           \
    }
            MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4)
#undef MQS_CASE
        default:
            mqs_set_error("the fused tail serves 1..4 cameras");
            return MQS_E_ARG;
        }
        MQS_HIP_CHECK(hipGetLastError());
        return MQS_OK;
    }
    // first form: persistent grid, at most 4 workgroups per CU (each solves the reduced system once), one batch of 256 landmarks per
    // workgroup below that; plus the publishing workgroup.  (Round 4 measured the alternative for large problems -- only eight
    // workgroups, one per XCD, solve and hand dpose to the others behind a flag: the redundant solves are 2.6 M of the launch's
    // 13.6 M vector instructions -- and it is SLOWER, 132.6 against 131.9 us per iteration at 1e6 and 75.5 against 74.2 us at
    // 400 k, profiles/r04/05: the solves of the four workgroups of a CU run side by side on its four SIMDs at the head of the
    // launch, when nothing else could use them, and the hand-over adds a hop to every workgroup's serial head.)
    int64_t g = (rows64 + 3) / 4;
    if (g < 1) g = 1;
    if (g > 1023) g = 1023;
    if (fin_ && g < 64) g = 64;                      // the finalizer pieces are the first kFinPieces * ceil(kRow / 64) <= 48 workgroups
    switch (C) {
#define MQS_CASE(c)                                                                                                      \
    case c:                                                                                                              \
        hipLaunchKernelGGL((ba_tail_kernel<c>), dim3((unsigned)g + 1), dim3(kBlock), 0, stream, lin, pr, fin, poses, calib, sigma, points, obs, \
                           mask, prior_w, prior_xyz, N, lambda, prior_poses, prior_sigmas, prior_mask, lin_out, dpose, poses_out, \
                           info, points_out);                                                                            \
        break;
        MQS_CASE(1) MQS_CASE(2) MQS_CASE(3) MQS_CASE(4)
#undef MQS_CASE
    default:
        mqs_set_error("the fused tail serves 1..4 cameras");
        return MQS_E_ARG;
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// A run of iterations, one launch each (ba_iterate_kernel): is this problem served?
bool mqs_ba_iterate_eligible(int C, int64_t N)
{
    static const bool off = [] { const char *e = getenv("MQS_BA_ITERATE"); return e && e[0] == '0'; }();      // A/B: the two-launch iteration
    if (off || !mqs_ba_wave_path(C) || !mqs_ba_fused_finalize_enabled() || kWaveLinOcc != 1 || tail_form() != 1) return false;
    if (N >= kWlScalarMinLandmarks) return false;
    const int64_t rows = (N + 63) / 64;
    const int64_t grid = (rows + kWaves - 1) / kWaves;
    return grid >= MQS_FIN_PIECES * mqs_ba_finalize_groups(C);           // the finalizer pieces are the first workgroups
}

// The tail of the iteration whose partial rows `fin` describes, and the linearisation of the next one at the new estimate
// (poses_out / points_out), into the same workspace: *fin_next describes the new rows (a new epoch; its push is the caller's).
int mqs_ba_iterate_launch(const mqs_peer_recv *peer, const mqs_ba_fin *fin_, int C, const double *poses, const double *calib, const double *sigma,
                          const double *points, const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz,
                          int64_t N, double lambda, const double *prior_poses, const double *prior_sigmas, const uint8_t *prior_mask,
                          double *lin_out, double *dpose, double *poses_out, double *info, double *points_out, void *workspace,
                          hipStream_t stream, mqs_ba_fin *fin_next)
{
    MQS_ARG_CHECK(fin_ != nullptr && fin_next != nullptr && mqs_ba_iterate_eligible(C, N), "not a problem the one-launch iteration serves");
    mqs_peer_recv pr = {};
    if (peer) pr = *peer;
    TailFin fin = {};
    fin.partials = fin_->partials; fin.nrows = fin_->nrows; fin.quarters = fin_->quarters; fin.flags = fin_->flags;
    fin.epoch = fin_->epoch;
    if (fin_->push) fin.push = *fin_->push;
    fin.status = fin_->status;
    fin.withhold = fin_->withhold;
    const int64_t rows = (N + 63) / 64;
    int grid = (int)((rows + kWaves - 1) / kWaves);
    if (grid > 256) grid = 256;
    MQS_ARG_CHECK(grid == fin_->nrows, "the rows of the previous launch come from the same grid");
    const size_t lds = (size_t)kWaveLinLdsL * C * 3 * kBlock * sizeof(double2) + (size_t)kWaves * 352 * sizeof(double);
    double *partials = static_cast<double *>(workspace);
    switch (C) {
#define MQS_CASE(c)                                                                                                          \
    case c: {                                                                                                                \
        static mqs_lds_opt_in opt;                                                                                           \
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(ba_iterate_kernel<c>), lds));                  \
        hipLaunchKernelGGL((ba_iterate_kernel<c>), dim3(grid), dim3(kBlock), lds, stream, pr, fin, poses, calib, sigma, points, obs, mask, \
                           prior_w, prior_xyz, N, lambda, prior_poses, prior_sigmas, prior_mask, lin_out, dpose, poses_out, info,   \
                           points_out, partials);                                                                            \
        break;                                                                                                               \
    }
        MQS_CASE(2) MQS_CASE(3) MQS_CASE(4)
#undef MQS_CASE
    default:
        mqs_set_error("the one-launch iteration serves 2..4 cameras");
        return MQS_E_ARG;
    }
    MQS_HIP_CHECK(hipGetLastError());
    *fin_next = *fin_;
    fin_next->epoch = g_fin_epoch.fetch_add(1);
    fin_next->push = nullptr;
    fin_next->withhold = g_withhold_flag;
    return MQS_OK;
}

extern "C" {

int mqs_ba_solve_backsub_dev(const double *lin, int C, const double *poses, const double *calib, const double *sigma,
                             const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                             const double *prior_xyz, int64_t N, double lambda, const double *prior_poses,
                             const double *prior_sigmas, const uint8_t *prior_mask, double *dpose, double *poses_out,
                             double *info, double *points_out, void *stream_)
{
    int rc = check_common(poses, calib, sigma, C, points, obs, N);
    if (rc != MQS_OK) return rc;
    MQS_ARG_CHECK(lin && dpose && (N == 0 || points_out), "lin, dpose, points_out must not be null");
    MQS_ARG_CHECK(mqs_aligned16(points_out), "points_out must be 16-byte aligned");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    MQS_ARG_CHECK(!prior_mask || (prior_poses && prior_sigmas), "prior_poses/prior_sigmas required with prior_mask");
    if (C <= 4 && tail_fusion_enabled())
        return mqs_ba_tail_launch(lin, nullptr, nullptr, C, poses, calib, sigma, points, obs, mask, prior_w, prior_xyz, N, lambda, prior_poses,
                                  prior_sigmas, prior_mask, nullptr, dpose, poses_out, info, points_out,
                                  static_cast<hipStream_t>(stream_));
    rc = mqs_ba_solve_dev(lin, C, poses, prior_poses, prior_sigmas, prior_mask, lambda, dpose, poses_out, info, stream_);
    if (rc != MQS_OK) return rc;
    return mqs_ba_backsub_dev(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, dpose, points_out, stream_);
}

// Average duration (ms) of `reps` back-to-back launches of ONE kernel of the iteration, hipEvents on `stream`
// (what: 0 lineariser kernel alone, 1 finalize alone, 2 solve + retract, 3 back-substitution, 4 solve + retract +
// back-substitution as the iteration issues them: one launch for C <= 4).
int mqs_ba_time_dev(int what, const double *poses, const double *calib, const double *sigma, int C, const double *points,
                    const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                    double lambda, double *lin, double *dpose, double *poses_out, double *points_out, void *workspace,
                    int64_t workspace_bytes, int reps, void *stream_, float *avg_ms)
{
    MQS_ARG_CHECK(what >= 0 && what <= 4 && reps >= 1 && avg_ms != nullptr, "what in 0..4, reps >= 1, avg_ms must not be null");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto once = [&]() -> int {
        switch (what) {
        case 0: case 1:
            return ba_linearize_parts(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, lin, workspace,
                                      workspace_bytes, stream_, what == 0 ? 1 : 2);
        case 2: return mqs_ba_solve_dev(lin, C, poses, nullptr, nullptr, nullptr, lambda, dpose, poses_out, nullptr, stream_);
        case 3: return mqs_ba_backsub_dev(poses, calib, sigma, C, points, obs, mask, prior_w, prior_xyz, N, lambda, dpose,
                                          points_out, stream_);
        default: return mqs_ba_solve_backsub_dev(lin, C, poses, calib, sigma, points, obs, mask, prior_w, prior_xyz, N, lambda, nullptr,
                                                 nullptr, nullptr, dpose, poses_out, nullptr, points_out, stream_);
        }
    };
    int rc = once();                                  // warm-up (and the argument checks)
    if (rc != MQS_OK) return rc;
    hipEvent_t e0, e1;
    MQS_HIP_CHECK(hipEventCreate(&e0));
    MQS_HIP_CHECK(hipEventCreate(&e1));
    MQS_HIP_CHECK(hipEventRecord(e0, stream));
    for (int k = 0; k < reps && rc == MQS_OK; ++k) rc = once();
    MQS_HIP_CHECK(hipEventRecord(e1, stream));
    MQS_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MQS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / reps;
    return rc;
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
