// N-view landmark triangulation kernels for gfx950 (MI355X): linear-LS (T1), Hartley-Sturm
// iterative-LS (T2) and the homogeneous "linear-eigen" DLT (T3).
//
// Replaces the per-point cvSolve/SVD loops of
//   Work/python_libs/triangulation_c/triangulation.c:65-83,104-161 and the
//   cv2.triangulatePoints call of Work/python_libs/triangulation.py:20
// (reference paths; the arithmetic each line mirrors is cited in tri_math.h).
//
// Mapping to the machine
//   * one thread per landmark, 256-thread workgroups, grid capped at 2048 groups and
//     grid-strided beyond that;
//   * camera matrices ([C][3][4], <= 768 B) are staged once per workgroup in LDS and read
//     from there as wave-uniform broadcasts;
//   * observations are camera-major ([C][N][2] f64): lane i reads 16 contiguous bytes per
//     camera, a wave reads 1 KiB per load instruction;
//   * results (24 B per landmark) are transposed through LDS so that the workgroup writes its
//     6 KiB of output as 16-byte-per-lane fully coalesced stores; status codes are 4 B per
//     lane contiguous;
//   * all arithmetic is fp64 in registers (the reference ABI is float64 end to end);
//     HBM traffic per landmark is the algorithmic minimum 16*C + 24 (+4 | +1) bytes.
#include "mqs_common.h"
#include "tri_math.h"
#include "cam_math.h"

namespace {

constexpr int kBlock = 256;

enum TriKind { kLinearLS = 0, kIterativeLS = 1, kLinearEigen = 2, kLsAndIterative = 3 };

// Minimum waves per SIMD requested from the register allocator per kernel kind (tuned on
// MI355X, see DESIGN.md "Triangulation kernels"): the iterative kernel is fp64-VALU bound and
// needs several resident waves per SIMD to cover dependent-FMA and v_rcp_f64 latency.
#ifndef MQS_ITER_WAVES
#define MQS_ITER_WAVES 3
#endif
// The iteration keeps 12 doubles per camera live (Gram piece 6, right-hand side 3, weight and two depths): beyond four
// cameras that no longer fits the 168 registers of three waves per SIMD (it spilled inside the loop: 5 / 6 / 8 cameras
// ran 1.8x / 3.6x / 11x the 4-camera time instead of ~1.2x / 1.35x / 1.7x) -- fewer resident waves, no spills.
constexpr int waves_for(int kind, int cams) { return (kind != 1 && kind != 3) ? 2 : (cams <= 4 ? MQS_ITER_WAVES : (cams <= 6 ? 2 : 1)); }

// PIX: the observations are PIXELS and `intr` holds [C][9] intrinsics (fx fy cx cy k1 k2 p1 p2 k3): the
// undistort + normalise step the reference runs right before triangulating (cv2.undistortPoints,
// slam2.py:551-552) is applied on load, saving its 2 x 16*C bytes per landmark of HBM round trip.
// F32: the observations are float32 [C][N][2] (what slam2.py hands over: it works in float32, slam2.py:19, and the reference
// wrapper widens them on the host, triangulation_c/__init__.py:32-33): widened on load instead -- the same doubles, half the
// input bytes (linear-LS, the HBM-bound kernel: 88 -> 56 B per landmark at four cameras).
template <int C, int KIND, bool PIX, bool F32 = false>
__global__ __launch_bounds__(kBlock, waves_for(KIND, C)) void tri_kernel(const double *__restrict__ u, const double *__restrict__ P,
                                                                      const double *__restrict__ intr,
                                                     int64_t N, double tol, int max_iter, double max_coord,
                                                     double *__restrict__ x, int32_t *__restrict__ status,
                                                     uint8_t *__restrict__ ok, double *__restrict__ x_ls)
{
    __shared__ double sP[C * 12];
    __shared__ double sX[kBlock * 3];
    __shared__ double sI[PIX ? C * 9 : 1];
    // fused linear-LS + iterative-LS: the first solve (3) and its factor (6 + ok) wait here for the refinement step
    __shared__ double sFirst[KIND == kLsAndIterative ? 10 * kBlock : 1];
    // pixel input + an iterative kind: the undistorted observations wait here for the refinement step (undistorting is 5
    // fixed-point iterations with a division each, ~200 instructions per camera: re-reading the pixels and undistorting
    // them again cost the fused iterative kernel a quarter of its time)
    constexpr bool kParkUV = PIX && (KIND == kIterativeLS || KIND == kLsAndIterative);
    __shared__ double sUV[kParkUV ? 2 * C * kBlock : 1];

    const int tid = threadIdx.x;
    if (tid < C * 12) sP[tid] = P[tid];
    if (PIX && tid < C * 9) sI[tid] = intr[tid];
    __syncthreads();

    const double2 *__restrict__ u2 = reinterpret_cast<const double2 *>(u);
    const int64_t stride = (int64_t)gridDim.x * kBlock;

    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < N; base += stride) {
        const int64_t i = base + tid;
        const bool live = i < N;

        double uv[C][2];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double2 v = make_double2(0.0, 0.0);
            if (live) {
                if (F32) {
                    const float2 t = reinterpret_cast<const float2 *>(u)[(int64_t)c * N + i];
                    v = make_double2((double)t.x, (double)t.y);
                } else {
                    v = u2[(int64_t)c * N + i];
                }
            }
            if (PIX) mqs::cam::undistort_pixel(sI + 9 * c, v.x, v.y, v.x, v.y);
            if (kParkUV) {
                sUV[(2 * c) * kBlock + tid] = v.x;
                sUV[(2 * c + 1) * kBlock + tid] = v.y;
            }
            uv[c][0] = v.x;
            uv[c][1] = v.y;
        }

        mqs::Vec3 r;
        if (KIND == kLinearLS) {
            r = mqs::linear_ls_point<C>(uv, sP);
        } else if (KIND == kIterativeLS || KIND == kLsAndIterative) {
            mqs::IterResult<C> it;
            if (KIND == kLsAndIterative) {
                struct Park {
                    double *s;
                    __device__ __forceinline__ void operator()(const mqs::Vec3 &x0, const mqs::Ldlt3 &f0) const
                    {
                        s[0 * kBlock] = x0.x; s[1 * kBlock] = x0.y; s[2 * kBlock] = x0.z;
                        s[3 * kBlock] = f0.i0; s[4 * kBlock] = f0.i1; s[5 * kBlock] = f0.i2;
                        s[6 * kBlock] = f0.l10; s[7 * kBlock] = f0.l20; s[8 * kBlock] = f0.l21;
                        s[9 * kBlock] = f0.ok ? 1.0 : 0.0;
                    }
                };
                const Park park = {sFirst + tid};
                mqs::iterative_ls_core<C, Park>(uv, sP, P, tol, max_iter, it, park);
            } else {
                mqs::iterative_ls_core<C>(uv, sP, P, tol, max_iter, it);
            }
            if (live) status[i] = it.status;
            r = it.x;
            if (it.solved) {
                // re-read the observations (L2 / Infinity-Cache hits) for the refinement step
                // rather than keeping 2C doubles live through the iteration
                const int64_t j = i + mqs::opaque_zero();
                double uv2[C][2];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    double2 v = make_double2(0.0, 0.0);
                    if (kParkUV) {
                        v = make_double2(sUV[(2 * c) * kBlock + tid + (int)(j - i)], sUV[(2 * c + 1) * kBlock + tid + (int)(j - i)]);
                    } else if (live) {
                        if (F32) {
                            const float2 t = reinterpret_cast<const float2 *>(u)[(int64_t)c * N + j];
                            v = make_double2((double)t.x, (double)t.y);
                        } else {
                            v = u2[(int64_t)c * N + j];
                        }
                    }
                    uv2[c][0] = v.x;
                    uv2[c][1] = v.y;
                }
                const mqs::Sym3 unused = {0, 0, 0, 0, 0, 0};
                r = mqs::refine<C>(it.x, uv2, sP, it.w2, unused, it.f);
                if (KIND == kLsAndIterative) {
                    // linear-LS = the first solve, refined with unit weights (linear_ls_point), written through the same
                    // LDS transpose as the iterative result below
                    const double *s = sFirst + tid;
                    const mqs::Vec3 x0 = {s[0 * kBlock], s[1 * kBlock], s[2 * kBlock]};
                    const mqs::Ldlt3 f0 = {s[3 * kBlock], s[4 * kBlock], s[5 * kBlock], s[6 * kBlock], s[7 * kBlock], s[8 * kBlock],
                                           s[9 * kBlock] != 0.0};
                    double ones[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) ones[c] = 1.0;
                    const mqs::Vec3 rl = mqs::refine<C>(x0, uv2, sP, ones, unused, f0);
                    sX[tid * 3 + 0] = rl.x;
                    sX[tid * 3 + 1] = rl.y;
                    sX[tid * 3 + 2] = rl.z;
                    __syncthreads();
                    const int64_t rem = N - base;
                    const int npts = rem < kBlock ? (int)rem : kBlock;
                    const int ndbl = npts * 3;
                    double *__restrict__ xo = x_ls + base * 3;
                    const double2 *sX2 = reinterpret_cast<const double2 *>(sX);
                    double2 *xo2 = reinterpret_cast<double2 *>(xo);
                    for (int p = tid; p < (ndbl >> 1); p += kBlock) xo2[p] = sX2[p];
                    if ((ndbl & 1) && tid == 0) xo[ndbl - 1] = sX[ndbl - 1];
                    __syncthreads();
                }
            }
        } else {
            bool o;
            r = mqs::linear_eigen_point<C>(uv, sP, max_coord, o);
            if (live) ok[i] = o ? 1 : 0;
        }

        // LDS transpose: [thread][3] doubles -> workgroup-contiguous 16-byte pieces.
        sX[tid * 3 + 0] = r.x;
        sX[tid * 3 + 1] = r.y;
        sX[tid * 3 + 2] = r.z;
        __syncthreads();
        const int64_t rem = N - base;
        const int npts = rem < kBlock ? (int)rem : kBlock;
        const int ndbl = npts * 3;                         // doubles this workgroup owns
        double *__restrict__ xo = x + base * 3;            // 16-byte aligned: base % 256 == 0
        const double2 *sX2 = reinterpret_cast<const double2 *>(sX);
        double2 *xo2 = reinterpret_cast<double2 *>(xo);
        const int npair = ndbl >> 1;
        for (int p = tid; p < npair; p += kBlock) xo2[p] = sX2[p];
        if ((ndbl & 1) && tid == 0) xo[ndbl - 1] = sX[ndbl - 1];
        __syncthreads();
    }
}

template <int KIND>
int launch_tri(const double *u, const double *P, int C, int64_t N, double tol, int max_iter, double max_coord,
               double *x, int32_t *status, uint8_t *ok, hipStream_t stream, const double *intr = nullptr,
               double *x_ls = nullptr, bool u_is_f32 = false)
{
    MQS_ARG_CHECK(!u_is_f32 || (!intr && KIND != kLsAndIterative), "float32 observations: un-fused kinds only");
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    if (N == 0) return MQS_OK;
    MQS_ARG_CHECK(u && P && x, "u, P, x must not be null");
    MQS_ARG_CHECK(mqs_aligned16(u) && mqs_aligned16(x), "device pointers must be 16-byte aligned");
    if (KIND == kIterativeLS || KIND == kLsAndIterative) MQS_ARG_CHECK(status != nullptr, "status must not be null");
    if (KIND == kLsAndIterative) {
        MQS_ARG_CHECK(x_ls != nullptr && mqs_aligned16(x_ls), "x_ls must not be null and 16-byte aligned");
        MQS_ARG_CHECK(max_iter >= 1, "max_iter >= 1 (linear-LS is the first solve of the iteration)");
    }
    if (KIND == kLinearEigen) MQS_ARG_CHECK(ok != nullptr, "ok must not be null");
    const dim3 grid(mqs_stream_grid(N, kBlock)), block(kBlock);
    switch (C) {
#define MQS_CASE(c)                                                                             \
    case c:                                                                                     \
        if (intr)                                                                               \
            hipLaunchKernelGGL((tri_kernel<c, KIND, true>), grid, block, 0, stream, u, P, intr, N, tol, \
                               max_iter, max_coord, x, status, ok, x_ls);                       \
        else if (u_is_f32)                                                                      \
            hipLaunchKernelGGL((tri_kernel<c, (KIND == kLsAndIterative ? kLinearLS : KIND), false, true>), grid, block, 0, stream, u, P, \
                               intr, N, tol, max_iter, max_coord, x, status, ok, x_ls);         \
        else                                                                                    \
            hipLaunchKernelGGL((tri_kernel<c, KIND, false>), grid, block, 0, stream, u, P, intr, N, tol, \
                               max_iter, max_coord, x, status, ok, x_ls);                       \
        break;
        MQS_CASE(2) MQS_CASE(3) MQS_CASE(4) MQS_CASE(5) MQS_CASE(6) MQS_CASE(7) MQS_CASE(8)
#undef MQS_CASE
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// Host-pointer driver: stage inputs into the ctx scratch, run, copy back.
template <int KIND, typename TU = double>
int run_host(mqs_ctx *ctx, const TU *const *u_cams, const double *const *P_cams, int C, int64_t N, double tol,
             int max_iter, double max_coord, double *x, int32_t *status, uint8_t *ok)
{
    constexpr bool kF32 = sizeof(TU) == 4;                 // float32 observations: half the bytes over the link, widened on load
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N >= 0, "N >= 0");
    if (N == 0) return MQS_OK;
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    // layout of the scratch: [u: C*N*2 f64][x: N*3 f64][P: C*12 f64][status: N i32][ok: N u8], 256-B aligned pieces
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_u = 0;
    const size_t o_x = up(o_u + (size_t)C * N * 2 * sizeof(TU));
    const size_t o_P = up(o_x + (size_t)N * 24);
    const size_t o_s = up(o_P + (size_t)C * 96);
    const size_t o_k = up(o_s + (size_t)N * 4);
    const size_t total = up(o_k + (size_t)N);
    mqs_stage st;
    int rc = mqs_stage_begin(ctx, total, &st);
    if (rc != MQS_OK) return rc;
    char *d = st.base;
    TU *d_u = reinterpret_cast<TU *>(d + o_u);
    double *d_x = reinterpret_cast<double *>(d + o_x);
    double *d_P = reinterpret_cast<double *>(d + o_P);
    int32_t *d_s = reinterpret_cast<int32_t *>(d + o_s);
    uint8_t *d_k = reinterpret_cast<uint8_t *>(d + o_k);
    for (int c = 0; c < C; ++c) {
        MQS_ARG_CHECK(u_cams[c] && P_cams[c], "per-camera pointers must not be null");
        MQS_HIP_CHECK(mqs_stage_in(&st, d_u + (size_t)c * N * 2, u_cams[c], (size_t)N * 2 * sizeof(TU)));
        MQS_HIP_CHECK(mqs_stage_in(&st, d_P + c * 12, P_cams[c], 96));
    }
    rc = launch_tri<KIND>(reinterpret_cast<const double *>(d_u), d_P, C, N, tol, max_iter, max_coord, d_x, d_s, d_k, ctx->stream,
                          nullptr, nullptr, kF32);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(mqs_stage_out(&st, x, d_x, (size_t)N * 24));
    if (KIND == kIterativeLS) MQS_HIP_CHECK(mqs_stage_out(&st, status, d_s, (size_t)N * 4));
    if (KIND == kLinearEigen) MQS_HIP_CHECK(mqs_stage_out(&st, ok, d_k, (size_t)N));
    MQS_HIP_CHECK(mqs_stage_end(&st));
    return MQS_OK;
}

void split_packed(const double *u, const double *P, int C, int64_t N, const double **uc, const double **Pc)
{
    for (int c = 0; c < C && c < MQS_MAX_CAMS; ++c) {
        uc[c] = u ? u + (size_t)c * N * 2 : nullptr;
        Pc[c] = P ? P + c * 12 : nullptr;
    }
}

}  // namespace

extern "C" {

int mqs_triangulate_linear_ls_dev(const double *u, const double *P, int C, int64_t N, double *x, void *stream)
{
    return launch_tri<kLinearLS>(u, P, C, N, 0.0, 0, 0.0, x, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int mqs_triangulate_iterative_ls_dev(const double *u, const double *P, int C, int64_t N, double tolerance, int max_iter,
                                     double *x, int32_t *status, void *stream)
{
    return launch_tri<kIterativeLS>(u, P, C, N, tolerance, max_iter, 0.0, x, status, nullptr,
                                    static_cast<hipStream_t>(stream));
}

int mqs_triangulate_ls_and_iterative_dev(const double *u, const double *P, int C, int64_t N, double tolerance, int max_iter,
                                         double *x_ls, double *x_it, int32_t *status, void *stream)
{
    return launch_tri<kLsAndIterative>(u, P, C, N, tolerance, max_iter, 0.0, x_it, status, nullptr,
                                       static_cast<hipStream_t>(stream), nullptr, x_ls);
}

int mqs_triangulate_linear_eigen_dev(const double *u, const double *P, int C, int64_t N, double max_coord, double *x,
                                     uint8_t *ok, void *stream)
{
    return launch_tri<kLinearEigen>(u, P, C, N, 0.0, 0, max_coord, x, nullptr, ok, static_cast<hipStream_t>(stream));
}

int mqs_triangulate_pixels_dev(int kind, const double *pixels, const double *intr, const double *P, int C, int64_t N,
                               double tolerance, int max_iter, double max_coord, double *x, int32_t *status, uint8_t *ok,
                               void *stream)
{
    MQS_ARG_CHECK(kind >= 0 && kind <= 2, "kind in {0 linear_ls, 1 iterative_ls, 2 linear_eigen}");
    MQS_ARG_CHECK(intr != nullptr, "intr must not be null");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (kind == 0) return launch_tri<kLinearLS>(pixels, P, C, N, 0.0, 0, 0.0, x, nullptr, nullptr, s, intr);
    if (kind == 1) return launch_tri<kIterativeLS>(pixels, P, C, N, tolerance, max_iter, 0.0, x, status, nullptr, s, intr);
    return launch_tri<kLinearEigen>(pixels, P, C, N, 0.0, 0, max_coord, x, nullptr, ok, s, intr);
}

int mqs_triangulate_linear_ls(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N, double *x)
{
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N == 0 || (u && P && x), "u, P, x must not be null");
    const double *uc[MQS_MAX_CAMS], *Pc[MQS_MAX_CAMS];
    split_packed(u, P, C, N, uc, Pc);
    return run_host<kLinearLS>(ctx, uc, Pc, C, N, 0.0, 0, 0.0, x, nullptr, nullptr);
}

int mqs_triangulate_iterative_ls(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N, double tolerance,
                                 int max_iter, double *x, int32_t *status)
{
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N == 0 || (u && P && x && status), "u, P, x, status must not be null");
    const double *uc[MQS_MAX_CAMS], *Pc[MQS_MAX_CAMS];
    split_packed(u, P, C, N, uc, Pc);
    return run_host<kIterativeLS>(ctx, uc, Pc, C, N, tolerance, max_iter, 0.0, x, status, nullptr);
}

int mqs_triangulate_linear_eigen(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N, double max_coord,
                                 double *x, uint8_t *ok)
{
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N == 0 || (u && P && x && ok), "u, P, x, ok must not be null");
    const double *uc[MQS_MAX_CAMS], *Pc[MQS_MAX_CAMS];
    split_packed(u, P, C, N, uc, Pc);
    return run_host<kLinearEigen>(ctx, uc, Pc, C, N, 0.0, 0, max_coord, x, nullptr, ok);
}

int mqs_linear_LS_triangulation(mqs_ctx *ctx, const double *u1, const double *P1, const double *u2, const double *P2,
                                int64_t N, double *x)
{
    MQS_ARG_CHECK(N == 0 || (u1 && P1 && u2 && P2 && x), "arguments must not be null");
    const double *uc[2] = {u1, u2}, *Pc[2] = {P1, P2};
    return run_host<kLinearLS>(ctx, uc, Pc, 2, N, 0.0, 0, 0.0, x, nullptr, nullptr);
}

int mqs_iterative_LS_triangulation(mqs_ctx *ctx, const double *u1, const double *P1, const double *u2, const double *P2,
                                   int64_t N, double tolerance, double *x, int32_t *x_status)
{
    MQS_ARG_CHECK(N == 0 || (u1 && P1 && u2 && P2 && x && x_status), "arguments must not be null");
    const double *uc[2] = {u1, u2}, *Pc[2] = {P1, P2};
    return run_host<kIterativeLS>(ctx, uc, Pc, 2, N, tolerance, MQS_TRI_MAX_ITER_DEFAULT, 0.0, x, x_status, nullptr);
}

// float32 observations through the host-pointer boundary (slam2.py hands float32 points over; the reference's wrapper widens
// them on the host, __init__.py:32-33 -- here they cross the link as float32 and are widened on load)
int mqs_triangulate_f32(mqs_ctx *ctx, int kind, const float *u, const double *P, int C, int64_t N, double tolerance,
                        int max_iter, double max_coord, double *x, int32_t *status, uint8_t *ok)
{
    MQS_ARG_CHECK(kind >= 0 && kind <= 2, "kind in {0,1,2}");
    MQS_ARG_CHECK(C >= 2 && C <= MQS_MAX_CAMS, "2 <= C <= MQS_MAX_CAMS");
    MQS_ARG_CHECK(N == 0 || (u && P && x), "u, P, x must not be null");
    const float *uc[MQS_MAX_CAMS];
    const double *Pc[MQS_MAX_CAMS];
    for (int c = 0; c < C; ++c) { uc[c] = u + (size_t)c * N * 2; Pc[c] = P + c * 12; }
    if (kind == 0) return run_host<kLinearLS, float>(ctx, uc, Pc, C, N, 0.0, 0, 0.0, x, nullptr, nullptr);
    if (kind == 1) return run_host<kIterativeLS, float>(ctx, uc, Pc, C, N, tolerance, max_iter, 0.0, x, status, nullptr);
    return run_host<kLinearEigen, float>(ctx, uc, Pc, C, N, 0.0, 0, max_coord, x, nullptr, ok);
}

int mqs_triangulation_2view_f32(mqs_ctx *ctx, int kind, const float *u1, const double *P1, const float *u2, const double *P2,
                                int64_t N, double tolerance, double max_coord, double *x, int32_t *status, uint8_t *ok)
{
    MQS_ARG_CHECK(kind >= 0 && kind <= 2, "kind in {0,1,2}");
    MQS_ARG_CHECK(N == 0 || (u1 && P1 && u2 && P2 && x), "arguments must not be null");
    const float *uc[2] = {u1, u2};
    const double *Pc[2] = {P1, P2};
    if (kind == 0) return run_host<kLinearLS, float>(ctx, uc, Pc, 2, N, 0.0, 0, 0.0, x, nullptr, nullptr);
    if (kind == 1) return run_host<kIterativeLS, float>(ctx, uc, Pc, 2, N, tolerance, MQS_TRI_MAX_ITER_DEFAULT, 0.0, x, status, nullptr);
    return run_host<kLinearEigen, float>(ctx, uc, Pc, 2, N, 0.0, 0, max_coord, x, nullptr, ok);
}

int mqs_triangulate_f32_dev(int kind, const float *u, const double *P, int C, int64_t N, double tolerance, int max_iter,
                            double max_coord, double *x, int32_t *status, uint8_t *ok, void *stream_)
{
    MQS_ARG_CHECK(kind >= 0 && kind <= 2, "kind in {0,1,2}");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double *ud = reinterpret_cast<const double *>(u);        // typed again inside the F32 kernels
    if (kind == 0) return launch_tri<kLinearLS>(ud, P, C, N, 0.0, 0, 0.0, x, nullptr, nullptr, stream, nullptr, nullptr, true);
    if (kind == 1) return launch_tri<kIterativeLS>(ud, P, C, N, tolerance, max_iter, 0.0, x, status, nullptr, stream, nullptr, nullptr, true);
    return launch_tri<kLinearEigen>(ud, P, C, N, 0.0, 0, max_coord, x, nullptr, ok, stream, nullptr, nullptr, true);
}

int mqs_time_triangulate_dev(int kernel, const double *u, const double *P, int C, int64_t N, double tolerance,
                             int max_iter, double *x, int32_t *status, uint8_t *ok, int reps, void *stream_,
                             float *avg_ms)
{
    MQS_ARG_CHECK((kernel >= 0 && kernel <= 2) || (kernel >= 10 && kernel <= 12), "kernel in {0,1,2} (float64 u) or {10,11,12} (float32 u)");
    MQS_ARG_CHECK(reps >= 1 && avg_ms, "reps >= 1, avg_ms not null");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipEvent_t e0, e1;
    MQS_HIP_CHECK(hipEventCreate(&e0));
    MQS_HIP_CHECK(hipEventCreate(&e1));
    int rc = MQS_OK;
    MQS_HIP_CHECK(hipEventRecord(e0, stream));
    for (int r = 0; r < reps && rc == MQS_OK; ++r) {
        if (kernel >= 10)
            rc = mqs_triangulate_f32_dev(kernel - 10, reinterpret_cast<const float *>(u), P, C, N, tolerance, max_iter,
                                         MQS_TRI_MAX_COORD_DEFAULT, x, status, ok, stream);
        else if (kernel == 0) rc = mqs_triangulate_linear_ls_dev(u, P, C, N, x, stream);
        else if (kernel == 1) rc = mqs_triangulate_iterative_ls_dev(u, P, C, N, tolerance, max_iter, x, status, stream);
        else rc = mqs_triangulate_linear_eigen_dev(u, P, C, N, MQS_TRI_MAX_COORD_DEFAULT, x, ok, stream);
    }
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
