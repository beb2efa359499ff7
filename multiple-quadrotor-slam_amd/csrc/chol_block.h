// Pieces the blocked Cholesky kernels share (ba_sparse.hip: the banded factorisation in natural order; chol_nd.hip: the
// same band cut into independent chunks).  Block size 32, fp64, lower triangle, row-major, in place.
#pragma once
#include "mqs_common.h"
#include "tri_math.h"

namespace mqs {
namespace chol {

constexpr int NB = 32;
constexpr int kLd = NB + 1;          // LDS row stride of a 32 x 32 tile

__device__ __forceinline__ double read_lane_d(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// 32 x 32 tile products on the fp64 matrix pipe.  v_mfma_f64_16x16x4_f64 (layout probed on gfx950, tools/probes/
// mfma_f64_layout.hip): a = A[i = lane % 16][k = lane / 16], b = B[k = lane / 16][j = lane % 16], and the four result registers
// hold D[i = lane / 16 + 4 v][j = lane % 16].  A workgroup's four wavefronts take one 16 x 16 quadrant (wr, wc) each of
//     C[r][c] = sum_k P[r][k] Q[c][k]          (P, Q: 32 x 32 in LDS, row stride kLd)
// -- 8 MFMAs and 16 ds_read_b64 per wavefront where the 2 x 2 register tile of the vector version issues 128 LDS reads for 128
// FMAs per thread and is bound by the LDS (0.85 us per tile product per compute unit).
using double4v = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double4v tile_quadrant_mfma(const double *P, const double *Q, int wr, int wc, int lane,
                                                       double4v acc = double4v{0.0, 0.0, 0.0, 0.0})
{
    const double *p = P + (16 * wr + (lane & 15)) * kLd + (lane >> 4);
    const double *q = Q + (16 * wc + (lane & 15)) * kLd + (lane >> 4);
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p[4 * s], q[4 * s], acc, 0, 0, 0);
    return acc;
}
// entry v of the quadrant result: tile row / column
__device__ __forceinline__ int quadrant_row(int wr, int lane, int v) { return 16 * wr + (lane >> 4) + 4 * v; }
__device__ __forceinline__ int quadrant_col(int wc, int lane) { return 16 * wc + (lane & 15); }

// The 32 x 32 diagonal block at origin t0, taken from LDS (sT, row stride kLd, lower triangle valid) and written to global
// memory factored: one wavefront (`lane` 0..63).  Lanes 0..31: lane r keeps row r in registers, right-looking Cholesky with
// v_readlane broadcasts (all register indices static).  Lanes 32..63 compute the INVERSE of the factor at the same time, in
// the same instructions: lane 32 + c solves L y = e_c column-oriented, and step k of that substitution,
// y[j] -= L[j][k] y[k] (j > k), is the trailing update's row[j] -= L[lane][k] L[j][k] with y[k] in the place of L[lane][k].
// Output layout: L in the lower triangle (diagonal included), inv(L)'s strictly lower part TRANSPOSED in the block's
// strictly upper triangle (row c, columns c+1.. = column c of inv(L)).  Rows beyond the matrix act as identity.
__device__ __forceinline__ void factor_diag_block_from_lds(const double *sT, double *__restrict__ A, int n, int t0,
                                                           int *__restrict__ bad, int lane)
{
    const int nb2 = (n - t0) < NB ? (n - t0) : NB;
    const int r = lane & 31;
    const bool upper = lane >= 32, live = r < nb2;
    double v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double a = sT[(live ? r : nb2 - 1) * kLd + (j < nb2 ? j : nb2 - 1)];
        v[j] = (!upper && live && j < nb2 && j <= r) ? a : ((j == r) ? 1.0 : 0.0);
    }
    bool notpd = false;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const double akk = read_lane_d(v[k], k);
        notpd = notpd || !(akk > 0.0);
        const double inv = mqs::rsqrt_d(akk > 0.0 ? akk : 1.0);
        const double m = v[k] * inv;                      // L[lane][k] (lane >= k) | inv(L)[k][c]
        v[k] = m;
#pragma unroll
        for (int j = k + 1; j < NB; ++j) {
            v[j] = fma(-m, read_lane_d(m, j), v[j]);      // the broadcast is L[j][k]: lane j < 32
            asm volatile("" : "+v"(v[j]));                // keeps the right-looking order (independent FMAs); the compiler
                                                          // otherwise turns the unrolled nest left-looking: one dependent chain per column
        }
    }
    if (lane == 0 && notpd) *bad = 1;
    if (live) {
        double *Arow = A + (int64_t)(t0 + r) * n + t0;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb2 && (upper ? j > r : j <= r)) Arow[j] = v[j];        // upper half: v[j] = inv(L)[j][r], j > r
    }
}

// The same factorisation by FOUR wavefronts (a 256-thread workgroup; `tid` 0..255, every thread must call): wavefront w keeps
// the columns j = w, w + 4, ... (8 registers) of the same lane layout -- lanes 0..31 rows of L, lanes 32..63 columns of the
// inverse.  The owner of pivot k scales its column and hands it to the others through LDS (sM: 2 x 64 doubles, alternating so
// that one workgroup barrier per pivot suffices); every wavefront then updates its own columns with broadcast reads of
// L[j][k] -- no v_readlane pair per (k, j), and a quarter of the FMAs per wavefront.  One wavefront issues an instruction every
// 4-5 cycles at best, and the single-wave version is ~2 100 of them (7.5 us of a 16 us factor step).  Measured: about 1 us less
// per factor step (linearise + solve 0.73 -> 0.707 ms chunked, 3.33 -> 3.21 ms in natural order); the barrier per pivot is what
// is left, and scaling the next pivot's column ahead of the other updates did not shorten it (0.713-0.724 ms).  Same operations
// on every element in the same order: the result is bit-identical to the single-wave version.
// kSc1: the factor is stored write-through (relaxed agent-scope stores: global_store ... sc1), for kernels whose other workgroups
// read it later in the SAME launch (slam_ba.hip).
template <int w, bool kSc1 = false>                          // the wavefront's index in the workgroup: compile time, so that
__device__ __forceinline__ void factor_diag_block_4w_wave(const double *sT, double *__restrict__ A, int n, int t0,    // j > k folds
                                                          int *__restrict__ bad, int lane, double *sM)
{
    const int nb2 = (n - t0) < NB ? (n - t0) : NB;
    const int r = lane & 31;
    const bool upper = lane >= 32, live = r < nb2;
    double v[NB / 4];
#pragma unroll
    for (int jj = 0; jj < NB / 4; ++jj) {
        const int j = 4 * jj + w;
        const double a = sT[(live ? r : nb2 - 1) * kLd + (j < nb2 ? j : nb2 - 1)];
        v[jj] = (!upper && live && j < nb2 && j <= r) ? a : ((j == r) ? 1.0 : 0.0);
    }
    bool notpd = false;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        double *buf = sM + (k & 1) * 64;
        if (w == (k & 3)) {                                  // the pivot column's owner
            const double akk = read_lane_d(v[k >> 2], k);
            notpd = notpd || !(akk > 0.0);
            const double inv = mqs::rsqrt_d(akk > 0.0 ? akk : 1.0);
            const double m = v[k >> 2] * inv;                // L[lane][k] (lane >= k) | inv(L)[k][c]
            v[k >> 2] = m;
            buf[lane] = m;
        }
        __syncthreads();
        const double m = buf[lane];
#pragma unroll
        for (int jj = 0; jj < NB / 4; ++jj) {
            const int j = 4 * jj + w;
            if (j > k) {                                     // compile-time per (k, jj)
                v[jj] = fma(-m, buf[j], v[jj]);              // buf[j] = L[j][k]: one address for the whole wavefront
                asm volatile("" : "+v"(v[jj]));
            }
        }
    }
    if (lane == 0 && notpd) {
        if (kSc1) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *bad = 1;
    }
    if (live) {
        double *Arow = A + (int64_t)(t0 + r) * n + t0;
#pragma unroll
        for (int jj = 0; jj < NB / 4; ++jj) {
            const int j = 4 * jj + w;
            if (j < nb2 && (upper ? j > r : j <= r)) {                       // upper half: v = inv(L)[j][r], j > r
                if (kSc1) __hip_atomic_store(Arow + j, v[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else Arow[j] = v[jj];
            }
        }
    }
}

template <bool kSc1 = false>
__device__ __forceinline__ void factor_diag_block_from_lds_4w(const double *sT, double *__restrict__ A, int n, int t0,
                                                              int *__restrict__ bad, int tid, double *sM)
{
    // (the four instances reach their barriers at different addresses: s_barrier counts arrivals, not program counters)
    switch (tid >> 6) {
    case 0: factor_diag_block_4w_wave<0, kSc1>(sT, A, n, t0, bad, tid & 63, sM); break;
    case 1: factor_diag_block_4w_wave<1, kSc1>(sT, A, n, t0, bad, tid & 63, sM); break;
    case 2: factor_diag_block_4w_wave<2, kSc1>(sT, A, n, t0, bad, tid & 63, sM); break;
    default: factor_diag_block_4w_wave<3, kSc1>(sT, A, n, t0, bad, tid & 63, sM); break;
    }
}

// inv(L_kk) of the diagonal block at origin k0 (nb live rows) into LDS as a plain lower-triangular 32 x 32 matrix
// (sLi[j * kLd + k] = inv(L)[j][k], zero above the diagonal), from the layout above.
__device__ __forceinline__ void load_inv_diag_block(const double *__restrict__ A, int n, int k0, int nb, double *sLi, int tid,
                                                    int nthreads)
{
    for (int e = tid; e < NB * NB; e += nthreads) {
        const int j = e >> 5, k = e & 31;
        double v = 0.0;
        if (j < nb && k < nb) {
            if (k < j) v = A[(int64_t)(k0 + k) * n + k0 + j];
            else if (k == j) v = 1.0 / A[(int64_t)(k0 + j) * n + k0 + j];
        }
        sLi[j * kLd + k] = v;
    }
}

}  // namespace chol
}  // namespace mqs
