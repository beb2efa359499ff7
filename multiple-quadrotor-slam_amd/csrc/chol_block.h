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

// The same 32 x 32 factorisation -- L in the lower triangle, inv(L)'s strictly lower part transposed in the strictly upper one -- by ONE
// wavefront in panels of FOUR pivots, the tile living in the fp64 matrix pipe's accumulator layout (round 6; slam_ba.hip's diagonal
// tiles: the factor sits on the critical path of every block column of the resident adjuster's Cholesky).
//
// What the four-wave form above pays per pivot is a workgroup barrier and an LDS round trip (6.3 us per tile: ~470 cycles per pivot, of
// which the pivot's own arithmetic -- a reciprocal square root and a multiplication -- is a fifth).  Here a panel of four columns
//   A. goes from the lanes that hold it (accumulator layout: register v of lane l = row l / 16 + 4 v, column l % 16 of a 16 x 16
//      quadrant) to LDS and comes back with one ROW per lane;
//   B. every lane factors the panel's 4 x 4 diagonal block for itself (ten values, four dependent reciprocal square roots) and
//      substitutes its own row: P = A(:, panel) inv(L44)^T -- the rows of L, final, stored at once;
//   C. P goes through LDS into the matrix pipe's operand order, and the trailing update A -= P P^T is three 16 x 16 x 4 instructions
//      (one behind column 16): exactly the instruction's shape;
//   D. the inverse rides along as the forward substitution of the identity: the four rows of the residual that belong to the panel ARE
//      one accumulator register in the B-operand order (row = l / 16, column = l % 16), every lane gathers the four of its column
//      (ds_bpermute), solves the 4 x 4 system, keeps its row -- a row of inv(L), final, stored transposed -- and the rows below
//      are updated by two more matrix instructions.
// Eight panels, no barrier.  MEASURED (round 6, slam_ba.hip on the example sequence): 7.7 us per tile against the four-wave form's 6.3 --
// ~0.96 us of dependent chain per panel -- so the resident adjuster keeps the four-wave form; this one is its A/B twin and tested by itself
// (mqs_debug_factor32).  Not bit-identical to the forms above (the trailing updates add four products at a time); same reciprocal square
// roots, same rounding-level result.
// sT: LDS [32][kLd], lower triangle valid.  A: global, 32 x 32 row-major.  sW: 256 doubles of LDS for this wavefront alone.
template <bool kSc1 = false>
__device__ __forceinline__ void factor_diag_block_mfma_wave(const double *sT, double *__restrict__ A, int *__restrict__ bad, int lane,
                                                            double *sW)
{
    const int jl = lane & 15, g = lane >> 4, r = lane & 31;
    auto put = [&](double *p, double v) {
        if (kSc1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
    double4v Q00, Q10, Q11, R00, R10, R11;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = g + 4 * v, hi = i > jl ? i : jl, lo = i > jl ? jl : i;
        Q00[v] = sT[hi * kLd + lo];
        Q10[v] = sT[(16 + i) * kLd + jl];
        Q11[v] = sT[(16 + hi) * kLd + 16 + lo];
        R00[v] = i == jl ? 1.0 : 0.0;
        R10[v] = 0.0;
        R11[v] = i == jl ? 1.0 : 0.0;
    }
    double *sP = sW, *sOp = sW + 128;                       // [32][4] the panel in row order; [4][32] P in operand order
    bool notpd = false;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int c0 = 4 * p, qc = c0 >> 4, cj = c0 & 15, vq = cj >> 2;
        // A. the panel's columns, rows >= 16 qc, into row order
        mqs_wave_lds_sync();                                // (the previous panel's readers of sP are done)
        if (jl >= cj && jl < cj + 4) {
            const int kk = jl - cj;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (qc == 0) { sP[(g + 4 * v) * 4 + kk] = Q00[v]; sP[(16 + g + 4 * v) * 4 + kk] = Q10[v]; }
                else sP[(16 + g + 4 * v) * 4 + kk] = Q11[v];
            }
        }
        mqs_wave_lds_sync();
        // B. the 4 x 4 diagonal block (every lane for itself) and this lane's row of the panel
        const double a0 = sP[r * 4 + 0], a1 = sP[r * 4 + 1], a2 = sP[r * 4 + 2], a3 = sP[r * 4 + 3];
        const double d00 = sP[c0 * 4], d10 = sP[(c0 + 1) * 4], d11 = sP[(c0 + 1) * 4 + 1], d20 = sP[(c0 + 2) * 4], d21 = sP[(c0 + 2) * 4 + 1],
                     d22 = sP[(c0 + 2) * 4 + 2], d30 = sP[(c0 + 3) * 4], d31 = sP[(c0 + 3) * 4 + 1], d32 = sP[(c0 + 3) * 4 + 2],
                     d33 = sP[(c0 + 3) * 4 + 3];
        notpd = notpd || !(d00 > 0.0);
        const double i0 = mqs::rsqrt_d(d00 > 0.0 ? d00 : 1.0);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double e11 = fma(-l10, l10, d11);
        notpd = notpd || !(e11 > 0.0);
        const double i1 = mqs::rsqrt_d(e11 > 0.0 ? e11 : 1.0);
        const double l21 = fma(-l20, l10, d21) * i1, l31 = fma(-l30, l10, d31) * i1;
        const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
        notpd = notpd || !(e22 > 0.0);
        const double i2 = mqs::rsqrt_d(e22 > 0.0 ? e22 : 1.0);
        const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * i2;
        const double e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
        notpd = notpd || !(e33 > 0.0);
        const double i3 = mqs::rsqrt_d(e33 > 0.0 ? e33 : 1.0);
        const int m = r - c0;                               // this row's place against the panel: < 0 finished, 0..3 inside the diagonal block
        double p0 = a0 * i0;
        double p1 = fma(-p0, l10, a1) * i1;
        double p2 = fma(-p1, l21, fma(-p0, l20, a2)) * i2;
        double p3 = fma(-p2, l32, fma(-p1, l31, fma(-p0, l30, a3))) * i3;
        p0 = m >= 0 ? p0 : 0.0; p1 = m >= 1 ? p1 : 0.0; p2 = m >= 2 ? p2 : 0.0; p3 = m >= 3 ? p3 : 0.0;
        if (lane < 32) {
            double *Arow = A + r * NB + c0;                 // rows of L: final
            if (m >= 0) put(Arow + 0, p0);
            if (m >= 1) put(Arow + 1, p1);
            if (m >= 2) put(Arow + 2, p2);
            if (m >= 3) put(Arow + 3, p3);
            // C. P in operand order
            sOp[0 * 32 + r] = p0; sOp[1 * 32 + r] = p1; sOp[2 * 32 + r] = p2; sOp[3 * 32 + r] = p3;
        }
        mqs_wave_lds_sync();
        const double op0 = sOp[g * 32 + jl], op1 = sOp[g * 32 + 16 + jl];       // P[jl][g], P[16 + jl][g]
        if (qc == 0) {
            Q00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-op0, op0, Q00, 0, 0, 0);
            Q10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-op1, op0, Q10, 0, 0, 0);
        }
        Q11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-op1, op1, Q11, 0, 0, 0);
        // D. the inverse: rows c0 .. c0 + 3 of the residual (register vq: row g of the panel, column jl / 16 + jl)
        const double rp0 = qc == 0 ? R00[vq] : R10[vq];     // columns 0 .. 15
        const double rp1 = qc == 0 ? 0.0 : R11[vq];         // columns 16 .. 31 (zero above row 16: the inverse is lower triangular)
        double x0, x1;
        {
            const double r0 = __shfl(rp0, jl, 64), r1 = __shfl(rp0, 16 + jl, 64), r2 = __shfl(rp0, 32 + jl, 64), r3 = __shfl(rp0, 48 + jl, 64);
            const double y0 = r0 * i0;
            const double y1 = fma(-l10, y0, r1) * i1;
            const double y2 = fma(-l21, y1, fma(-l20, y0, r2)) * i2;
            const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, r3))) * i3;
            x0 = g == 0 ? y0 : (g == 1 ? y1 : (g == 2 ? y2 : y3));
        }
        if (qc == 1) {
            const double r0 = __shfl(rp1, jl, 64), r1 = __shfl(rp1, 16 + jl, 64), r2 = __shfl(rp1, 32 + jl, 64), r3 = __shfl(rp1, 48 + jl, 64);
            const double y0 = r0 * i0;
            const double y1 = fma(-l10, y0, r1) * i1;
            const double y2 = fma(-l21, y1, fma(-l20, y0, r2)) * i2;
            const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, r3))) * i3;
            x1 = g == 0 ? y0 : (g == 1 ? y1 : (g == 2 ? y2 : y3));
        } else {
            x1 = 0.0;
        }
        // row c0 + g of inv(L), columns jl and 16 + jl: stored transposed into the strictly upper triangle (column < row only)
        if (jl < c0 + g) put(A + jl * NB + c0 + g, x0);
        if (qc == 1 && 16 + jl < c0 + g) put(A + (16 + jl) * NB + c0 + g, x1);
        // the rows below the panel: residual -= P X
        const double om0 = jl > c0 + 3 ? op0 : 0.0, om1 = 16 + jl > c0 + 3 ? op1 : 0.0;
        if (qc == 0) {
            R00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-om0, x0, R00, 0, 0, 0);
            R10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-om1, x0, R10, 0, 0, 0);
        } else {
            R10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-om1, x0, R10, 0, 0, 0);
            R11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-om1, x1, R11, 0, 0, 0);
        }
    }
    if (lane == 0 && notpd) {
        if (kSc1) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *bad = 1;
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
