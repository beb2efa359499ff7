// Banded Cholesky solve of the reduced camera system with the band CUT INTO INDEPENDENT CHUNKS (nested dissection of a
// band): ba_sparse.hip factors the band in natural order, one launch per 32-column block -- 166 dependent launches at
// n = 5286, each about 16 us of which half is one wavefront factoring a 32 x 32 block; nothing inside a step is left to
// shorten, so the number of DEPENDENT steps has to come down.  Rows further apart than the half bandwidth are not coupled:
// w = ceil(hb / 32) consecutive blocks form a separator that splits the band into two halves with no coupling between
// them, recursively.  The leaves (chunks) are eliminated side by side, one launch per block column of ALL chunks, then the
// separators level by level (those of one level are independent too).  With 8 chunks the 166 steps become 18 + 3 x 4.
//
// What this costs is fill: a chunk's Schur complement couples its left and its right separator, so the block column of a
// chunk block holds up to w chunk blocks plus both separators (3 w blocks instead of w).  The structure of every block
// column of L is found on the host by a symbolic factorisation of the block graph (166 nodes: microseconds, cached per
// (n, hb)), and the kernels walk those lists; nothing in them knows about bands.
//
// Two workgroups of one launch must never update the same tile, or the summation order -- and with it the last bits of
// the result -- would depend on scheduling.  Updates a pivot block makes to tiles of ITS OWN front (chunk or separator)
// are applied right away (right-looking), one writer per tile.  Updates to tiles of a LATER front -- the separator x
// separator tiles that the two neighbouring chunks both contribute to -- are deferred: before that front's stage one
// "lazy" launch forms, per tile, the sum over all contributing block columns in elimination order (left-looking).  The
// right-hand side is treated the same way in the forward substitution.  Results are bit-reproducible.
//
// Storage: S dense row-major n x n with BOTH triangles valid on entry (the linearisers write both).  "L slot" of a
// block pair (X later than k in elimination order) is A[X rows][k columns] wherever that lies in memory; the mirror
// A[k rows][X columns] carries the not-yet-scaled panel input (see chol_step_kernel in ba_sparse.hip: same trick).
// Diagonal blocks: L below, inv(L)^T strictly above (chol_block.h).
#include "chol_block.h"
#include <algorithm>
#include <list>
#include <memory>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace mqs {
namespace chol {
namespace {

constexpr int kMaxList = 28;            // blocks below the diagonal in one block column of L
constexpr int kThreads = 256;
constexpr int kVecThreads = 1024;

struct StepDesc {                       // one block column of one front in one launch (32 ints)
    int32_t kblk;                       // pivot block, -1: this front has finished
    int32_t cnt;                        // blocks in the column's structure
    int32_t eager;                      // the first `eager` of them belong to the pivot's own front (kblk + 1, ...)
    int32_t tiles;                      // workgroups this column needs in the factor step
    int32_t blk[kMaxList];              // the structure, in elimination order
};
static_assert(sizeof(StepDesc) == 128, "StepDesc layout");

struct LazyTile {                       // tile (x1, x2): contributions contrib[start .. start + count), summed kLazySlice at a time into
    int32_t x1, x2, start, count;       // partial tiles scratch[slot0 .. slot0 + nslots) (slots counted per stage)
    int32_t factor, slot0, nslots, pad;
};
struct LazyVec { int32_t x, start, count, pad; };

// ---------------------------------------------------------------------------------------------------------------------
// Input contract = the LOWER triangle of S (natural indexing).  The elimination order of the plan is not the natural one,
// so the kernels below read and write tiles on both sides of the diagonal; every tile they ever touch is a (pivot block,
// structure block) pair of some block column, so one pass over the descriptors makes both images of every such tile valid:
// the one above the diagonal becomes the transpose of the one below (zero fill included), diagonal tiles are symmetrised.
// Workgroup (d, 0): the diagonal tile of column d; (d, 1 + i): the tile of structure entry i.  One writer per tile.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nd_mirror_kernel(double *__restrict__ A, int n, const StepDesc *__restrict__ descs)
{
    const StepDesc &d = descs[blockIdx.x];
    if (d.kblk < 0) return;
    const int y = blockIdx.y;
    if (y > d.cnt) return;
    const int other = (y == 0) ? d.kblk : d.blk[y - 1];
    const int lo = other > d.kblk ? other : d.kblk, hi = other > d.kblk ? d.kblk : other;     // tile (lo, hi) is below the diagonal
    __shared__ double sT[NB][NB + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = lo * NB + ty + 8 * k, c = hi * NB + tx;
        sT[ty + 8 * k][tx] = (r < n && c < n) ? A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = hi * NB + ty + 8 * k, r = lo * NB + tx;          // element (c, r) above the diagonal = element (r, c) below it
        if (r < n && c < n && r > c) A[(int64_t)c * n + r] = sT[tx][ty + 8 * k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// factor step: panel + eager trailing update of one block column per front, and the front's next diagonal block
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nd_step_kernel(double *__restrict__ A, int n, const StepDesc *__restrict__ descs,
                                                           int *__restrict__ bad, int nfronts, int max_tiles, int xcd_pin)
{
    // front and tile of this workgroup.  Workgroups go to the eight XCDs round-robin by their linear id, and an XCD's L2 is its
    // own: with the fronts a multiple of 8, all tiles of a front are given ids of ONE residue, so what a front's step wrote is
    // read back by its next step from the same L2 (MQS_ND_XCD_PIN, A/B in DESIGN.md).
    int f, t;
    if (xcd_pin) {
        const int L = blockIdx.x, xcd = L & 7, seq = L >> 3;
        f = xcd + 8 * (seq / max_tiles);
        t = seq % max_tiles;
        if (f >= nfronts) return;
    } else {
        f = blockIdx.y;
        t = blockIdx.x;
    }
    const StepDesc &d = descs[f];
    if (d.kblk < 0 || d.cnt == 0) return;
    if (t >= d.tiles) return;
    const bool do_update = d.eager > 0;
    const int ecols = do_update ? d.eager : 1;          // a column without own-front dependants still has its panel to write
    int bj = 0;
    for (; bj < ecols; ++bj) {
        const int len = d.cnt - bj;
        if (t < len) break;
        t -= len;
    }
    const int bi = bj + t;
    const int k0 = d.kblk * NB;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int i0 = d.blk[bi] * NB, j0 = d.blk[bj] * NB;
    __shared__ double sLi[NB][kLd], sAi[NB][kLd], sAj[NB][kLd], sXi[NB][kLd], sXj[NB][kLd];
    __shared__ double sM[128];                              // pivot columns of the four-wave diagonal factorisation
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += kThreads) {
        const int a = e / NB, b = e % NB;
        double v = 0.0;
        if (a < nb && b < nb) v = (b < a) ? A[(int64_t)(k0 + b) * n + k0 + a] : ((b == a) ? 1.0 / A[(int64_t)(k0 + a) * n + k0 + a] : 0.0);
        sLi[a][b] = v;
        // panel input from the mirror: S[X row][k column] = A[k0 + a][X0 + b]
        sAi[b][a] = (a < nb && i0 + b < n) ? A[(int64_t)(k0 + a) * n + i0 + b] : 0.0;
        sAj[b][a] = (a < nb && j0 + b < n) ? A[(int64_t)(k0 + a) * n + j0 + b] : 0.0;
    }
    __syncthreads();
    const int lane = tid & 63, wr = (tid >> 6) >> 1, wc = (tid >> 6) & 1;
    {
        // X = A inv(L)^T for both blocks on the matrix pipe (inv(L) is lower triangular: its zeros are multiplied along)
        const double4v xi = tile_quadrant_mfma(&sAi[0][0], &sLi[0][0], wr, wc, lane);
        const double4v xj = (bi == bj) ? xi : tile_quadrant_mfma(&sAj[0][0], &sLi[0][0], wr, wc, lane);
        const int c = quadrant_col(wc, lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = quadrant_row(wr, lane, v);
            sXi[r][c] = xi[v];
            sXj[r][c] = xj[v];
            if (bj == 0 && i0 + r < n && c < nb) A[(int64_t)(i0 + r) * n + k0 + c] = xi[v];      // final entries of L
        }
    }
    if (!do_update) return;
    __syncthreads();
    const double4v acc = tile_quadrant_mfma(&sXi[0][0], &sXj[0][0], wr, wc, lane);
    const bool next_diag = bi == 0 && bj == 0;                  // blk[0] = kblk + 1: the front's next pivot block
    double *sT = &sAi[0][0];
    __syncthreads();
    {
        const int tc = quadrant_col(wc, lane), c = j0 + tc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int tr = quadrant_row(wr, lane, v), r = i0 + tr;
            if (next_diag) {
                if (r < n && c <= r) sT[tr * kLd + tc] = A[(int64_t)r * n + c] - acc[v];
            } else if (r < n && c < n && (bi != bj || c <= r)) {
                const double val = A[(int64_t)r * n + c] - acc[v];
                A[(int64_t)r * n + c] = val;
                if (bi != bj) A[(int64_t)c * n + r] = val;
            }
        }
    }
    if (!next_diag) return;
    __syncthreads();
    factor_diag_block_from_lds_4w(sT, A, n, i0, bad, tid, sM);
}

// ---------------------------------------------------------------------------------------------------------------------
// lazy tiles: A[x1][x2] -= sum_k L[x1][k] L[x2][k]^T over the block columns of EARLIER fronts, in elimination order; the
// first diagonal block of a front is factored on the spot (factor = 1)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLazySlice = 4;             // contributions per workgroup of the first lazy launch

// Two launches.  A tile's sum runs over up to ~40 block columns, and the tiles of the upper levels are few (10 at the
// root): one workgroup per tile walked them one after the other on 10 of 256 compute units -- 67 us at the root, bound by
// the LDS bandwidth of that compute unit (the 2 x 2 register tile reads four operands per four FMAs).  So the sum is cut
// into slices of kLazySlice block columns, one workgroup each, all loads of a slice in flight at once; the partial tiles
// go to scratch memory and the second launch adds them IN SLICE ORDER (the result does not depend on scheduling),
// applies the sum and factors the fronts' first diagonal blocks.
__global__ __launch_bounds__(kThreads) void nd_lazy_part_kernel(const double *__restrict__ A, int n, const LazyTile *__restrict__ tiles,
                                                                const int32_t *__restrict__ slot_tile,
                                                                const int32_t *__restrict__ contrib, double *__restrict__ scratch)
{
    const int slot = blockIdx.x;
    const LazyTile T = tiles[slot_tile[slot]];
    const int q0 = (slot - T.slot0) * kLazySlice;
    const int i0 = T.x1 * NB, j0 = T.x2 * NB;
    const bool diag = T.x1 == T.x2;
    __shared__ double sI[NB][kLd], sJ[NB][kLd];
    const int tid = threadIdx.x;
    double pi[kLazySlice][4], pj[kLazySlice][4];
#pragma unroll
    for (int bq = 0; bq < kLazySlice; ++bq) {
        const int q = q0 + bq;
        const bool live = q < T.count;
        const int k0 = live ? contrib[T.start + q] * NB : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * kThreads, r = e >> 5, c = e & 31;
            const bool col = live && k0 + c < n;
            pi[bq][u] = (col && i0 + r < n) ? A[(int64_t)(i0 + r) * n + k0 + c] : 0.0;
            pj[bq][u] = (!diag && col && j0 + r < n) ? A[(int64_t)(j0 + r) * n + k0 + c] : 0.0;
        }
    }
    const int lane = tid & 63, wr = (tid >> 6) >> 1, wc = (tid >> 6) & 1;
    double4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int bq = 0; bq < kLazySlice; ++bq) {
        if (q0 + bq >= T.count) break;                        // uniform over the workgroup
        if (bq > 0) __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * kThreads, r = e >> 5, c = e & 31;
            sI[r][c] = pi[bq][u];
            sJ[r][c] = diag ? pi[bq][u] : pj[bq][u];
        }
        __syncthreads();
        acc = tile_quadrant_mfma(&sI[0][0], &sJ[0][0], wr, wc, lane, acc);
    }
    double *out = scratch + (size_t)slot * NB * NB;
    const int c = quadrant_col(wc, lane);
#pragma unroll
    for (int v = 0; v < 4; ++v) out[quadrant_row(wr, lane, v) * NB + c] = acc[v];
}

constexpr int kLazyMaxSlots = 16;         // partial tiles one thread keeps in flight in the second launch

__global__ __launch_bounds__(kThreads) void nd_lazy_apply_kernel(double *__restrict__ A, int n, const LazyTile *__restrict__ tiles,
                                                                 const double *__restrict__ scratch, int *__restrict__ bad)
{
    const LazyTile T = tiles[blockIdx.x];
    const int i0 = T.x1 * NB, j0 = T.x2 * NB;
    const bool diag = T.x1 == T.x2;
    __shared__ double sT[NB * kLd];
    __shared__ double sM[128];
    const int tid = threadIdx.x;
    const int tr = (tid / 16) * 2, tc = (tid % 16) * 2;
    double cur[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = i0 + tr + a, c = j0 + tc + b;
            cur[a][b] = (r < n && c < n && (!diag || c <= r)) ? A[(int64_t)r * n + c] : 0.0;
        }
    double2 p0[kLazyMaxSlots], p1[kLazyMaxSlots];
    const double2 *in = reinterpret_cast<const double2 *>(scratch + (size_t)T.slot0 * NB * NB);
#pragma unroll
    for (int u = 0; u < kLazyMaxSlots; ++u) {
        p0[u] = make_double2(0.0, 0.0);
        p1[u] = make_double2(0.0, 0.0);
        if (u < T.nslots) {
            p0[u] = in[((size_t)u * NB * NB + tr * NB + tc) / 2];
            p1[u] = in[((size_t)u * NB * NB + (tr + 1) * NB + tc) / 2];
        }
    }
    double acc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
    for (int u = 0; u < kLazyMaxSlots; ++u) {               // slice order: zeros beyond nslots change nothing
        acc[0][0] += p0[u].x; acc[0][1] += p0[u].y; acc[1][0] += p1[u].x; acc[1][1] += p1[u].y;
    }
    for (int u = kLazyMaxSlots; u < T.nslots; ++u) {         // very long sums (deep cuts of wide bands)
        const double2 q0 = in[((size_t)u * NB * NB + tr * NB + tc) / 2], q1 = in[((size_t)u * NB * NB + (tr + 1) * NB + tc) / 2];
        acc[0][0] += q0.x; acc[0][1] += q0.y; acc[1][0] += q1.x; acc[1][1] += q1.y;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = i0 + tr + a, c = j0 + tc + b;
            const double v = cur[a][b] - acc[a][b];
            if (T.factor) {
                if (r < n && c <= r) sT[(tr + a) * kLd + tc + b] = v;
            } else if (r < n && c < n && (!diag || c <= r)) {
                A[(int64_t)r * n + c] = v;
                if (!diag) A[(int64_t)c * n + r] = v;
            }
        }
    if (!T.factor) return;
    __syncthreads();
    factor_diag_block_from_lds_4w(sT, A, n, i0, bad, tid, sM);
}

// ---------------------------------------------------------------------------------------------------------------------
// substitution.  Forward, per stage: the deferred part  b_X -= sum_k L[X][k] y_k  (k in earlier fronts), then every front
// walks its own block columns with its unknowns in LDS.  Backward needs no deferral: a block column only READS later
// unknowns.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kVecThreads) void nd_fwd_lazy_kernel(const double *__restrict__ A, int n, double *__restrict__ x,
                                                                  const LazyVec *__restrict__ vecs,
                                                                  const int32_t *__restrict__ cols)
{
    const LazyVec V = vecs[blockIdx.x];
    const int tid = threadIdx.x, r = tid >> 5, c = tid & 31;
    const int row = V.x * NB + r;
    double s = 0.0;
    for (int q0 = 0; q0 < V.count; q0 += 8) {                // eight columns' loads in flight at a time: the loop is pure latency
        double a[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u;
            const int k0 = (q < V.count) ? cols[V.start + q] * NB : 0;
            const bool live = q < V.count && row < n && k0 + c < n;
            a[u] = live ? A[(int64_t)row * n + k0 + c] : 0.0;
            v[u] = live ? x[k0 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fma(a[u], v[u], s);
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (c == 0 && row < n) x[row] -= s;
}

constexpr int kMaxEagerPre = 8;          // panel entries per thread held in flight by the forward kernel (eager blocks <= 8)

__global__ __launch_bounds__(kVecThreads) void nd_fwd_front_kernel(const double *__restrict__ A, int n, double *__restrict__ x,
                                                                   const StepDesc *__restrict__ descs, int nfronts, int nsteps)
{
    extern __shared__ double sMem[];
    double *sLi = sMem;                          // [NB][kLd]
    double *sP = sLi + NB * kLd;                 // [kMaxEagerPre * NB][kLd] rows of the own-front dependants
    double *sX = sP + kMaxEagerPre * NB * kLd;   // the front's unknowns
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = descs[f].kblk;
    int len = 0;
    while (len < nsteps && descs[(size_t)len * nfronts + f].kblk >= 0) ++len;
    for (int i = tid; i < len * NB; i += kVecThreads) sX[i] = (first * NB + i < n) ? x[first * NB + i] : 0.0;
    double pLi = 0.0, pP[kMaxEagerPre];
    auto fetch = [&](int t) {
        const StepDesc &d = descs[(size_t)t * nfronts + f];
        const int k0 = d.kblk * NB;
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int j = tid >> 5, k = tid & 31;
        pLi = 0.0;
        if (j < nb && k <= j) pLi = A[(int64_t)(k0 + k) * n + k0 + j];      // k == j: the diagonal itself, inverted at commit
#pragma unroll
        for (int u = 0; u < kMaxEagerPre; ++u) {
            pP[u] = 0.0;
            if (u < d.eager) {
                const int row = d.blk[u] * NB + j;
                if (row < n && k < nb) pP[u] = A[(int64_t)row * n + k0 + k];
            }
        }
    };
    if (len > 0) fetch(0);
    for (int t = 0; t < len; ++t) {
        const StepDesc &d = descs[(size_t)t * nfronts + f];
        const int k0 = d.kblk * NB;
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int eager = d.eager;
        {
            const int j = tid >> 5, k = tid & 31;
            sLi[j * kLd + k] = (j < nb && k == j) ? 1.0 / pLi : pLi;
#pragma unroll
            for (int u = 0; u < kMaxEagerPre; ++u)
                if (u < eager) sP[(u * NB + j) * kLd + k] = pP[u];
        }
        __syncthreads();
        if (t + 1 < len) fetch(t + 1);
        double *xk = sX + (d.kblk - first) * NB;
        if (wave == 0) {
            const int j = lane & 31;
            double s = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) s = fma(sLi[j * kLd + k], xk[k], s);      // rows / columns >= nb of inv(L) are zero
            mqs_wave_lds_sync();
            if (lane < NB) xk[lane] = (lane < nb) ? s : 0.0;
        }
        __syncthreads();
        if (tid < eager * NB) {                          // own-front dependants: blocks kblk + 1 .. kblk + eager
            double s = 0.0;
#pragma unroll 8
            for (int c = 0; c < NB; ++c) s = fma(sP[tid * kLd + c], xk[c], s);
            xk[NB + tid] -= s;
        }
        __syncthreads();
    }
    for (int i = tid; i < len * NB; i += kVecThreads)
        if (first * NB + i < n) x[first * NB + i] = sX[i];
}

constexpr int kBwdPre = 12;              // factor entries (and ancestors' unknowns) per thread held in flight by the backward kernel

__global__ __launch_bounds__(kVecThreads) void nd_bwd_front_kernel(const double *__restrict__ A, int n, double *__restrict__ x,
                                                                   const StepDesc *__restrict__ descs, int nfronts, int nsteps)
{
    extern __shared__ double sMem[];
    double *sLi = sMem;                          // [NB][kLd]
    double *sQ = sLi + NB * kLd;                 // [NB][kLd] partial sums
    double *sCol = sQ + NB * kLd;                // [64]
    double *sX = sCol + 64;                      // the front's unknowns
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = descs[f].kblk;
    int len = 0;
    while (len < nsteps && descs[(size_t)len * nfronts + f].kblk >= 0) ++len;
    for (int i = tid; i < len * NB; i += kVecThreads) sX[i] = (first * NB + i < n) ? x[first * NB + i] : 0.0;
    const int part = tid >> 5, c = tid & 31;
    // Neither the factor entries of a block column nor the unknowns of LATER fronts depend on this front's progress: both are
    // fetched one step ahead into registers (the step itself is three barriers and two 32-long dot products; its global-memory
    // latency, about 3 us of 4.6, leaves the serial chain).  Own-front unknowns come from LDS at the time of use.
    double pLi = 0.0, pL[kBwdPre], pX[kBwdPre];
    auto fetch = [&](int t) {
        const StepDesc &d = descs[(size_t)t * nfronts + f];
        const int k0 = d.kblk * NB;
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        pLi = 0.0;
        if (part < nb && c <= part) pLi = A[(int64_t)(k0 + c) * n + k0 + part];     // c == part: the diagonal itself
#pragma unroll
        for (int u = 0; u < kBwdPre; ++u) {
            pL[u] = 0.0;
            pX[u] = 0.0;
            if (u < d.cnt) {
                const int row = d.blk[u] * NB + part;
                if (row < n && c < nb) {
                    pL[u] = A[(int64_t)row * n + k0 + c];
                    if (u >= d.eager) pX[u] = x[row];
                }
            }
        }
    };
    if (len > 0) fetch(len - 1);
    __syncthreads();
    for (int t = len - 1; t >= 0; --t) {
        const StepDesc &d = descs[(size_t)t * nfronts + f];
        const int k0 = d.kblk * NB;
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        sLi[part * kLd + c] = (part < nb && c == part) ? 1.0 / pLi : pLi;
        // t_c = sum over the column's structure of L[X][k](r, c) x_X[r]; thread (part = r, c), the 32 parts combined by wave 0
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < kBwdPre; ++u)
            if (u < d.cnt) {
                const double xv = (u < d.eager) ? sX[(d.blk[u] - first) * NB + part] : pX[u];
                s = fma(pL[u], xv, s);
            }
        for (int b = kBwdPre; b < d.cnt; ++b) {              // structures longer than the prefetch window (wide bands)
            const int X = d.blk[b];
            const int row = X * NB + part;
            if (row < n && c < nb) {
                const double xv = (b < d.eager) ? sX[(X - first) * NB + part] : x[row];
                s = fma(A[(int64_t)row * n + k0 + c], xv, s);
            }
        }
        sQ[part * kLd + c] = s;
        if (t > 0) fetch(t - 1);
        __syncthreads();
        double *xk = sX + (d.kblk - first) * NB;
        if (wave == 0) {
            const int cc = lane & 31;
            double tsum = 0.0;
#pragma unroll 8
            for (int p2 = 0; p2 < NB; ++p2) tsum += sQ[p2 * kLd + cc];
            const double yc = (cc < nb) ? xk[cc] - tsum : 0.0;
            if (lane < NB) sCol[lane] = yc;
            mqs_wave_lds_sync();
            double r = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) r = fma(sLi[k * kLd + cc], sCol[k], r);     // inv(L)^T: zero for k < c
            if (lane < NB) xk[lane] = (lane < nb) ? r : 0.0;
        }
        __syncthreads();
    }
    for (int i = tid; i < len * NB; i += kVecThreads)
        if (first * NB + i < n) x[first * NB + i] = sX[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// the plan: elimination order, structure of L's block columns, what is deferred -- host side, cached
// ---------------------------------------------------------------------------------------------------------------------
struct StagePlan {
    int nfronts = 0, nsteps = 0;
    size_t desc_off = 0;                 // StepDesc index of (step 0, front 0); step t, front f at desc_off + t * nfronts + f
    std::vector<int> max_tiles;          // per step
    size_t lazy_off = 0; int nlazy = 0;
    size_t slot_off = 0; int nslots = 0;  // partial tiles of the stage's lazy sums (index into slot_tile)
    size_t vec_off = 0; int nvec = 0;
    int max_len = 0;                     // longest front, blocks
};

struct Plan {
    bool usable = false;
    int n = 0, hb = 0, parts = 0;
    std::vector<StagePlan> stages;
    std::vector<StepDesc> descs;         // host copies (uploaded once; kept for mqs_sba_solve_plan_dump)
    std::vector<LazyTile> lazy;
    std::vector<int32_t> contrib;
    std::vector<int32_t> slot_tile;      // per stage: which lazy tile (stage-relative) a partial tile belongs to
    std::vector<LazyVec> vecs;
    std::vector<int32_t> cols;
    StepDesc *d_descs = nullptr;
    LazyTile *d_lazy = nullptr;
    int32_t *d_contrib = nullptr;
    int32_t *d_slot_tile = nullptr;
    double *d_scratch = nullptr;
    LazyVec *d_vecs = nullptr;
    int32_t *d_cols = nullptr;
    int launches = 0;
};

struct Front { int a, b, stage; };

void split_range(int a, int b, int depth, int max_depth, int w, std::vector<Front> &fronts, int &height)
{
    const int len = b - a;
    if (depth >= max_depth || len < 3 * w + 2) {
        fronts.push_back({a, b, 0});
        height = 0;
        return;
    }
    const int m = a + (len - w) / 2;
    int hl = 0, hr = 0;
    split_range(a, m, depth + 1, max_depth, w, fronts, hl);
    split_range(m + w, b, depth + 1, max_depth, w, fronts, hr);
    height = 1 + std::max(hl, hr);
    fronts.push_back({m, m + w, height});
}

// parts = 0: choose; 1: do not cut
int choose_depth(int nblk, int w, int parts)
{
    if (parts == 1) return 0;
    int depth = 0;
    if (parts > 1) {
        while ((1 << (depth + 1)) <= parts) ++depth;
        return depth;
    }
    // dependent launches: one per block column of the longest leaf, then w per separator level, each about as long as the
    // next (one wavefront factoring a 32 x 32 block is the critical path of a step whatever its width); a level also costs
    // a deferred-update launch, about 2.5 steps.  The deepest cut is not the best one: leaves shorter than a separator only
    // add levels.
    double best = (double)nblk;
    for (int d = 1; d <= 7; ++d) {
        const int p = 1 << d;
        const int leaf = (nblk - (p - 1) * w + p - 1) / p;
        if (nblk - (p - 1) * w < 2 * p || leaf < 2) break;
        const double cost = leaf + d * (w + 2.5);
        if (cost < best) { best = cost; depth = d; }
    }
    return depth;
}

bool build_plan_host(Plan &plan, int n, int hb, int parts)
{
    const int nblk = (n + NB - 1) / NB;
    const int w = hb <= 0 ? 0 : (hb - 1) / NB + 1;              // blocks a, b are coupled iff |a - b| <= w
    if (w == 0 || nblk < 3 * w + 2) return false;
    const int depth = choose_depth(nblk, w, parts);
    if (depth == 0) return false;
    std::vector<Front> fronts;
    int height = 0;
    split_range(0, nblk, 0, depth, w, fronts, height);
    if (height == 0) return false;
    std::stable_sort(fronts.begin(), fronts.end(), [](const Front &p, const Front &q) {
        return p.stage != q.stage ? p.stage < q.stage : p.a < q.a;
    });
    const int nstages = height + 1;
    std::vector<int> pos(nblk, -1), front_of(nblk, -1), order;
    for (size_t f = 0; f < fronts.size(); ++f)
        for (int b = fronts[f].a; b < fronts[f].b; ++b) {
            pos[b] = (int)order.size();
            front_of[b] = (int)f;
            order.push_back(b);
        }
    if ((int)order.size() != nblk) return false;
    // symbolic factorisation on blocks
    std::vector<std::vector<int>> st(nblk);
    for (int a = 0; a < nblk; ++a)
        for (int b = std::max(0, a - w); b <= std::min(nblk - 1, a + w); ++b)
            if (b != a && pos[b] > pos[a]) st[a].push_back(b);
    auto by_pos = [&](int p, int q) { return pos[p] < pos[q]; };
    for (int k : order) {
        auto &s = st[k];
        std::sort(s.begin(), s.end(), by_pos);
        s.erase(std::unique(s.begin(), s.end()), s.end());
        if ((int)s.size() > kMaxList) return false;
        if (!s.empty()) {
            auto &par = st[s[0]];
            par.insert(par.end(), s.begin() + 1, s.end());
        }
    }
    // per stage: step descriptors
    std::vector<StepDesc> &descs = plan.descs;
    std::vector<LazyTile> &lazy = plan.lazy;
    std::vector<int32_t> &contrib = plan.contrib;
    std::vector<LazyVec> &vecs = plan.vecs;
    std::vector<int32_t> &cols = plan.cols;
    plan.stages.assign(nstages, StagePlan());
    std::vector<std::vector<int>> stage_fronts(nstages);
    for (size_t f = 0; f < fronts.size(); ++f) stage_fronts[fronts[f].stage].push_back((int)f);
    // deferred contributions: for block column k, every pair of its structure whose earlier member lies in another front
    std::map<std::pair<int, int>, std::vector<int>> tile_contrib;       // (x1, x2) -> k's, filled in elimination order
    std::vector<std::vector<int>> vec_contrib(nblk);
    for (int k : order) {
        const auto &s = st[k];
        for (size_t bj = 0; bj < s.size(); ++bj) {
            if (front_of[s[bj]] == front_of[k]) continue;
            if (fronts[front_of[s[bj]]].stage <= fronts[front_of[k]].stage) return false;      // must be a later stage
            vec_contrib[s[bj]].push_back(k);
            for (size_t bi = bj; bi < s.size(); ++bi) tile_contrib[{s[bi], s[bj]}].push_back(k);
        }
    }
    for (int sidx = 0; sidx < nstages; ++sidx) {
        StagePlan &sp = plan.stages[sidx];
        const auto &fl = stage_fronts[sidx];
        sp.nfronts = (int)fl.size();
        if (sp.nfronts == 0) return false;
        for (int f : fl) sp.max_len = std::max(sp.max_len, fronts[f].b - fronts[f].a);
        sp.nsteps = sp.max_len;
        sp.desc_off = descs.size();
        sp.max_tiles.assign(sp.nsteps, 0);
        for (int t = 0; t < sp.nsteps; ++t)
            for (int f : fl) {
                StepDesc d;
                d.kblk = -1; d.cnt = 0; d.eager = 0; d.tiles = 0;
                for (int &v : d.blk) v = 0;
                const int k = fronts[f].a + t;
                if (k < fronts[f].b) {
                    d.kblk = k;
                    const auto &s = st[k];
                    d.cnt = (int)s.size();
                    for (int i = 0; i < d.cnt; ++i) {
                        d.blk[i] = s[i];
                        if (front_of[s[i]] == f) {
                            if (d.eager != i || s[i] != k + 1 + i) return false;      // own-front dependants come first, contiguous
                            ++d.eager;
                        }
                    }
                    if (d.eager > kMaxEagerPre) return false;
                    if (k + 1 < fronts[f].b && d.eager == 0) return false;              // consecutive blocks of a front are coupled
                    const int ecols = d.eager > 0 ? d.eager : 1;
                    for (int bj = 0; bj < ecols; ++bj) d.tiles += d.cnt - bj;
                    if (d.cnt == 0) d.tiles = 0;
                    sp.max_tiles[t] = std::max(sp.max_tiles[t], d.tiles);
                }
                descs.push_back(d);
            }
        // lazy tiles of this stage: every front's first diagonal block (factored there), then the deferred sums
        sp.lazy_off = lazy.size();
        sp.slot_off = plan.slot_tile.size();
        for (int f : fl) {
            for (int x2 = fronts[f].a; x2 < fronts[f].b; ++x2) {
                std::vector<int> x1s;
                x1s.push_back(x2);
                for (int v : st[x2]) x1s.push_back(v);
                for (int x1 : x1s) {
                    auto it = tile_contrib.find({x1, x2});
                    const bool first_diag = x1 == x2 && x2 == fronts[f].a;
                    if (it == tile_contrib.end() && !first_diag) continue;
                    LazyTile T = {x1, x2, (int32_t)contrib.size(), 0, first_diag ? 1 : 0, 0, 0, 0};
                    if (it != tile_contrib.end()) {
                        T.count = (int32_t)it->second.size();
                        contrib.insert(contrib.end(), it->second.begin(), it->second.end());
                        tile_contrib.erase(it);
                    }
                    T.slot0 = (int32_t)(plan.slot_tile.size() - sp.slot_off);
                    T.nslots = (T.count + kLazySlice - 1) / kLazySlice;
                    for (int q = 0; q < T.nslots; ++q) plan.slot_tile.push_back((int32_t)(lazy.size() - sp.lazy_off));
                    lazy.push_back(T);
                }
            }
        }
        sp.nlazy = (int)(lazy.size() - sp.lazy_off);
        sp.nslots = (int)(plan.slot_tile.size() - sp.slot_off);
        sp.vec_off = vecs.size();
        for (int f : fl)
            for (int x = fronts[f].a; x < fronts[f].b; ++x)
                if (!vec_contrib[x].empty()) {
                    vecs.push_back({x, (int32_t)cols.size(), (int32_t)vec_contrib[x].size(), 0});
                    cols.insert(cols.end(), vec_contrib[x].begin(), vec_contrib[x].end());
                }
        sp.nvec = (int)(vecs.size() - sp.vec_off);
    }
    if (!tile_contrib.empty()) return false;         // a deferred tile nobody picked up: the structure is not what is assumed here
    // LDS of the substitution kernels
    int max_len = 0;
    for (auto &sp : plan.stages) max_len = std::max(max_len, sp.max_len);
    if ((size_t)(NB * kLd + kMaxEagerPre * NB * kLd + max_len * NB) * 8 > 150 * 1024) return false;
    plan.n = n; plan.hb = hb; plan.parts = 1 << depth;
    return true;
}

void free_plan_device(Plan &plan)
{
    void **ptrs[] = {(void **)&plan.d_descs, (void **)&plan.d_lazy, (void **)&plan.d_contrib, (void **)&plan.d_slot_tile,
                     (void **)&plan.d_scratch, (void **)&plan.d_vecs, (void **)&plan.d_cols};
    for (void **p : ptrs) {
        if (*p) (void)hipFree(*p);                 // hipFree waits for the device: a solve still in flight on the plan finishes first
        *p = nullptr;
    }
}

bool build_plan(Plan &plan, int n, int hb, int parts)
{
    if (!build_plan_host(plan, n, hb, parts)) return false;
    const std::vector<StepDesc> &descs = plan.descs;
    const std::vector<LazyTile> &lazy = plan.lazy;
    const std::vector<int32_t> &contrib = plan.contrib;
    const std::vector<LazyVec> &vecs = plan.vecs;
    const std::vector<int32_t> &cols = plan.cols;
    auto up = [](const void *src, size_t bytes, void **dst) -> bool {
        if (bytes == 0) bytes = 8;
        if (hipMalloc(dst, bytes) != hipSuccess) { *dst = nullptr; return false; }
        return src == nullptr || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    size_t max_slots = 1;
    for (const StagePlan &sp : plan.stages) max_slots = std::max(max_slots, (size_t)sp.nslots);
    const bool ok = up(descs.data(), descs.size() * sizeof(StepDesc), (void **)&plan.d_descs) &&
                    up(lazy.data(), lazy.size() * sizeof(LazyTile), (void **)&plan.d_lazy) &&
                    up(contrib.empty() ? nullptr : contrib.data(), contrib.size() * sizeof(int32_t), (void **)&plan.d_contrib) &&
                    up(plan.slot_tile.empty() ? nullptr : plan.slot_tile.data(), plan.slot_tile.size() * sizeof(int32_t),
                       (void **)&plan.d_slot_tile) &&
                    up(nullptr, max_slots * NB * NB * sizeof(double), (void **)&plan.d_scratch) &&
                    up(vecs.empty() ? nullptr : vecs.data(), vecs.size() * sizeof(LazyVec), (void **)&plan.d_vecs) &&
                    up(cols.empty() ? nullptr : cols.data(), cols.size() * sizeof(int32_t), (void **)&plan.d_cols);
    if (!ok) {
        // out of device memory part-way: give back what was taken and clear the error, so that the natural-order solve the
        // caller falls back to does not trip over it in its own hipGetLastError()
        free_plan_device(plan);
        (void)hipGetLastError();
        return false;
    }
    plan.usable = true;
    return true;
}

std::mutex g_plan_mutex;
// Plans are cached per (device, stream, shape): a plan owns the scratch memory of its lazy sums, and solves on different
// streams may overlap.  In a SLAM session 6P grows with every keyframe, so the cache is a small LRU -- the least recently
// used plan's device memory is freed when a new shape arrives (kMaxPlans shapes alive at a time; a bundle adjustment calls
// the solve tens of times per shape, the plan costs ~1 ms of host work to rebuild).
constexpr size_t kMaxPlans = 8;
// plans are shared_ptrs: a solve holds its plan while it enqueues from it, so an eviction by another host thread (a ninth shape)
// only drops the cache's reference; the device memory goes (hipFree: waits for the device) with the last holder
struct PlanDeleter { void operator()(Plan *p) const { if (p) { free_plan_device(*p); delete p; } } };
struct PlanEntry { std::tuple<int, void *, int, int, int> key; std::shared_ptr<Plan> plan; };
std::list<PlanEntry> g_plans;                     // most recently used first

bool xcd_pin_enabled()
{
    const char *e = getenv("MQS_ND_XCD_PIN");
    return !(e && e[0] == '0');
}

int parts_from_env()
{
    const char *e = getenv("MQS_SBA_PARTS");
    if (!e || !*e) return 0;
    const int v = atoi(e);
    return v < 0 ? 0 : v;
}

std::shared_ptr<Plan> get_plan(int n, int hb, hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const int parts = parts_from_env();
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    const auto key = std::make_tuple(dev, (void *)stream, n, hb, parts);
    for (auto it = g_plans.begin(); it != g_plans.end(); ++it)
        if (it->key == key) {
            g_plans.splice(g_plans.begin(), g_plans, it);
            return g_plans.front().plan;
        }
    while (g_plans.size() >= kMaxPlans) g_plans.pop_back();
    std::shared_ptr<Plan> p(new Plan(), PlanDeleter());
    if (!build_plan(*p, n, hb, parts)) p->usable = false;
    g_plans.push_front({key, p});
    return p;
}

int plan_cache_size()
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    return (int)g_plans.size();
}

}  // namespace
}  // namespace chol
}  // namespace mqs

// Factor S in place and solve S x = b (x = b on entry) with the chunked elimination.  *done = false (and nothing
// touched) when the band is too wide or the matrix too small for a cut to pay; the caller then takes the natural order.
int mqs_chol_nd_solve(double *S, double *x, int n, int hb, int *bad, hipStream_t stream, bool *done)
{
    using namespace mqs::chol;
    *done = false;
    const std::shared_ptr<Plan> plan = get_plan(n, hb, stream);         // held until every launch below is enqueued
    if (!plan || !plan->usable) return MQS_OK;
    static mqs_lds_opt_in opt_f, opt_b;                 // per device
    MQS_HIP_CHECK(mqs_lds_opt_in_once(opt_f, reinterpret_cast<const void *>(nd_fwd_front_kernel), 150 * 1024));
    MQS_HIP_CHECK(mqs_lds_opt_in_once(opt_b, reinterpret_cast<const void *>(nd_bwd_front_kernel), 150 * 1024));
    hipLaunchKernelGGL(nd_mirror_kernel, dim3((unsigned)plan->descs.size(), kMaxList + 1), dim3(kThreads), 0, stream, S, n, plan->d_descs);
    for (const StagePlan &sp : plan->stages) {
        if (sp.nslots > 0)
            hipLaunchKernelGGL(nd_lazy_part_kernel, dim3(sp.nslots), dim3(kThreads), 0, stream, S, n, plan->d_lazy + sp.lazy_off,
                               plan->d_slot_tile + sp.slot_off, plan->d_contrib, plan->d_scratch);
        if (sp.nlazy > 0)
            hipLaunchKernelGGL(nd_lazy_apply_kernel, dim3(sp.nlazy), dim3(kThreads), 0, stream, S, n, plan->d_lazy + sp.lazy_off,
                               plan->d_scratch, bad);
        for (int t = 0; t < sp.nsteps; ++t)
            if (sp.max_tiles[t] > 0)
            {
                // fewer than 8 fronts: one XCD each (its 32 compute units take a level's ~60 tiles in two rounds, still inside
                // the time of the tile that factors the next diagonal block)
                const bool pin = xcd_pin_enabled() && (sp.nfronts % 8 == 0 || sp.nfronts < 8);
                const int slots = sp.nfronts < 8 ? 8 : sp.nfronts;
                const dim3 grid = pin ? dim3((unsigned)(sp.max_tiles[t] * slots)) : dim3(sp.max_tiles[t], sp.nfronts);
                hipLaunchKernelGGL(nd_step_kernel, grid, dim3(kThreads), 0, stream, S, n,
                                   plan->d_descs + sp.desc_off + (size_t)t * sp.nfronts, bad, sp.nfronts, sp.max_tiles[t], pin ? 1 : 0);
            }
    }
    for (const StagePlan &sp : plan->stages) {
        if (sp.nvec > 0)
            hipLaunchKernelGGL(nd_fwd_lazy_kernel, dim3(sp.nvec), dim3(kVecThreads), 0, stream, S, n, x, plan->d_vecs + sp.vec_off,
                               plan->d_cols);
        const size_t lds = (size_t)(NB * kLd + kMaxEagerPre * NB * kLd + sp.max_len * NB) * 8;
        hipLaunchKernelGGL(nd_fwd_front_kernel, dim3(sp.nfronts), dim3(kVecThreads), lds, stream, S, n, x,
                           plan->d_descs + sp.desc_off, sp.nfronts, sp.nsteps);
    }
    for (auto it = plan->stages.rbegin(); it != plan->stages.rend(); ++it) {
        const StagePlan &sp = *it;
        const size_t lds = (size_t)(2 * NB * kLd + 64 + sp.max_len * NB) * 8;
        hipLaunchKernelGGL(nd_bwd_front_kernel, dim3(sp.nfronts), dim3(kVecThreads), lds, stream, S, n, x,
                           plan->d_descs + sp.desc_off, sp.nfronts, sp.nsteps);
    }
    MQS_HIP_CHECK(hipGetLastError());
    *done = true;
    return MQS_OK;
}

// How many chunked-solve plans (with their device memory) the process holds: bounded by the LRU above.
extern "C" int mqs_sba_solve_plan_cache_size(void) { return mqs::chol::plan_cache_size(); }

// The plan as plain integers, for inspection and for the CPU test that replays it with numpy (no GPU involved):
// header[8] = {stages, descs, lazy tiles, contributions, lazy vectors, columns, parts, 0}, then per stage 8 ints
// {fronts, steps, desc offset, lazy offset, lazy tiles, vector offset, vectors, longest front}, then the descriptors
// (32 ints each), lazy tiles (8: x1, x2, first contribution, count, factor flag, first partial tile, partial tiles, 0), contributions, lazy vectors (4), columns.  Returns the number of ints needed (0: the
// cut does not apply to this shape); writes them when `cap` suffices.
extern "C" int64_t mqs_sba_solve_plan_dump(int64_t n6, int64_t half_bandwidth, int parts, int32_t *out, int64_t cap)
{
    using namespace mqs::chol;
    if (n6 <= 0 || n6 > 46000 * 6 || half_bandwidth < 0) return 0;
    Plan plan;
    if (!build_plan_host(plan, (int)n6, (int)(half_bandwidth < n6 ? half_bandwidth : n6), parts)) return 0;
    const int64_t need = 8 + 8 * (int64_t)plan.stages.size() + 32 * (int64_t)plan.descs.size() + 8 * (int64_t)plan.lazy.size() +
                         (int64_t)plan.contrib.size() + 4 * (int64_t)plan.vecs.size() + (int64_t)plan.cols.size();
    if (!out || cap < need) return need;
    int32_t *p = out;
    const int32_t header[8] = {(int32_t)plan.stages.size(), (int32_t)plan.descs.size(), (int32_t)plan.lazy.size(),
                               (int32_t)plan.contrib.size(), (int32_t)plan.vecs.size(), (int32_t)plan.cols.size(), plan.parts, 0};
    memcpy(p, header, sizeof(header)); p += 8;
    for (const StagePlan &sp : plan.stages) {
        const int32_t row[8] = {sp.nfronts, sp.nsteps, (int32_t)sp.desc_off, (int32_t)sp.lazy_off, sp.nlazy, (int32_t)sp.vec_off,
                                sp.nvec, sp.max_len};
        memcpy(p, row, sizeof(row)); p += 8;
    }
    memcpy(p, plan.descs.data(), plan.descs.size() * sizeof(StepDesc)); p += 32 * plan.descs.size();
    memcpy(p, plan.lazy.data(), plan.lazy.size() * sizeof(LazyTile)); p += 8 * plan.lazy.size();
    if (!plan.contrib.empty()) memcpy(p, plan.contrib.data(), plan.contrib.size() * 4);
    p += plan.contrib.size();
    if (!plan.vecs.empty()) memcpy(p, plan.vecs.data(), plan.vecs.size() * sizeof(LazyVec));
    p += 4 * plan.vecs.size();
    if (!plan.cols.empty()) memcpy(p, plan.cols.data(), plan.cols.size() * 4);
    return need;
}
