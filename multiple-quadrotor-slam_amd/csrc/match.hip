// Brute-force descriptor matching for gfx950 (MI355X): for every query row the two nearest
// train rows under L2, ties towards the lower train index.
//
// Replaces cv2.batchDistance(..., K = Nt) + the per-query Python loop of
// Work/python_libs/cv2_helpers.py:296-339 (BFMatcher.radiusMatch, k = 2) and the native
// cv2.BFMatcher().knnMatch(k = 2) (reference paths).  The caller-side radius cut, Lowe ratio test
// and trainIdx de-duplication (Work/SLAM/application/own/slam.py:108-125) stay on the host.
//
// Two paths:
//  * knn2_f32  -- exact float32 arithmetic for any D (the reference matches D = 2 pixel
//    coordinates, Nq, Nt ~ 1e2..1e3): one thread per query, train rows staged in LDS, squared
//    distance accumulated in dimension order WITHOUT fused multiply-add, sqrtf at the end --
//    bit-identical to the oracle (oracle/matching_np.py) and to the scalar loop of OpenCV 2.4's
//    normL2Sqr for D < 4.
//  * knn2_f16  -- the 64k x 64k x 256 benchmark shape for binary descriptors expanded to {0,1}
//    half floats: |q - t|^2 = |q|^2 + |t|^2 - 2 q.t with q.t on v_mfma_f32_32x32x16_f16.  Every
//    partial sum is a small integer, so fp16 inputs / fp32 accumulation are EXACT.
//      - operands are swapped (D = T * Q^T): the MFMA's row index is the train row, its column
//        (= lane & 31) the query, so each lane owns ONE query and sees 16 train rows per tile in
//        its 16 accumulator registers: the running top-2 is a register-local scan, no cross-lane
//        traffic until one final exchange between lanes l and l + 32;
//      - a wave keeps its 64 queries (2 column tiles x K = D) resident as B fragments in
//        registers for the whole kernel; train rows stream through a double-buffered, padded
//        (bank-conflict-free ds_read_b128) LDS tile shared by the workgroup's 4 waves;
//      - distance and index are packed into one u32 key (d2 << 20 | index), so the top-2 update
//        is v_min_u32 + v_med3_u32 and the lower index wins ties by construction.
#include "mqs_common.h"
#include <hip/hip_fp16.h>
#include <math.h>

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------
// exact float32 path
// ---------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// A query is shared by kLanesPerQuery lanes, each scanning every kLanesPerQuery-th train row of the LDS tile; the lanes'
// top-2 are then merged in the order the reference's scan implies -- (distance, index) lexicographic: the smaller distance
// first, the LOWER index among equal distances (batchDistance inserts on strict `<` while walking the rows in order).
// One thread per query (the first version) left a 1000 x 1000 problem -- the reference's actual size -- on four
// workgroups walking 1000 rows each: 211 us.  DS > 0: compile-time descriptor length (2 = pixel coordinates).
constexpr int kLanesPerQuery = 16;

__device__ __forceinline__ bool lex_less(float d1, int32_t i1, float d2, int32_t i2)
{
    return (d1 < d2) || (d1 == d2 && i1 < i2);
}

template <int DS>
__global__ __launch_bounds__(kBlock) void knn2_f32_kernel(const float *__restrict__ query, int64_t Nq,
                                                          const float *__restrict__ train, int64_t Nt, int D,
                                                          int tile_rows, int32_t *__restrict__ idx,
                                                          float *__restrict__ dist)
{
    extern __shared__ float sT[];                     // [tile_rows][D]
    constexpr int kQPB = kBlock / kLanesPerQuery;     // queries per workgroup
    constexpr int kEmpty = 0x7fffffff;
    const int tid = threadIdx.x;
    const int g = tid / kLanesPerQuery, l = tid % kLanesPerQuery;
    const int64_t q = (int64_t)blockIdx.x * kQPB + g;
    const bool live = q < Nq;
    const int Dn = DS > 0 ? DS : D;
    const float *qrow = query + (live ? q : 0) * Dn;
    float qv[DS > 0 ? DS : 1];
    if (DS > 0) {
#pragma unroll
        for (int k = 0; k < DS; ++k) qv[k] = qrow[k];
    }
    float b0 = INFINITY, b1 = INFINITY;
    int32_t i0 = kEmpty, i1 = kEmpty;
    for (int64_t t0 = 0; t0 < Nt; t0 += tile_rows) {
        const int rows = (Nt - t0) < tile_rows ? (int)(Nt - t0) : tile_rows;
        __syncthreads();
        for (int e = tid; e < rows * Dn; e += kBlock) sT[e] = train[t0 * Dn + e];
        __syncthreads();
        if (live) {
            for (int r = l; r < rows; r += kLanesPerQuery) {
                const float *trow = sT + r * Dn;
                float s = 0.0f;
                if (DS > 0) {
#pragma unroll
                    for (int k = 0; k < DS; ++k) {
                        const float d = qv[k] - trow[k];
                        s = s + d * d;                 // contract(off): separate multiply and add
                    }
                } else {
                    for (int k = 0; k < Dn; ++k) {
                        const float d = qrow[k] - trow[k];
                        s = s + d * d;
                    }
                }
                const int32_t j = (int32_t)(t0 + r);   // increasing within a lane: strict `<` keeps the lower index
                if (s < b0) { b1 = b0; i1 = i0; b0 = s; i0 = j; }
                else if (s < b1) { b1 = s; i1 = j; }
            }
        }
    }
    // merge the kLanesPerQuery sorted pairs of a query (xor butterfly inside its 16-lane group)
#pragma unroll
    for (int m = kLanesPerQuery / 2; m >= 1; m >>= 1) {
        const float c0 = __shfl_xor(b0, m, 64), c1 = __shfl_xor(b1, m, 64);
        const int32_t j0 = __shfl_xor(i0, m, 64), j1 = __shfl_xor(i1, m, 64);
        const bool mine_first = lex_less(b0, i0, c0, j0) || (b0 == c0 && i0 == j0);
        // first = the smaller head; second = the smaller of the loser's head and the winner's second
        const float f0 = mine_first ? b0 : c0;
        const int32_t g0 = mine_first ? i0 : j0;
        const float x = mine_first ? b1 : c1, y = mine_first ? c0 : b0;
        const int32_t xi = mine_first ? i1 : j1, yi = mine_first ? j0 : i0;
        const bool x_first = lex_less(x, xi, y, yi);
        b0 = f0; i0 = g0;
        b1 = x_first ? x : y; i1 = x_first ? xi : yi;
    }
    if (live && l == 0) {
        const bool h0 = i0 != kEmpty, h1 = i1 != kEmpty;
        idx[2 * q] = h0 ? i0 : -1; idx[2 * q + 1] = h1 ? i1 : -1;
        dist[2 * q] = h0 ? sqrtf(b0) : INFINITY;
        dist[2 * q + 1] = h1 ? sqrtf(b1) : INFINITY;
    }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------
// MFMA path
// ---------------------------------------------------------------------------------------
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using float16v = __attribute__((ext_vector_type(16))) float;

// D = 256 geometry (A/B-tuned on MI355X): query tiles per wave, waves per workgroup
#ifndef MQS_MATCH_QT256
#define MQS_MATCH_QT256 2
#endif
#ifndef MQS_MATCH_SCHED
#define MQS_MATCH_SCHED 1              // the issue order of a step given to the scheduler as sched_group_barriers (round 6: 1.273 -> 1.264 ms on the fp16 path)
#endif
#ifndef MQS_MATCH_NOPEEL
#define MQS_MATCH_NOPEEL 1
#endif
#ifndef MQS_MATCH_NW512
#define MQS_MATCH_NW512 8
#endif
#ifndef MQS_MATCH_PF
#define MQS_MATCH_PF 4
#endif
#ifndef MQS_MATCH_PRUNE_F16
#define MQS_MATCH_PRUNE_F16 1          // the same on the fp16 path.  (Rounds 2-5: off -- with the stage fill's ~100 instructions and a load + wait at every stage's start the
#endif                                 // scan was not what the waves were short of; with those gone, round 6, it is: 1.265 -> 1.192 ms, profiles/r06)
#ifndef MQS_MATCH_F4_QT
#define MQS_MATCH_F4_QT 4
#endif
#ifndef MQS_MATCH_F4_NW
#define MQS_MATCH_F4_NW 8
#endif
#ifndef MQS_MATCH_F4_REJECT16
#define MQS_MATCH_F4_REJECT16 0        // A/B: early reject per accumulator (16 values) instead of per four values (group_step)
#endif
#ifndef MQS_MATCH_F4_REJECT8
#define MQS_MATCH_F4_REJECT8 1         // early reject per eight values in the grouped step (FP4 path; round 6: 0.4326 -> 0.428 ms; per four: 0)
#endif
#ifndef MQS_MATCH_F16_GROUP
#define MQS_MATCH_F16_GROUP 2          // query tiles per train-fragment read on the fp16 path (group_step when > 1).  Round 6: 2 -- a fragment feeds the two query tiles' MFMAs, two
#endif                                 // INDEPENDENT chains per wave (one dependent chain on a lone wave issues an MFMA every 52-56 cycles, tools/probes/mfma_peak.hip) and half
                                       // the LDS reads -- with the early reject on and the fragment prefetch at 2 the four accumulator sets fit (rounds 3-5: 38 spilled registers, 1.76 ms)
#ifndef MQS_MATCH_F16_PF
#define MQS_MATCH_F16_PF 2             // fragment reads in flight ahead of their MFMA in the grouped step of the fp16 path (4: 64 bytes of scratch, 1.32 ms; 2: none, 1.095)
#endif
#ifndef MQS_MATCH_F4_GROUP
#define MQS_MATCH_F4_GROUP 2
#endif
#ifndef MQS_MATCH_PRUNE_F4
#define MQS_MATCH_PRUNE_F4 1
#endif
#ifndef MQS_MATCH_NW256
#define MQS_MATCH_NW256 8
#endif
#ifndef MQS_MATCH_STAGE_ROWS
#define MQS_MATCH_STAGE_ROWS 128
#endif
#ifndef MQS_MATCH_MIN_BLOCKS
#define MQS_MATCH_MIN_BLOCKS 1           // A/B: workgroups per CU the register budget is cut for (2 with MQS_MATCH_NW256=4, MQS_MATCH_STAGE_ROWS=64)
#endif
#ifndef MQS_MATCH_TN_AHEAD
#define MQS_MATCH_TN_AHEAD 1           // A/B: a stage's row start values (|t|^2 + bias + tile) are LOADED a stage before they are written to LDS (0: round 5's load + wait + write at the stage's start)
#endif
#ifndef MQS_MATCH_BUFFER_DMA
#define MQS_MATCH_BUFFER_DMA 1         // A/B: the stage fill by buffer_load ... lds (descriptor + per-lane offsets computed once + one scalar per stage) instead of global_load_lds with per-chunk 64-bit address arithmetic
#endif
#if MQS_MATCH_BUFFER_DMA && defined(__HIP_DEVICE_COMPILE__)
#define MQS_MATCH_BUFFER_DMA_DEV 1     // (the buffer builtins exist in the device pass only; a kernel body that does not compile in the host pass loses its launch stub)
#else
#define MQS_MATCH_BUFFER_DMA_DEV 0
#endif
#ifndef MQS_MATCH_SPREAD_DMA
#define MQS_MATCH_SPREAD_DMA 1         // A/B: the next stage's fill issued piece by piece between this stage's (tile, query tile) steps instead of all at the stage's start
#endif
#ifndef MQS_MATCH_STAGGER
#define MQS_MATCH_STAGGER 0            // A/B: s_sleep argument (x 64 cycles) by which the second wave of every SIMD trails the first behind each stage barrier
#endif
// Train rows per LDS stage (a whole number of 32-row MFMA tiles): the largest of MQS_MATCH_STAGE_ROWS, its half, .. 64 of
// which two stages fit the LDS.  Every stage ends in a workgroup barrier behind which all eight waves start again with the
// latency of their first fragment reads exposed; at 64 rows (64 MFMAs per wave and stage) that was 8 % of the fp16 kernel's
// time (1.41 -> 1.30 ms at 128 rows, same box) -- a stage should be as long as the LDS allows.
constexpr int stage_bytes(int rows, int ks) { return rows * (2 * ks + 1) * 16; }
constexpr int stage_rows(int ks, int rows_max = MQS_MATCH_STAGE_ROWS)
{
    int rows = rows_max;
    while (rows > 64 && 2 * stage_bytes(rows, ks) > 150 * 1024) rows /= 2;
    return rows;
}
// The running comparison is on ONE float per (query, train row),
//     d' = kBias + |t|^2 - 2 q.t + tile / 256            (tile = (row >> 5) & 255)
// produced by the MFMAs themselves: the queries are held pre-scaled by -2 and the accumulator of a
// tile starts from kBias + |t|^2 + tile / 256 instead of zero.  |q|^2 is constant per lane (one query per
// lane) and added back at the end.  For {0,1} data with D <= 512, d' lies in [512, 1537): as an IEEE float
// its integer part needs <= 11 bits, so the 8 fraction bits 2^-1..2^-8 carry the tile and the 5 lowest
// mantissa bits (<= 31 * 2^-13 < 2^-8) are free for the row inside the tile, OR-ed in.  The bit pattern of
// a positive float orders like the float, so (distance, row index) is ONE unsigned key and a tile value
// costs 3 VALU instructions (v_or3, v_med3_u32, v_min_u32).  Every kWindowTiles tiles (8192 rows) the
// window's best two are decoded and merged into the running result (strict <: the lower row wins ties).
constexpr float kBias = 1024.0f;
constexpr float kPadNorm = 8192.0f;                    // padding rows: d' >= 8192 - 2 * 512
constexpr float kInvalid = 2048.0f;                    // d' >= this: padding

// squared norms (exact for {0,1} data): one thread per row
__global__ void row_sqnorm_kernel(const _Float16 *__restrict__ x, int64_t n, int D, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const half8 *r = reinterpret_cast<const half8 *>(x + i * D);
    float s = 0.0f;
    for (int k = 0; k < D / 8; ++k) {
        const half8 v = r[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j] * (float)v[j];
    }
    out[i] = s;
}

using int4v = __attribute__((ext_vector_type(4))) int;

// What differs between the two element types of the MFMA path.  A fragment is 16 bytes per lane either way
// (8 halves: k = 16 ks + 8 h + j;  32 FP4 nibbles: k = 64 ks + 32 h + j), so the LDS image and its addressing are shared.
struct F16Path {
    using elem = _Float16;
    using frag = half8;
    using accv = float16v;
    using start_t = float;
    using start4 = __attribute__((ext_vector_type(4))) float;
    static constexpr int kPerMfma = 16;                 // contraction depth of one MFMA
    static constexpr int kWindowTiles = 256;            // 8 fraction bits carry the tile
    static __device__ __forceinline__ frag prep_query(frag v) { return v * (_Float16)(-2.0f); }
    static __device__ __forceinline__ accv mfma(frag a, frag b, accv c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    // d' crosses a binade (512 .. 1537), so the row bits are OR-ed in after the sum.  (Keeping the sum in one binade -- bias
    // 2560, windows of 128 tiles, tile and row both in the start value -- removes the v_or3 but measured the same 1.342 ms on the
    // same box: at 3 vector instructions per MFMA the fp16 kernel is not bound by vector issue.)
    static constexpr bool kPrune = MQS_MATCH_PRUNE_F16 != 0;
    static constexpr int kGroup = MQS_MATCH_F16_GROUP;
    static constexpr int kGroupPF = MQS_MATCH_F16_PF;      // fragment prefetch depth of the grouped step
    static constexpr int kStageRowsMax = MQS_MATCH_STAGE_ROWS;
    // smallest key a value at distance part >= d can have (positive floats order like their bit patterns)
    static __device__ __forceinline__ unsigned key_floor(float d) { return __float_as_uint(d); }
    static constexpr float kBiasV = kBias;
    static __device__ __forceinline__ start_t start(float tnorm, int64_t t) { return tnorm + kBias + (float)((t >> 5) & (kWindowTiles - 1)) * (1.0f / kWindowTiles); }
    static __device__ __forceinline__ start_t pad() { return kPadNorm; }
    static __device__ __forceinline__ unsigned key(float v) { return __float_as_uint(v); }
    // key -> (distance part kBias + |t|^2 - 2 q.t, tile in window, row in tile); false: padding
    static __device__ __forceinline__ bool decode(unsigned k, float &d, int &tile, int &row)
    {
        const float f = __uint_as_float(k);
        if (!(f < kInvalid)) return false;
        d = floorf(f);
        tile = (int)((f - d) * (float)kWindowTiles);
        row = (int)(k & 31u);
        return true;
    }
};

// {0,1} descriptors as FP4 (E2M1) on the block-scaled matrix instruction without scales, v_mfma_f32_32x32x64_f8f6f4 cbsz:4
// blgp:4: 64 contraction steps per instruction at the cycles of the 16-step fp16 / 32-step int8 forms (MI355X: ~10 PF dense),
// half the LDS bytes of the int8 image.  Train nibbles are 0 / 1.0 (0x2), query nibbles 0 / -2.0 (0xC): every product and
// every partial sum is a small integer, exact in the fp32 accumulator, which starts from kBias + |t|^2 + tile / 256 as on the
// fp16 path -- the key, its decoding and the windows are F16Path's.  A fragment is 32 nibbles = 16 bytes per lane; which
// contraction index a nibble position stands for does not matter as long as query and train rows are packed alike (they are:
// the same expansion kernel), and the parity tests against the oracle would show a mismatch.
using int8v = __attribute__((ext_vector_type(8))) int;
struct F4Path : F16Path {
    using elem = signed char;                           // two nibbles per byte: 32 bytes of a row per MFMA
    using frag = int4v;
    static constexpr int kPerMfma = 32;
    static constexpr bool kPrune = MQS_MATCH_PRUNE_F4 != 0;
    static constexpr int kGroup = MQS_MATCH_F4_GROUP;      // query tiles per train-fragment read (see group_step)
    static constexpr int kGroupPF = MQS_MATCH_PF;
    static constexpr int kStageRowsMax = 256;              // 36 KB stages (A/B: 128 rows + 3 %, 512 rows + 5 % time)
    static __device__ __forceinline__ frag prep_query(frag v) { return v; }
    static __device__ __forceinline__ accv mfma(frag a, frag b, accv c)
    {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(int8v{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                               int8v{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, c, 4, 4, 0, 0, 0, 0);
    }
};

template <class TP /* F16Path or F4Path */, int KS /* MFMAs per (train tile, query tile) = D / TP::kPerMfma */,
          int QT /* 32-query column tiles per wave */, int NW /* waves per workgroup */>
__global__ __launch_bounds__(NW * 64, MQS_MATCH_MIN_BLOCKS) void knn2_mfma_kernel(const typename TP::elem *__restrict__ query, int64_t Nq,
                                                           const typename TP::elem *__restrict__ train, int64_t Nt,
                                                          const float *__restrict__ qnorm,
                                                          const float *__restrict__ tnorm,
                                                          int32_t *__restrict__ idx, float *__restrict__ dist,
                                                          float *__restrict__ part_d, int32_t *__restrict__ part_i)
{
    using frag_t = typename TP::frag;
    using accv_t = typename TP::accv;
    using start_t = typename TP::start_t;
    constexpr int D = KS * TP::kPerMfma;
    constexpr int kWindowTiles = TP::kWindowTiles;
    constexpr int kStageRows = stage_rows(KS, TP::kStageRowsMax);
    constexpr int kStageTiles = kStageRows / 32;
    constexpr int kWindowStages = kWindowTiles * 32 / kStageRows;
    constexpr int kVecPerRow = 2 * KS;                 // 16-byte pieces of data per row (one per (k-step, lane half))
    constexpr int kPiecesPerRow = kVecPerRow + 1;      // + one piece of padding: row stride = 4 banks mod 64
    constexpr int kRowBytes = kPiecesPerRow * 16;
    constexpr int kStageBytes = kStageRows * kRowBytes;        // a whole number of 1-KiB LDS-DMA chunks (64 rows)
    constexpr int kChunks = kStageBytes / 1024;
    constexpr int kChunksPerWave = (kChunks + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * kStageBytes];
    __shared__ __attribute__((aligned(16))) start_t sTn[2 * kStageRows];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t qbase = (int64_t)blockIdx.x * (NW * QT * 32) + wave * (QT * 32);

    // resident query fragments, scaled by -2 (exact): B[k = 16 ks + 8 h + j][col r] = -2 Q[qbase + 32 qt + r][...]
    frag_t qf[QT][KS];
    float qn[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t q = qbase + 32 * qt + r;
        const bool ok = q < Nq;
        const frag_t *row = reinterpret_cast<const frag_t *>(query + (ok ? q : 0) * D);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t v = row[2 * ks + h];
            if (!ok) v = frag_t{};
            qf[qt][ks] = TP::prep_query(v);
        }
        qn[qt] = ok ? qnorm[q] : 0.0f;
    }
    unsigned best[QT], second[QT];           // current window: keys
    // Early reject (TP::kPrune).  A value can only end among a query's best two if its key is below BOTH the window's
    // current second key and every key at the distance of the running second-best over the finished windows (later rows
    // lose ties).  thr = the smaller of the two; four values are reduced with v_min3 + v_min, compared with thr, and the
    // wave skips their scan when no lane has a candidate -- after the first few hundred rows that is the usual case, and
    // the per-value cost falls from 3 vector instructions to 3/4.
    unsigned thr[QT];
    float gd0[QT], gd1[QT];                  // running result over the finished windows: kBias + |t|^2 - 2 q.t
    int gi0[QT], gi1[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        best[qt] = 0xFFFFFFFFu; second[qt] = 0xFFFFFFFFu;
        thr[qt] = 0xFFFFFFFFu;
        gd0[qt] = INFINITY; gd1[qt] = INFINITY; gi0[qt] = -1; gi1[qt] = -1;
    }

    // gridDim.y > 1: the train rows are split over blockIdx.y in whole windows (a small query block cannot
    // amortise the train stream it pulls from L2; two query tiles per wave can, but then Nq / 512 workgroups
    // alone would leave CUs idle); each part writes its best two to part_d / part_i, merged by merge_parts_kernel
    const int64_t nstages_all = (Nt + kStageRows - 1) / kStageRows;
    const int64_t nwin = (nstages_all + kWindowStages - 1) / kWindowStages;
    const int64_t win_per_part = (nwin + gridDim.y - 1) / gridDim.y;
    const int64_t s_begin = blockIdx.y * win_per_part * kWindowStages;
    const int64_t s_end_ = s_begin + win_per_part * kWindowStages;
    const int64_t s_end = s_end_ < nstages_all ? s_end_ : nstages_all;
    const int64_t nstages = s_end > s_begin ? s_end - s_begin : 0;
    const uint4 *tvec = reinterpret_cast<const uint4 *>(train);
    (void)tvec;

    constexpr int G = (TP::kGroup > 1 && QT % TP::kGroup == 0) ? TP::kGroup : 1;     // query tiles per fragment read (group_step)

    // Stage fill by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write).  One
    // wave-instruction writes 64 x 16 B = 1 KiB of LDS linearly (wave-uniform base + lane * 16); which
    // global bytes land there is the per-lane SOURCE address, so the LDS image is laid out with padded
    // rows (kPiecesPerRow pieces: the fragment reads of 16 consecutive rows then fall in 16 different
    // bank groups, and their addresses are row base + an immediate: no address arithmetic in the loop);
    // the padding piece of a row re-reads its first piece.  Rows past Nt re-read the last row (their start
    // value keeps them out of the result).
    // Round 6: what a stage's fill costs the waves that issue it.  Round 5 issued all of a wave's pieces at the stage's start (per
    // piece a 64-bit row address, a clamp to the last row, the LDS address through v_readfirstlane: ~11 instructions, ~100 per
    // stage, by both waves of every SIMD at once -- the matrix pipe idles meanwhile), and the waves that also write the rows' start
    // values loaded |t|^2 and WAITED for it there (s_waitcnt vmcnt(0): behind every piece they had just issued -- a DMA round
    // trip per stage, and the other waves wait for them at the stage's barrier).  Now: a buffer descriptor over the train rows,
    // the pieces' per-lane byte offsets computed once, one scalar offset per stage (rows beyond the end read as zeros: their start
    // value keeps them out of the result); the pieces issued one by one between the stage's steps; |t|^2 loaded a stage AHEAD of
    // its write (it has landed by the barrier in between).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // (the wave index as a scalar: LDS-DMA destinations are wave-uniform)
    constexpr int64_t kRowBytesG = (int64_t)kVecPerRow * 16;          // bytes of a row in global memory
    (void)kRowBytesG;
#if MQS_MATCH_BUFFER_DMA_DEV
    // the descriptor starts at this part's first stage (a part's rows are < 4 GiB); num_records = the bytes up to the last row
    const int64_t part_first_row = s_begin * kStageRows;
    const int64_t part_bytes_ = (Nt - part_first_row) * kRowBytesG;
    const unsigned part_bytes = part_bytes_ <= 0 ? 0u : (part_bytes_ > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)part_bytes_);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const unsigned char *>(train) + part_first_row * kRowBytesG),
                                                        (short)0, (int)part_bytes, 0x00020000);
    int voff[kChunksPerWave];
#pragma unroll
    for (int i = 0; i < kChunksPerWave; ++i) {
        const int v = (i * NW + wave) * 64 + lane;
        const int row = v / kPiecesPerRow, colp = v % kPiecesPerRow;
        voff[i] = (row * kVecPerRow + (colp < kVecPerRow ? colp : 0)) * 16;
    }
#endif
    // piece i of this wave of stage s's fill
    auto stage_piece = [&](int64_t s, int i) {
        const int buf = (int)(s & 1);
        const int chunk = i * NW + wave_u;
        if (chunk < kChunks) {
#if MQS_MATCH_BUFFER_DMA_DEV
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(sTile + buf * kStageBytes + chunk * 1024), 16, voff[i],
                                                     (int)((s - s_begin) * (kStageRows * kRowBytesG)), 0, 0);
#else
            const int v = chunk * 64 + lane;
            const int row = v / kPiecesPerRow, colp = v % kPiecesPerRow;
            const int col = colp < kVecPerRow ? colp : 0;
            int64_t grow = s * kStageRows + row;
            grow = grow < Nt ? grow : Nt - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(tvec + grow * kVecPerRow + col),
                (__attribute__((address_space(3))) void *)(sTile + buf * kStageBytes + chunk * 1024), 16, 0, 0);
#endif
        }
    };
    // the rows' start values of stage s: kBias + |t|^2 + tile / 256 (padding rows: out of the result)
    float tn_ahead = 0.0f;                                            // |t|^2 of this thread's row of the stage after next (tid < kStageRows)
    auto tn_fetch = [&](int64_t s) {
        const int64_t t = s * kStageRows + tid;
        if (tid < kStageRows && s < s_end && t < Nt) tn_ahead = tnorm[t];
    };
    auto tn_write = [&](int64_t s, float tn) {
        if (tid < kStageRows) {
            const int64_t t = s * kStageRows + tid;
            sTn[(int)(s & 1) * kStageRows + tid] = (t < Nt) ? TP::start(tn, t) : TP::pad();
        }
    };
    auto stage_issue = [&](int64_t s) {                               // a whole stage at once (the first one; MQS_MATCH_SPREAD_DMA = 0)
#pragma unroll
        for (int i = 0; i < kChunksPerWave; ++i) stage_piece(s, i);
    };

    // One step = the KS MFMAs of (train tile, query tile qt) into `acc` (started from the rows' start values
    // `tn`), interleaved with the top-2 scan of the PREVIOUS step's accumulators `prev` (query tile pq): the
    // scan's 3 VALU instructions per value issue in the shadow of the MFMAs instead of after them.  Steps
    // run (tile 0, qt 0), (tile 0, qt 1), .., (tile 1, qt 0), ..: with QT >= 2 the accumulators simply
    // rotate over the query tiles (no second set), and every train stage is amortised over QT * 32 queries
    // per wave -- the LDS-DMA stream from L2, not the matrix pipe, is what a small query block runs out of.
    auto tile_step = [&](const unsigned char *tile, int tt, int qt, accv_t &acc, const accv_t &prev, int pq, const start_t *tn) {
        const unsigned char *arow = tile + (tt * 32 + r) * kRowBytes + 16 * h;   // fragment ks: + 32 ks (immediate)
        accv_t start;                                         // row(e) = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const typename TP::start4 t4 = *reinterpret_cast<const typename TP::start4 *>(tn + 8 * g);
#pragma unroll
            for (int k = 0; k < 4; ++k) start[4 * g + k] = t4[k];
        }
        constexpr int PF = MQS_MATCH_PF < KS ? MQS_MATCH_PF : KS;     // fragment reads in flight ahead of their MFMA
        frag_t a[KS];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) a[ks] = *reinterpret_cast<const frag_t *>(arow + 32 * ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) a[ks + PF] = *reinterpret_cast<const frag_t *>(arow + 32 * (ks + PF));
            acc = TP::mfma(a[ks], qf[qt][ks], ks == 0 ? start : acc);
            if constexpr (TP::kPrune && KS % 4 == 0) {
                // group g = values 4g .. 4g + 3 of the previous step, behind the last MFMA of its quarter of the k-steps
                if (ks % (KS / 4) == KS / 4 - 1) {
                    const int g = ks / (KS / 4);
                    const unsigned k0 = TP::key(prev[4 * g]), k1 = TP::key(prev[4 * g + 1]), k2 = TP::key(prev[4 * g + 2]),
                                   k3 = TP::key(prev[4 * g + 3]);
                    unsigned m3;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(k0), "v"(k1), "v"(k2));
                    const unsigned m4 = min(m3, k3);
                    // without the row bits a key can only be smaller: the test errs on the side of scanning
                    if (__builtin_amdgcn_ballot_w64(m4 < thr[pq]) != 0) {
#pragma unroll
                        for (int e = 4 * g; e < 4 * g + 4; ++e) {
                            const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                            unsigned m;
                            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                            second[pq] = m;
                            best[pq] = min(best[pq], key);
                        }
                        thr[pq] = min(second[pq], TP::key_floor(gd1[pq]));
                    }
                }
            } else {
            // scan values [v0, v1) of the previous step behind this MFMA
            const int v0 = ks * 16 / KS, v1 = (ks + 1) * 16 / KS;
#pragma unroll
            for (int e = v0; e < v1; ++e) {
                const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));   // v_or3_b32
                unsigned m;
                asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                second[pq] = m;
                best[pq] = min(best[pq], key);
            }
            }
        }
        if (QT >= 2) __builtin_amdgcn_sched_barrier(0);       // keep the next step's loads out of this one (VGPRs)
#if MQS_MATCH_SCHED
        // issue order for the scheduler: the start values and the first fragment reads, then per k-step its
        // MFMA, one more fragment read (several steps ahead of its use) and its share of the scan
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + MQS_MATCH_PF, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3 * 16 / KS, 0);
        }
#endif
    };

    // G query tiles per train-fragment read.  A step above re-reads the tile's KS fragments and its 16 start values for every
    // query tile: (KS + 4) / KS LDS instructions per MFMA -- 2.0 on the FP4 path, where an MFMA covers 64 contraction steps, and
    // with four SIMDs sharing the LDS (256 B/clk: 4 cycles per ds_read_b128) that is 32 of every 32 cycles of a saturated matrix
    // pipe.  Here a fragment feeds G MFMAs (independent accumulators) and the start values are read once per tile by the caller:
    // 1 / G LDS instructions per MFMA.  The accumulators form QT / G groups (two sets of one group when G == QT); step j writes
    // group j % NG and scans the previous one.
    auto group_step = [&](const unsigned char *tile, int tt, const accv_t &start, accv_t *acc, int cur0, int prev0, int curq0, int prevq0) {
        if constexpr (G > 1) {
        const unsigned char *arow = tile + (tt * 32 + r) * kRowBytes + 16 * h;
        constexpr int PF = TP::kGroupPF < KS ? TP::kGroupPF : KS;
        frag_t a[KS];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) a[ks] = *reinterpret_cast<const frag_t *>(arow + 32 * ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) a[ks + PF] = *reinterpret_cast<const frag_t *>(arow + 32 * (ks + PF));
#pragma unroll
            for (int u = 0; u < G; ++u) acc[cur0 + u] = TP::mfma(a[ks], qf[curq0 + u][ks], ks == 0 ? start : acc[cur0 + u]);
            // the previous group's 16 G values, 16 G / KS behind each k-step
            constexpr int kPer = 16 * G / KS;
            static_assert(16 * G % KS == 0 && (!TP::kPrune || kPer % 4 == 0 || 4 % kPer == 0), "scan shares");
            if constexpr (TP::kPrune && MQS_MATCH_F4_REJECT16 && KS % G == 0) {
                // one early-reject test per ACCUMULATOR of the previous group (16 values: a tree of five v_min3, two v_min3, one
                // v_min -- 8 instructions + the compare, where four tests of four values cost 16), behind the k-step at which its
                // share of the scan would start; only a wave that holds a candidate among the 16 falls back to the groups of four
                if (ks % (KS / G) == 0) {
                    const int which = ks / (KS / G), pq = prevq0 + which;
                    const accv_t &prev = acc[prev0 + which];
                    unsigned k[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) k[e] = TP::key(prev[e]);
                    unsigned a0, a1, a2, a3, a4, b0, b1;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a0) : "v"(k[0]), "v"(k[1]), "v"(k[2]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a1) : "v"(k[3]), "v"(k[4]), "v"(k[5]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a2) : "v"(k[6]), "v"(k[7]), "v"(k[8]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a3) : "v"(k[9]), "v"(k[10]), "v"(k[11]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a4) : "v"(k[12]), "v"(k[13]), "v"(k[14]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(b0) : "v"(a0), "v"(a1), "v"(a2));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(b1) : "v"(a3), "v"(a4), "v"(k[15]));
                    const unsigned m16 = min(b0, b1);
                    if (__builtin_amdgcn_ballot_w64(m16 < thr[pq]) != 0) {
#pragma unroll
                        for (int e0 = 0; e0 < 16; e0 += 4) {
                            unsigned m3;
                            asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(k[e0]), "v"(k[e0 + 1]), "v"(k[e0 + 2]));
                            const unsigned m4 = min(m3, k[e0 + 3]);
                            if (__builtin_amdgcn_ballot_w64(m4 < thr[pq]) != 0) {
#pragma unroll
                                for (int e = e0; e < e0 + 4; ++e) {
                                    const unsigned key = k[e] | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                                    unsigned m;
                                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                                    second[pq] = m;
                                    best[pq] = min(best[pq], key);
                                }
                                thr[pq] = min(second[pq], TP::key_floor(gd1[pq]));
                            }
                        }
                    }
                }
            } else if constexpr (TP::kPrune && MQS_MATCH_F4_REJECT8 && kPer % 8 == 0) {
                // one early-reject test per EIGHT values (three v_min3 + one v_min + the compare: 5 instructions where two tests of four
                // take 8), spread over the k-steps like the tests of four; a wave that holds a candidate among the eight scans them all
#pragma unroll
                for (int gq = 0; gq < kPer / 8; ++gq) {
                    const int v0 = ks * kPer + 8 * gq, which = v0 >> 4, e0 = v0 & 15, pq = prevq0 + which;
                    const accv_t &prev = acc[prev0 + which];
                    unsigned k[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) k[e] = TP::key(prev[e0 + e]);
                    unsigned a0, a1, a2;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a0) : "v"(k[0]), "v"(k[1]), "v"(k[2]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a1) : "v"(k[3]), "v"(k[4]), "v"(k[5]));
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(a2) : "v"(a0), "v"(a1), "v"(k[6]));
                    const unsigned m8 = min(a2, k[7]);
                    if (__builtin_amdgcn_ballot_w64(m8 < thr[pq]) != 0) {
#pragma unroll
                        for (int e = e0; e < e0 + 8; ++e) {
                            const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                            unsigned m;
                            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                            second[pq] = m;
                            best[pq] = min(best[pq], key);
                        }
                        thr[pq] = min(second[pq], TP::key_floor(gd1[pq]));
                    }
                }
            } else if constexpr (TP::kPrune && kPer < 4) {
                // fewer than four values per k-step (the fp16 path with G = 2: two): one test of four every 4 / kPer steps
                constexpr int kEvery = 4 / kPer;
                if (ks % kEvery == kEvery - 1) {
                    const int v0 = (ks - (kEvery - 1)) * kPer, which = v0 >> 4, e0 = v0 & 15, pq = prevq0 + which;
                    const accv_t &prev = acc[prev0 + which];
                    const unsigned k0 = TP::key(prev[e0]), k1 = TP::key(prev[e0 + 1]), k2 = TP::key(prev[e0 + 2]), k3 = TP::key(prev[e0 + 3]);
                    unsigned m3;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(k0), "v"(k1), "v"(k2));
                    const unsigned m4 = min(m3, k3);
                    if (__builtin_amdgcn_ballot_w64(m4 < thr[pq]) != 0) {
#pragma unroll
                        for (int e = e0; e < e0 + 4; ++e) {
                            const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                            unsigned m;
                            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                            second[pq] = m;
                            best[pq] = min(best[pq], key);
                        }
                        thr[pq] = min(second[pq], TP::key_floor(gd1[pq]));
                    }
                }
            } else if constexpr (TP::kPrune) {
#pragma unroll
                for (int gq = 0; gq < kPer / 4; ++gq) {
                    const int v0 = ks * kPer + 4 * gq, which = v0 >> 4, e0 = v0 & 15, pq = prevq0 + which;
                    const accv_t &prev = acc[prev0 + which];
                    const unsigned k0 = TP::key(prev[e0]), k1 = TP::key(prev[e0 + 1]), k2 = TP::key(prev[e0 + 2]), k3 = TP::key(prev[e0 + 3]);
                    unsigned m3;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(k0), "v"(k1), "v"(k2));
                    const unsigned m4 = min(m3, k3);
                    if (__builtin_amdgcn_ballot_w64(m4 < thr[pq]) != 0) {
#pragma unroll
                        for (int e = e0; e < e0 + 4; ++e) {
                            const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                            unsigned m;
                            asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                            second[pq] = m;
                            best[pq] = min(best[pq], key);
                        }
                        thr[pq] = min(second[pq], TP::key_floor(gd1[pq]));
                    }
                }
            } else {
#pragma unroll
                for (int v = ks * kPer; v < (ks + 1) * kPer; ++v) {
                    const int which = v >> 4, e = v & 15, pq = prevq0 + which;
                    const accv_t &prev = acc[prev0 + which];
                    const unsigned key = TP::key(prev[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                    unsigned m;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                    second[pq] = m;
                    best[pq] = min(best[pq], key);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        }
    };

    // decode a window key: distance part, row index
    auto push = [&](int qt, unsigned key, int64_t window_base) {
        float d;
        int tile, row;
        if (!TP::decode(key, d, tile, row)) return;
        const int i = (int)window_base + tile * 32 + row;
        // selects, not branches: the compiler merged the two branch bodies into one store through a computed address, which
        // put the index arrays in scratch memory
        const bool lt0 = d < gd0[qt], lt1 = d < gd1[qt];
        gd1[qt] = lt0 ? gd0[qt] : (lt1 ? d : gd1[qt]);
        gi1[qt] = lt0 ? gi0[qt] : (lt1 ? i : gi1[qt]);
        gd0[qt] = lt0 ? d : gd0[qt];
        gi0[qt] = lt0 ? i : gi0[qt];
    };
    auto close_window = [&](int64_t window_base) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            push(qt, best[qt], window_base);
            push(qt, second[qt], window_base);
            best[qt] = 0xFFFFFFFFu; second[qt] = 0xFFFFFFFFu;
            thr[qt] = TP::key_floor(gd1[qt]);
        }
    };

    if (nstages > 0) {                                   // Nt == 0: nothing to read, every key stays invalid
        stage_issue(s_begin);
        tn_fetch(s_begin);
        tn_write(s_begin, tn_ahead);                     // (the one wait for a start value: the first stage's)
        tn_fetch(s_begin + 1);                           // lands by the barrier below
    }
    __syncthreads();                                     // (waits for the LDS-DMA: vmcnt(0) + barrier)

    constexpr int NG = QT / G;                           // accumulator groups; one group alone needs a second set to scan
    constexpr int NSETS = G > 1 ? (NG < 2 ? 2 : NG) : (QT < 2 ? 2 : QT);
    constexpr int R = G > 1 ? NSETS * G : NSETS;         // accumulator ring: step j writes set j % NSETS, scans set (j - 1) % NSETS
    static_assert((kStageTiles * (G > 1 ? NG : QT)) % NSETS == 0, "the ring position must repeat every stage");
    accv_t acc[R];
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int u = 0; u < G; ++u) acc[R - 1 - u][e] = TP::pad();   // "previous step" of the first one: nothing

#if MQS_MATCH_NOPEEL
#pragma clang loop unroll(disable)
#endif
    for (int64_t s = s_begin; s < s_end; ++s) {
        // the next stage: its rows' start values from the |t|^2 loaded a stage ago (complete since the last barrier: no wait), the
        // load for the stage after it, and its fill -- at once, or piece by piece between this stage's steps
        const bool more = s + 1 < s_end;
#if MQS_MATCH_TN_AHEAD
        if (more) { tn_write(s + 1, tn_ahead); tn_fetch(s + 2); }
#else
        if (more) { tn_fetch(s + 1); tn_write(s + 1, tn_ahead); }
#endif
#if !MQS_MATCH_SPREAD_DMA
        if (more) stage_issue(s + 1);                     // lands while this stage is computed
#endif
        constexpr int kSteps = kStageTiles * ((TP::kGroup > 1 && QT % TP::kGroup == 0) ? QT / TP::kGroup : QT);
        auto pieces_before_step = [&](int j) {            // step j of kSteps: the pieces [j * n / kSteps, (j + 1) * n / kSteps) of the next stage's fill
#if MQS_MATCH_SPREAD_DMA
            if (more) {
#pragma unroll
                for (int i = j * kChunksPerWave / kSteps; i < (j + 1) * kChunksPerWave / kSteps; ++i) stage_piece(s + 1, i);
            }
#else
            (void)j;
#endif
        };
        const unsigned char *tile = sTile + (int)(s & 1) * kStageBytes;
        const start_t *tn0 = sTn + (int)(s & 1) * kStageRows + 4 * h;
        if constexpr (G > 1) {
#pragma unroll
            for (int tt = 0; tt < kStageTiles; ++tt) {
                accv_t start;                                 // row(e) = (e & 3) + 8 (e >> 2) + 4 h: once per tile
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const typename TP::start4 t4 = *reinterpret_cast<const typename TP::start4 *>(tn0 + 32 * tt + 8 * g4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) start[4 * g4 + k] = t4[k];
                }
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int j = tt * NG + gi;
                    pieces_before_step(j);
                    group_step(tile, tt, start, acc, (j % NSETS) * G, ((j + NSETS - 1) % NSETS) * G, gi * G, ((gi + NG - 1) % NG) * G);
                    if (j == 0 && (s & (kWindowStages - 1)) == 0 && s > s_begin)
                        close_window((s / kWindowStages - 1) * (int64_t)(kWindowTiles * 32));
                }
            }
        } else {
#pragma unroll
        for (int tt = 0; tt < kStageTiles; ++tt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int j = tt * QT + qt;
                pieces_before_step(j);
                tile_step(tile, tt, qt, acc[j % R], acc[(j + R - 1) % R], (qt + QT - 1) % QT, tn0 + 32 * tt);
                // step 0 scanned the last (tile, query tile) of the previous stage: a window may end there
                if (j == 0 && (s & (kWindowStages - 1)) == 0 && s > s_begin)
                    close_window((s / kWindowStages - 1) * (int64_t)(kWindowTiles * 32));
            }
        }
#if defined(MQS_MATCH_EXPERIMENT_NOBARRIER)      // timing experiment only (races): what the per-stage workgroup barrier costs
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        __syncthreads();
#endif
#if MQS_MATCH_STAGGER > 0
        // waves w and w + NW / 2 share a SIMD and run the same program: behind the barrier both would sit out the latency of their
        // first start-value and fragment reads together, and meet every later step boundary together too
        if (wave >= NW / 2) __builtin_amdgcn_s_sleep(MQS_MATCH_STAGGER);
#endif
    }
    // scan of the last step, last window
    if (nstages > 0) {
        // the last step's accumulators: set (steps per stage - 1) % NSETS, query tiles QT - G ..
        constexpr int jl = kStageTiles * (G > 1 ? NG : QT) - 1;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const accv_t &last = acc[G > 1 ? (jl % NSETS) * G + u : jl % R];
            const int qt = QT - G + u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned key = TP::key(last[e]) | (unsigned)(4 * h) | (unsigned)(8 * (e >> 2) + (e & 3));
                second[qt] = max(best[qt], min(second[qt], key));
                best[qt] = min(best[qt], key);
            }
        }
        close_window(((s_end - 1) / kWindowStages) * (int64_t)(kWindowTiles * 32));
    }

    // merge the two half-waves (same query, disjoint train rows; ties: the lower row), undo the bias, write
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float od0 = __shfl_xor(gd0[qt], 32), od1 = __shfl_xor(gd1[qt], 32);
        const int oi0 = __shfl_xor(gi0[qt], 32), oi1 = __shfl_xor(gi1[qt], 32);
        auto before = [](float da, int ia, float db, int ib) { return da < db || (da == db && (unsigned)ia < (unsigned)ib); };
        float md0, md1; int mi0, mi1;
        if (before(gd0[qt], gi0[qt], od0, oi0)) {
            md0 = gd0[qt]; mi0 = gi0[qt];
            if (before(gd1[qt], gi1[qt], od0, oi0)) { md1 = gd1[qt]; mi1 = gi1[qt]; } else { md1 = od0; mi1 = oi0; }
        } else {
            md0 = od0; mi0 = oi0;
            if (before(od1, oi1, gd0[qt], gi0[qt])) { md1 = od1; mi1 = oi1; } else { md1 = gd0[qt]; mi1 = gi0[qt]; }
        }
        const int64_t q = qbase + 32 * qt + r;
        if (h == 0 && q < Nq) {
            if (gridDim.y == 1) {
                idx[2 * q] = mi0;
                idx[2 * q + 1] = mi1;
                dist[2 * q] = mi0 >= 0 ? sqrtf(md0 - TP::kBiasV + qn[qt]) : INFINITY;
                dist[2 * q + 1] = mi1 >= 0 ? sqrtf(md1 - TP::kBiasV + qn[qt]) : INFINITY;
            } else {
                const int64_t o = ((int64_t)blockIdx.y * Nq + q) * 2;
                part_d[o] = md0 - TP::kBiasV + kBias; part_d[o + 1] = md1 - TP::kBiasV + kBias;      // parts carry the common bias (merge_parts_kernel)
                part_i[o] = mi0; part_i[o + 1] = mi1;
            }
        }
    }
}

// Best two over the parts of a split train set, parts in row order: strict < keeps the lower row on ties.
__global__ void merge_parts_kernel(const float *__restrict__ part_d, const int32_t *__restrict__ part_i, int parts,
                                   int64_t Nq, const float *__restrict__ qnorm, int32_t *__restrict__ idx,
                                   float *__restrict__ dist)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Nq) return;
    float d0 = INFINITY, d1 = INFINITY;
    int i0 = -1, i1 = -1;
    for (int p = 0; p < parts; ++p)
        for (int k = 0; k < 2; ++k) {
            const float d = part_d[((int64_t)p * Nq + q) * 2 + k];
            const int i = part_i[((int64_t)p * Nq + q) * 2 + k];
            if (i < 0) continue;
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
            else if (d < d1) { d1 = d; i1 = i; }
        }
    idx[2 * q] = i0;
    idx[2 * q + 1] = i1;
    dist[2 * q] = i0 >= 0 ? sqrtf(d0 - kBias + qnorm[q]) : INFINITY;
    dist[2 * q + 1] = i1 >= 0 ? sqrtf(d1 - kBias + qnorm[q]) : INFINITY;
}

int launch_f32(const float *query, int64_t Nq, const float *train, int64_t Nt, int D, int32_t *idx, float *dist,
               hipStream_t stream)
{
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && D >= 1, "Nq, Nt >= 0, D >= 1");
    MQS_ARG_CHECK(D <= 4096, "D <= 4096");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query && idx && dist && (Nt == 0 || train), "pointers must not be null");
    int tile_rows = 16384 / D;                           // 64 KiB of LDS
    if (tile_rows > 1024) tile_rows = 1024;
    if (tile_rows < 1) tile_rows = 1;
    const size_t lds = (size_t)tile_rows * D * sizeof(float);
    const unsigned grid = (unsigned)((Nq + kBlock / kLanesPerQuery - 1) / (kBlock / kLanesPerQuery));
    switch (D) {
    case 2: hipLaunchKernelGGL(knn2_f32_kernel<2>, dim3(grid), dim3(kBlock), lds, stream, query, Nq, train, Nt, D, tile_rows, idx, dist); break;
    case 3: hipLaunchKernelGGL(knn2_f32_kernel<3>, dim3(grid), dim3(kBlock), lds, stream, query, Nq, train, Nt, D, tile_rows, idx, dist); break;
    case 4: hipLaunchKernelGGL(knn2_f32_kernel<4>, dim3(grid), dim3(kBlock), lds, stream, query, Nq, train, Nt, D, tile_rows, idx, dist); break;
    default: hipLaunchKernelGGL(knn2_f32_kernel<0>, dim3(grid), dim3(kBlock), lds, stream, query, Nq, train, Nt, D, tile_rows, idx, dist); break;
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

constexpr int kMaxParts = 16;                            // parts of the train set (whole windows each) when the query blocks alone cannot fill the CUs

// Can workgroups of NW waves x QT query tiles, times the parts the train set can be split into, occupy every CU?
template <class TP>
bool tiles_fill(int64_t Nq, int64_t Nt, int NW, int QT, int num_cus)
{
    const int64_t qblocks = (Nq + NW * QT * 32 - 1) / (NW * QT * 32);
    int64_t nwin = (Nt + TP::kWindowTiles * 32 - 1) / (TP::kWindowTiles * 32);
    if (nwin > kMaxParts) nwin = kMaxParts;
    return qblocks * nwin >= num_cus;
}

template <class TP, int KS, int QT, int NW>
void launch_mfma_t(const typename TP::elem *q, int64_t Nq, const typename TP::elem *t, int64_t Nt, const float *qn,
                   const float *tn, int32_t *idx, float *dist, float *part_d, int32_t *part_i, int num_cus, hipStream_t stream)
{
    const int64_t per_block = NW * QT * 32;
    const int64_t qblocks = (Nq + per_block - 1) / per_block;
    // split the train rows (whole windows) until every CU has a workgroup
    const int64_t nwin = (Nt + TP::kWindowTiles * 32 - 1) / (TP::kWindowTiles * 32);
    int64_t parts = (num_cus + qblocks - 1) / qblocks;
    if (parts > nwin) parts = nwin;
    if (parts > kMaxParts) parts = kMaxParts;
    if (parts < 1) parts = 1;
    hipLaunchKernelGGL((knn2_mfma_kernel<TP, KS, QT, NW>), dim3((unsigned)qblocks, (unsigned)parts), dim3(NW * 64), 0, stream,
                       q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i);
    if (parts > 1)
        hipLaunchKernelGGL(merge_parts_kernel, dim3((unsigned)((Nq + 255) / 256)), dim3(256), 0, stream, part_d, part_i,
                           (int)parts, Nq, qn, idx, dist);
}

// Packed descriptor bits -> the FP4 operands of F4Path and the squared norm (= popcount): nibble k of the expanded row =
// bit k of the descriptor ? `code` : 0 (`code` = 0x2: 1.0 for train
// rows, 0xC: -2.0 for query rows), two nibbles per byte.  One thread per descriptor BYTE (one output dword): consecutive
// threads read consecutive bytes and write consecutive dwords (one thread per row wrote 128-byte rows 128 bytes apart:
// 17 us per 65 536 rows, 7 % of a matcher call; now 3-4); the row's popcount is summed over its D / 8 threads by shuffles.
__global__ void expand_bits_fp4_kernel(const uint8_t *__restrict__ bits, int64_t n, int D, unsigned code,
                                       signed char *__restrict__ out, float *__restrict__ norm)
{
    const int bpr = D / 8;                                 // bytes (= threads) per row: 16, 32 or 64 -- a power of two <= 64
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * bpr;
    unsigned v = 0u;
    if (t < total) {
        v = bits[t];
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) w |= ((v >> k) & 1u) ? (code << (4 * k)) : 0u;
        reinterpret_cast<unsigned *>(out)[t] = w;
    }
    int pop = __popc(v);
    for (int off = bpr / 2; off >= 1; off >>= 1) pop += __shfl_xor(pop, off, 64);
    if (t < total && (t & (bpr - 1)) == 0) norm[t / bpr] = (float)pop;
}

// ---- caller-side filter of the reference (Work/SLAM/application/own/slam.py:108-125) -------------------------------
// Per query: the radius filter of radiusMatch (cv2_helpers.py:311-331), then the ratio test (a single match inside the
// radius passes; two pass when d0 / d1 < ratio, the division in double like the Python floats of DMatch.distance), then one
// match per train index: the query with the smallest priority wins, the earlier query on equal priority (the reference
// replaces only on strict `<` while walking the queries in order).  One 64-bit atomicMin per passing query on the key
// (order-preserving image of the priority) << 32 | query -- the result does not depend on the execution order.
__device__ __forceinline__ uint32_t orderable_f32(float f)
{
    if (f != f) return 0xffffffffu;                     // NaN ranks last (never preferred)
    if (f == 0.0f) f = 0.0f;                            // -0.0 and +0.0 are equal for the reference's `<`
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void __launch_bounds__(256)
ratio_unique_scatter_kernel(const int32_t *__restrict__ idx, const float *__restrict__ dist, int64_t Nq, int64_t Nt,
                            float max_radius, double max_ratio, const float *__restrict__ priority,
                            unsigned long long *__restrict__ keys)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= Nq) return;
    const int2 i2 = reinterpret_cast<const int2 *>(idx)[q];
    const float2 d2 = reinterpret_cast<const float2 *>(dist)[q];
    const bool in0 = i2.x >= 0 && i2.x < Nt && d2.x <= max_radius;
    const bool in1 = i2.y >= 0 && d2.y <= max_radius;
    if (!in0) return;
    if (in1 && !((double)d2.x / (double)d2.y < max_ratio)) return;       // 0 / 0 (NaN) fails, like every NaN compare
    const float pr = priority ? priority[q] : d2.x;
    const unsigned long long key = ((unsigned long long)orderable_f32(pr) << 32) | (unsigned long long)(uint32_t)q;
    atomicMin(&keys[i2.x], key);
}

__global__ void __launch_bounds__(256)
ratio_unique_gather_kernel(const unsigned long long *__restrict__ keys, const float *__restrict__ dist, int64_t Nt,
                           int32_t *__restrict__ query_of_train, float *__restrict__ dist_of_train)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Nt) return;
    const unsigned long long k = keys[t];
    const bool has = k != ~0ull;
    const uint32_t q = (uint32_t)(k & 0xffffffffull);
    query_of_train[t] = has ? (int32_t)q : -1;
    if (dist_of_train) dist_of_train[t] = has ? dist[2 * (int64_t)q] : __builtin_inff();
}

}  // namespace

extern "C" {

int64_t mqs_match_ratio_unique_workspace_bytes(int64_t Nt) { return Nt < 0 ? 0 : (Nt + 1) * 8; }

int mqs_match_ratio_unique_dev(const int32_t *idx, const float *dist, int64_t Nq, int64_t Nt, float max_radius,
                               double max_dist_ratio, const float *priority, int32_t *query_of_train,
                               float *dist_of_train, void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && Nq <= 0x7fffffff, "0 <= Nq < 2^31, Nt >= 0");
    if (Nt == 0) return MQS_OK;
    MQS_ARG_CHECK(query_of_train && workspace && (Nq == 0 || (idx && dist)), "pointers must not be null");
    MQS_ARG_CHECK((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "workspace must be 8-byte aligned");
    MQS_ARG_CHECK(workspace_bytes >= mqs_match_ratio_unique_workspace_bytes(Nt), "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned long long *keys = static_cast<unsigned long long *>(workspace);
    MQS_HIP_CHECK(hipMemsetAsync(keys, 0xff, (size_t)Nt * 8, stream));
    if (Nq > 0)
        hipLaunchKernelGGL(ratio_unique_scatter_kernel, dim3((unsigned)((Nq + 255) / 256)), dim3(256), 0, stream, idx, dist, Nq,
                           Nt, max_radius, max_dist_ratio, priority, keys);
    hipLaunchKernelGGL(ratio_unique_gather_kernel, dim3((unsigned)((Nt + 255) / 256)), dim3(256), 0, stream, keys, dist, Nt,
                       query_of_train, dist_of_train);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_match_knn2_f32_dev(const float *query, int64_t Nq, const float *train, int64_t Nt, int D, int32_t *idx,
                           float *dist, void *stream)
{
    return launch_f32(query, Nq, train, Nt, D, idx, dist, static_cast<hipStream_t>(stream));
}

int64_t mqs_match_knn2_f16_workspace_bytes(int64_t Nq, int64_t Nt)
{
    if (Nq < 0 || Nt < 0) return 0;
    // squared norms + the per-part best two of a split train set (distance, index)
    return ((Nq + 63) / 64 * 64 + (Nt + 63) / 64 * 64) * (int64_t)sizeof(float) + (int64_t)kMaxParts * Nq * 2 * 8;
}

int mqs_match_knn2_f16_dev(const uint16_t *query, int64_t Nq, const uint16_t *train, int64_t Nt, int D, int32_t *idx,
                           float *dist, void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0, "Nq, Nt >= 0");
    MQS_ARG_CHECK(D == 32 || D == 64 || D == 128 || D == 256 || D == 512, "D must be 32, 64, 128, 256 or 512");
    MQS_ARG_CHECK(Nt <= 0x7fffffff, "Nt must be < 2^31");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query && idx && dist && workspace && (Nt == 0 || train), "pointers must not be null");
    MQS_ARG_CHECK(mqs_aligned16(query) && mqs_aligned16(train), "descriptor pointers must be 16-byte aligned");
    MQS_ARG_CHECK(workspace_bytes >= mqs_match_knn2_f16_workspace_bytes(Nq, Nt), "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const _Float16 *q = reinterpret_cast<const _Float16 *>(query);
    const _Float16 *t = reinterpret_cast<const _Float16 *>(train);
    float *qn = static_cast<float *>(workspace);
    float *tn = qn + (Nq + 63) / 64 * 64;
    float *part_d = tn + (Nt + 63) / 64 * 64;
    int32_t *part_i = reinterpret_cast<int32_t *>(part_d + (int64_t)kMaxParts * Nq * 2);
    int dev = 0, num_cus = 256;
    MQS_HIP_CHECK(hipGetDevice(&dev));
    MQS_HIP_CHECK(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((Nq + 255) / 256)), dim3(256), 0, stream, q, Nq, D, qn);
    if (Nt > 0)
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((Nt + 255) / 256)), dim3(256), 0, stream, t, Nt, D, tn);
    switch (D) {
    case 32: launch_mfma_t<F16Path, 2, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream); break;
    case 64: launch_mfma_t<F16Path, 4, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream); break;
    case 128: launch_mfma_t<F16Path, 8, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream); break;
    case 256:
        // two query tiles per wave halve the train stream per MFMA; worth it once the split can still fill the CUs
        if (MQS_MATCH_QT256 == 2 && tiles_fill<F16Path>(Nq, Nt, MQS_MATCH_NW256, 2, num_cus))
            launch_mfma_t<F16Path, 16, 2, MQS_MATCH_NW256>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        else
            launch_mfma_t<F16Path, 16, 1, MQS_MATCH_NW256>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        break;
    case 512: launch_mfma_t<F16Path, 32, 1, MQS_MATCH_NW512>(q, Nq, t, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream); break;
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// Packed binary descriptors (D bits per row, D / 8 bytes, D in {128, 256, 512}): Hamming distance = |q - t|^2 on the
// FP4 matrix path (F4Path), same output contract as the fp16 path (dist = sqrt(Hamming)).  65 536^2 x 256 bits: fp16 1.33 ms,
// FP4 0.45 ms (the int8 form this path had until round 2 -- v_mfma_i32_32x32x32_i8, 0.70 ms -- is gone from the source).
int64_t mqs_match_knn2_bits_workspace_bytes(int64_t Nq, int64_t Nt, int D)
{
    if (Nq < 0 || Nt < 0 || D < 8) return 0;
    if (D == 128) D = 256;                                  // (sized for one byte per bit, as the ABI promised before FP4: an upper bound)
    const int64_t up = 255;
    return (((Nq + 63) / 64 * 64 + (Nt + 63) / 64 * 64) * (int64_t)sizeof(float) + up) / 256 * 256 +
           (int64_t)kMaxParts * Nq * 2 * 8 + ((Nq * D + up) / 256 * 256) + ((Nt * D + up) / 256 * 256) + 256;
}

int mqs_match_knn2_bits_dev(const uint8_t *query_bits, int64_t Nq, const uint8_t *train_bits, int64_t Nt, int D, int32_t *idx,
                            float *dist, void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0, "Nq, Nt >= 0");
    MQS_ARG_CHECK(D == 128 || D == 256 || D == 512, "D must be 128, 256 or 512 bits");
    MQS_ARG_CHECK(Nt <= 0x7fffffff, "Nt must be < 2^31");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query_bits && idx && dist && workspace && (Nt == 0 || train_bits), "pointers must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_match_knn2_bits_workspace_bytes(Nq, Nt, D), "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *w = static_cast<char *>(workspace);
    float *qn = reinterpret_cast<float *>(w);
    float *tn = qn + (Nq + 63) / 64 * 64;
    w += (((Nq + 63) / 64 * 64 + (Nt + 63) / 64 * 64) * (int64_t)sizeof(float) + 255) / 256 * 256;
    float *part_d = reinterpret_cast<float *>(w);
    int32_t *part_i = reinterpret_cast<int32_t *>(part_d + (int64_t)kMaxParts * Nq * 2);
    w += (int64_t)kMaxParts * Nq * 2 * 8;
    signed char *q8 = reinterpret_cast<signed char *>(w); w += (Nq * (D == 128 ? 256 : D) + 255) / 256 * 256;
    signed char *t8 = reinterpret_cast<signed char *>(w);
    int dev = 0, num_cus = 256;
    MQS_HIP_CHECK(hipGetDevice(&dev));
    MQS_HIP_CHECK(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    // expanded rows are D / 2 bytes (two nibbles per byte); 128-bit descriptors need no padding: two FP4 MFMAs per tile cover them
    hipLaunchKernelGGL(expand_bits_fp4_kernel, dim3((unsigned)((Nq * (D / 8) + 255) / 256)), dim3(256), 0, stream, query_bits, Nq, D, 0xCu,
                       q8, qn);
    if (Nt > 0)
        hipLaunchKernelGGL(expand_bits_fp4_kernel, dim3((unsigned)((Nt * (D / 8) + 255) / 256)), dim3(256), 0, stream, train_bits, Nt, D,
                           0x2u, t8, tn);
    switch (D) {
    case 128:
        if (tiles_fill<F4Path>(Nq, Nt, MQS_MATCH_F4_NW, MQS_MATCH_F4_QT, num_cus))
            launch_mfma_t<F4Path, 2, MQS_MATCH_F4_QT, MQS_MATCH_F4_NW>(q8, Nq, t8, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        else
            launch_mfma_t<F4Path, 2, 1, 8>(q8, Nq, t8, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        break;
    case 256:
        if (tiles_fill<F4Path>(Nq, Nt, MQS_MATCH_F4_NW, MQS_MATCH_F4_QT, num_cus))
            launch_mfma_t<F4Path, 4, MQS_MATCH_F4_QT, MQS_MATCH_F4_NW>(q8, Nq, t8, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        else
            launch_mfma_t<F4Path, 4, 1, 8>(q8, Nq, t8, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream);
        break;
    case 512: launch_mfma_t<F4Path, 8, 2, 8>(q8, Nq, t8, Nt, qn, tn, idx, dist, part_d, part_i, num_cus, stream); break;
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_match_knn2_bits(mqs_ctx *ctx, const uint8_t *query_bits, int64_t Nq, const uint8_t *train_bits, int64_t Nt, int D,
                        int32_t *idx, float *dist)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && D >= 8 && D % 8 == 0, "Nq, Nt >= 0, D a multiple of 8");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query_bits && idx && dist && (Nt == 0 || train_bits), "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t wsb = (size_t)mqs_match_knn2_bits_workspace_bytes(Nq, Nt, D);
    const size_t o_q = 0, o_t = up((size_t)Nq * D / 8), o_i = up(o_t + (size_t)Nt * D / 8), o_d = up(o_i + (size_t)Nq * 8);
    const size_t o_w = up(o_d + (size_t)Nq * 8), total = up(o_w + wsb);
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_q, query_bits, (size_t)Nq * D / 8, hipMemcpyHostToDevice, ctx->stream));
    if (Nt > 0) MQS_HIP_CHECK(hipMemcpyAsync(d + o_t, train_bits, (size_t)Nt * D / 8, hipMemcpyHostToDevice, ctx->stream));
    rc = mqs_match_knn2_bits_dev((const uint8_t *)(d + o_q), Nq, (const uint8_t *)(d + o_t), Nt, D, (int32_t *)(d + o_i),
                                 (float *)(d + o_d), d + o_w, (int64_t)wsb, ctx->stream);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(idx, d + o_i, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipMemcpyAsync(dist, d + o_d, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

// host-pointer wrappers
int mqs_match_knn2_f32(mqs_ctx *ctx, const float *query, int64_t Nq, const float *train, int64_t Nt, int D,
                       int32_t *idx, float *dist)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && D >= 1, "Nq, Nt >= 0, D >= 1");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query && idx && dist && (Nt == 0 || train), "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_q = 0, o_t = up((size_t)Nq * D * 4), o_i = up(o_t + (size_t)Nt * D * 4), o_d = up(o_i + (size_t)Nq * 8);
    const size_t total = up(o_d + (size_t)Nq * 8);
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_q, query, (size_t)Nq * D * 4, hipMemcpyHostToDevice, ctx->stream));
    if (Nt > 0) MQS_HIP_CHECK(hipMemcpyAsync(d + o_t, train, (size_t)Nt * D * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_f32((const float *)(d + o_q), Nq, (const float *)(d + o_t), Nt, D, (int32_t *)(d + o_i),
                    (float *)(d + o_d), ctx->stream);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(idx, d + o_i, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipMemcpyAsync(dist, d + o_d, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

int mqs_match_knn2_f16(mqs_ctx *ctx, const uint16_t *query, int64_t Nq, const uint16_t *train, int64_t Nt, int D,
                       int32_t *idx, float *dist)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && D >= 1, "Nq, Nt >= 0, D >= 1");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query && idx && dist && (Nt == 0 || train), "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t wsb = (size_t)mqs_match_knn2_f16_workspace_bytes(Nq, Nt);
    const size_t o_q = 0, o_t = up((size_t)Nq * D * 2), o_i = up(o_t + (size_t)Nt * D * 2), o_d = up(o_i + (size_t)Nq * 8);
    const size_t o_w = up(o_d + (size_t)Nq * 8), total = up(o_w + wsb);
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_q, query, (size_t)Nq * D * 2, hipMemcpyHostToDevice, ctx->stream));
    if (Nt > 0) MQS_HIP_CHECK(hipMemcpyAsync(d + o_t, train, (size_t)Nt * D * 2, hipMemcpyHostToDevice, ctx->stream));
    rc = mqs_match_knn2_f16_dev((const uint16_t *)(d + o_q), Nq, (const uint16_t *)(d + o_t), Nt, D, (int32_t *)(d + o_i),
                                (float *)(d + o_d), d + o_w, (int64_t)wsb, ctx->stream);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(idx, d + o_i, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipMemcpyAsync(dist, d + o_d, (size_t)Nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

// radiusMatch + ratio test + one match per train index in one call (slam.py:101-125), float32 exact path
int mqs_match_radius_ratio_unique(mqs_ctx *ctx, const float *query, int64_t Nq, const float *train, int64_t Nt, int D,
                                  float max_radius, double max_dist_ratio, const float *priority, int32_t *query_of_train,
                                  float *dist_of_train)
{
    MQS_ARG_CHECK(ctx != nullptr, "ctx must not be null");
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0 && D >= 1 && Nq <= 0x7fffffff, "0 <= Nq < 2^31, Nt >= 0, D >= 1");
    if (Nt == 0) return MQS_OK;
    MQS_ARG_CHECK(train && query_of_train && (Nq == 0 || query), "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_q = 0, o_t = up((size_t)Nq * D * 4), o_i = up(o_t + (size_t)Nt * D * 4), o_d = up(o_i + (size_t)Nq * 8);
    const size_t o_p = up(o_d + (size_t)Nq * 8), o_k = up(o_p + (size_t)Nq * 4), o_o = up(o_k + (size_t)(Nt + 1) * 8);
    const size_t o_e = up(o_o + (size_t)Nt * 4), total = up(o_e + (size_t)Nt * 4);
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    if (Nq > 0) {
        MQS_HIP_CHECK(hipMemcpyAsync(d + o_q, query, (size_t)Nq * D * 4, hipMemcpyHostToDevice, ctx->stream));
        if (priority) MQS_HIP_CHECK(hipMemcpyAsync(d + o_p, priority, (size_t)Nq * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_t, train, (size_t)Nt * D * 4, hipMemcpyHostToDevice, ctx->stream));
    if (Nq > 0) {
        rc = launch_f32((const float *)(d + o_q), Nq, (const float *)(d + o_t), Nt, D, (int32_t *)(d + o_i), (float *)(d + o_d),
                        ctx->stream);
        if (rc != MQS_OK) return rc;
    }
    rc = mqs_match_ratio_unique_dev((const int32_t *)(d + o_i), (const float *)(d + o_d), Nq, Nt, max_radius, max_dist_ratio,
                                    priority ? (const float *)(d + o_p) : nullptr, (int32_t *)(d + o_o), (float *)(d + o_e),
                                    d + o_k, (int64_t)((Nt + 1) * 8), ctx->stream);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(query_of_train, d + o_o, (size_t)Nt * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dist_of_train) MQS_HIP_CHECK(hipMemcpyAsync(dist_of_train, d + o_e, (size_t)Nt * 4, hipMemcpyDeviceToHost, ctx->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MQS_OK;
}

}  // extern "C"
