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
__global__ __launch_bounds__(kBlock) void knn2_f32_kernel(const float *__restrict__ query, int64_t Nq,
                                                          const float *__restrict__ train, int64_t Nt, int D,
                                                          int tile_rows, int32_t *__restrict__ idx,
                                                          float *__restrict__ dist)
{
    extern __shared__ float sT[];                     // [tile_rows][D]
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * kBlock + tid;
    const bool live = q < Nq;
    const float *qrow = query + (live ? q : 0) * D;
    float b0 = INFINITY, b1 = INFINITY;
    int32_t i0 = -1, i1 = -1;
    for (int64_t t0 = 0; t0 < Nt; t0 += tile_rows) {
        const int rows = (Nt - t0) < tile_rows ? (int)(Nt - t0) : tile_rows;
        __syncthreads();
        for (int e = tid; e < rows * D; e += kBlock) sT[e] = train[t0 * D + e];
        __syncthreads();
        if (live) {
            for (int r = 0; r < rows; ++r) {
                const float *trow = sT + r * D;
                float s = 0.0f;
                for (int k = 0; k < D; ++k) {
                    const float d = qrow[k] - trow[k];
                    s = s + d * d;                     // contract(off): separate multiply and add
                }
                const int32_t j = (int32_t)(t0 + r);
                if (s < b0) { b1 = b0; i1 = i0; b0 = s; i0 = j; }
                else if (s < b1) { b1 = s; i1 = j; }
            }
        }
    }
    if (live) {
        idx[2 * q] = i0; idx[2 * q + 1] = i1;
        dist[2 * q] = (i0 >= 0) ? sqrtf(b0) : INFINITY;
        dist[2 * q + 1] = (i1 >= 0) ? sqrtf(b1) : INFINITY;
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
#define MQS_MATCH_QT256 1
#endif
#ifndef MQS_MATCH_NW256
#define MQS_MATCH_NW256 8
#endif
constexpr int kStageRows = 64;                         // train rows per LDS stage (2 MFMA row tiles)
// Key = (|t|^2 + kBias - 2 q.t) << 20 | train index.  |q|^2 is constant per lane (one query per
// lane), so it is left out of the running comparison and added back at the end; kBias keeps the
// biased distance non-negative (q.t <= |t|^2 <= D <= 512).  Valid rows: <= 1024; padding rows
// (|t|^2 := kPadNorm): in [2560, 3584]; 12 bits suffice.
constexpr unsigned kBias = 512;
constexpr unsigned kPadNorm = 3072;
constexpr unsigned kInvalidD2 = 2048;                  // biased distances >= this are padding rows
constexpr float kKeyScale = 1048576.0f;                // 2^20: float -> u32 conversion yields d2 << 20
constexpr unsigned kIdxBits = 20;

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

template <int KS /* D / 16 */, int QT /* 32-query column tiles per wave */, int NW /* waves per workgroup */>
__global__ __launch_bounds__(NW * 64) void knn2_f16_kernel(const _Float16 *__restrict__ query, int64_t Nq,
                                                          const _Float16 *__restrict__ train, int64_t Nt,
                                                          const float *__restrict__ qnorm,
                                                          const float *__restrict__ tnorm,
                                                          int32_t *__restrict__ idx, float *__restrict__ dist)
{
    constexpr int D = KS * 16;
    constexpr int kRowBytes = D * 2;                   // unpadded: the LDS image is filled by LDS-DMA
    constexpr int kStageBytes = kStageRows * kRowBytes;
    constexpr int kVecPerRow = D * 2 / 16;             // 16-byte pieces per row
    constexpr int kThreads = NW * 64;
    constexpr int kVecPerThread = kStageRows * kVecPerRow / kThreads;
    constexpr int kSwzMask = (kVecPerRow < 32 ? kVecPerRow : 32) - 1;   // XOR swizzle of the 16-B column
    static_assert(kStageRows * kVecPerRow % kThreads == 0 && kVecPerThread >= 1, "stage must divide over the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * kStageBytes];
    __shared__ __attribute__((aligned(16))) float sTn[3 * kStageRows];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t qbase = (int64_t)blockIdx.x * (NW * QT * 32) + wave * (QT * 32);

    // resident query fragments: B[k = 16 ks + 8 h + j][col r] = Q[qbase + 32 qt + r][...]
    half8 qf[QT][KS];
    float qn[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t q = qbase + 32 * qt + r;
        const bool ok = q < Nq;
        const half8 *row = reinterpret_cast<const half8 *>(query + (ok ? q : 0) * D);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 v = row[2 * ks + h];
            if (!ok) v = half8{0, 0, 0, 0, 0, 0, 0, 0};
            qf[qt][ks] = v;
        }
        qn[qt] = ok ? qnorm[q] : 0.0f;
    }
    unsigned best[QT], second[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { best[qt] = 0xFFFFFFFFu; second[qt] = 0xFFFFFFFFu; }

    const int64_t nstages = (Nt + kStageRows - 1) / kStageRows;
    const uint4 *tvec = reinterpret_cast<const uint4 *>(train);

    // Stage fill by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write).  One
    // wave-instruction writes 64 x 16 B = 1 KiB of LDS linearly (wave-uniform base + lane * 16), so
    // the swizzle that makes the fragment reads bank-conflict free is applied to the per-lane
    // SOURCE address: LDS piece (row, c') holds global piece (row, c' ^ (row & kSwzMask)); the
    // reader applies the same XOR.  Rows past Nt re-read the last row (their |t|^2 sentinel keeps
    // them out of the result).  Norms go to a 3-deep ring: a tile's scan runs one tile late.
    auto stage_issue = [&](int64_t s) {
        const int buf = (int)(s & 1);
#pragma unroll
        for (int i = 0; i < kVecPerThread; ++i) {
            const int piece0 = (i * NW + wave) * 64;              // first 16-B piece of this wave-instruction
            const int v = piece0 + lane;
            const int row = v / kVecPerRow, colp = v % kVecPerRow;
            const int col = colp ^ (row & kSwzMask);
            int64_t grow = s * kStageRows + row;
            grow = grow < Nt ? grow : Nt - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(tvec + grow * kVecPerRow + col),
                (__attribute__((address_space(3))) void *)(sTile + buf * kStageBytes + piece0 * 16), 16, 0, 0);
        }
        if (tid < kStageRows) {
            const int64_t t = s * kStageRows + tid;
            // (|t|^2 + kBias) * 2^20: the scan computes  key = u32(acc * (-2 * 2^20) + this) | index
            sTn[(s % 3) * kStageRows + tid] = (((t < Nt) ? tnorm[t] : (float)kPadNorm) + (float)kBias) * kKeyScale;
        }
    };

    // MFMAs of tile (s, tt) into `acc`, interleaved one-for-one with the top-2 scan of the PREVIOUS
    // tile's accumulators `prev` (32 MFMAs, 32 values): the scan's ~6 VALU instructions per value
    // issue in the shadow of the MFMA they are paired with instead of after the tile.
    auto tile_step = [&](const unsigned char *tile, int tt, float16v (&acc)[QT], const float16v (&prev)[QT],
                         const float *ptn, unsigned pbase) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[qt][e] = 0.0f;
        const unsigned char *arow = tile + (tt * 32 + r) * kRowBytes;
        const int swz = r & kSwzMask;                         // (tt * 32 + r) & kSwzMask
        float tnv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t4 = *reinterpret_cast<const float4 *>(ptn + 8 * g);
            tnv[4 * g] = t4.x; tnv[4 * g + 1] = t4.y; tnv[4 * g + 2] = t4.z; tnv[4 * g + 3] = t4.w;
        }
        constexpr int kSteps = KS * QT;                       // MFMAs in this tile
        constexpr int kVals = 16 * QT;                        // values to scan from the previous tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 a = *reinterpret_cast<const half8 *>(arow + (((2 * ks + h) ^ swz) << 4));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[qt][ks], acc[qt], 0, 0, 0);
                // scan values [v0, v1) of the previous tile behind this MFMA
                const int step = ks * QT + qt;
                const int v0 = step * kVals / kSteps, v1 = (step + 1) * kVals / kSteps;
#pragma unroll
                for (int v = v0; v < v1; ++v) {
                    const int pq = v / 16, e = v % 16;        // row(e) = (e & 3) + 8 (e >> 2) + 4 h
                    const float kf = fmaf(prev[pq][e], -2.0f * kKeyScale, tnv[e]);
                    const unsigned key = (unsigned)kf | pbase | (unsigned)(8 * (e >> 2) + (e & 3));   // disjoint bits: one v_or3_b32
                    unsigned m;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best[pq]), "v"(second[pq]), "v"(key));
                    second[pq] = m;
                    best[pq] = min(best[pq], key);
                }
            }
        }
    };

    // slot 2 of the norm ring doubles as the "no previous tile" source for the very first step
    if (tid < kStageRows) sTn[2 * kStageRows + tid] = ((float)kPadNorm + (float)kBias) * kKeyScale;
    if (nstages > 0) stage_issue(0);                     // Nt == 0: nothing to read, every key stays invalid
    __syncthreads();                                     // (waits for the LDS-DMA: vmcnt(0) + barrier)

    float16v accA[QT], accB[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int e = 0; e < 16; ++e) accB[qt][e] = 0.0f;
    const float *ptn = sTn + 2 * kStageRows + 4 * h;     // pending tile = none (all padding)
    unsigned pbase = 4 * h;

    for (int64_t s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) stage_issue(s + 1);          // lands while this stage is computed
        const unsigned char *tile = sTile + (int)(s & 1) * kStageBytes;
        const float *tn0 = sTn + (int)(s % 3) * kStageRows + 4 * h;
        const unsigned base0 = (unsigned)(s * kStageRows) + 4 * h;
        tile_step(tile, 0, accA, accB, ptn, pbase);       // tile 2s   <- scan of tile 2s-1
        tile_step(tile, 1, accB, accA, tn0, base0);       // tile 2s+1 <- scan of tile 2s
        ptn = tn0 + 32;
        pbase = base0 + 32;
        __syncthreads();
    }
    // scan of the last tile
    if (nstages > 0) {
        float tnv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t4 = *reinterpret_cast<const float4 *>(ptn + 8 * g);
            tnv[4 * g] = t4.x; tnv[4 * g + 1] = t4.y; tnv[4 * g + 2] = t4.z; tnv[4 * g + 3] = t4.w;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float kf = fmaf(accB[qt][e], -2.0f * kKeyScale, tnv[e]);
                const unsigned key = (unsigned)kf | pbase | (unsigned)(8 * (e >> 2) + (e & 3));   // disjoint bits: one v_or3_b32
                second[qt] = max(best[qt], min(second[qt], key));
                best[qt] = min(best[qt], key);
            }
    }

    // merge the two half-waves (same query, disjoint train rows), undo the bias, write
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const unsigned ob = __shfl_xor(best[qt], 32), os = __shfl_xor(second[qt], 32);
        const unsigned mb = min(best[qt], ob);
        const unsigned ms = min(max(best[qt], ob), min(second[qt], os));
        const int64_t q = qbase + 32 * qt + r;
        if (h == 0 && q < Nq) {
            const unsigned db = mb >> kIdxBits, ds = ms >> kIdxBits;
            const bool vb = db < kInvalidD2, vs = ds < kInvalidD2;
            idx[2 * q] = vb ? (int32_t)(mb & ((1u << kIdxBits) - 1)) : -1;
            idx[2 * q + 1] = vs ? (int32_t)(ms & ((1u << kIdxBits) - 1)) : -1;
            dist[2 * q] = vb ? sqrtf((float)db - (float)kBias + qn[qt]) : INFINITY;
            dist[2 * q + 1] = vs ? sqrtf((float)ds - (float)kBias + qn[qt]) : INFINITY;
        }
    }
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
    hipLaunchKernelGGL(knn2_f32_kernel, dim3((unsigned)((Nq + kBlock - 1) / kBlock)), dim3(kBlock), lds, stream, query,
                       Nq, train, Nt, D, tile_rows, idx, dist);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

template <int KS, int QT, int NW>
void launch_f16_t(const _Float16 *q, int64_t Nq, const _Float16 *t, int64_t Nt, const float *qn, const float *tn,
                  int32_t *idx, float *dist, hipStream_t stream)
{
    const int64_t per_block = NW * QT * 32;
    hipLaunchKernelGGL((knn2_f16_kernel<KS, QT, NW>), dim3((unsigned)((Nq + per_block - 1) / per_block)),
                       dim3(NW * 64), 0, stream, q, Nq, t, Nt, qn, tn, idx, dist);
}

}  // namespace

extern "C" {

int mqs_match_knn2_f32_dev(const float *query, int64_t Nq, const float *train, int64_t Nt, int D, int32_t *idx,
                           float *dist, void *stream)
{
    return launch_f32(query, Nq, train, Nt, D, idx, dist, static_cast<hipStream_t>(stream));
}

int64_t mqs_match_knn2_f16_workspace_bytes(int64_t Nq, int64_t Nt)
{
    if (Nq < 0 || Nt < 0) return 0;
    return ((Nq + 63) / 64 * 64 + (Nt + 63) / 64 * 64) * (int64_t)sizeof(float);
}

int mqs_match_knn2_f16_dev(const uint16_t *query, int64_t Nq, const uint16_t *train, int64_t Nt, int D, int32_t *idx,
                           float *dist, void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(Nq >= 0 && Nt >= 0, "Nq, Nt >= 0");
    MQS_ARG_CHECK(D == 32 || D == 64 || D == 128 || D == 256 || D == 512, "D must be 32, 64, 128, 256 or 512");
    MQS_ARG_CHECK(Nt < (1 << kIdxBits), "Nt must be < 2^20");
    if (Nq == 0) return MQS_OK;
    MQS_ARG_CHECK(query && idx && dist && workspace && (Nt == 0 || train), "pointers must not be null");
    MQS_ARG_CHECK(mqs_aligned16(query) && mqs_aligned16(train), "descriptor pointers must be 16-byte aligned");
    MQS_ARG_CHECK(workspace_bytes >= mqs_match_knn2_f16_workspace_bytes(Nq, Nt), "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const _Float16 *q = reinterpret_cast<const _Float16 *>(query);
    const _Float16 *t = reinterpret_cast<const _Float16 *>(train);
    float *qn = static_cast<float *>(workspace);
    float *tn = qn + (Nq + 63) / 64 * 64;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((Nq + 255) / 256)), dim3(256), 0, stream, q, Nq, D, qn);
    if (Nt > 0)
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((Nt + 255) / 256)), dim3(256), 0, stream, t, Nt, D, tn);
    switch (D) {
    case 32: launch_f16_t<2, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, stream); break;
    case 64: launch_f16_t<4, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, stream); break;
    case 128: launch_f16_t<8, 2, 4>(q, Nq, t, Nt, qn, tn, idx, dist, stream); break;
    case 256: launch_f16_t<16, MQS_MATCH_QT256, MQS_MATCH_NW256>(q, Nq, t, Nt, qn, tn, idx, dist, stream); break;
    case 512: launch_f16_t<32, 1, 4>(q, Nq, t, Nt, qn, tn, idx, dist, stream); break;
    }
    MQS_HIP_CHECK(hipGetLastError());
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

}  // extern "C"
