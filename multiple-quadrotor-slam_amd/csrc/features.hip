// Image front-end of the per-frame loop on gfx950 (MI355X) -- SURVEY.md 8(f) rank 4:
//   goodFeaturesToTrack   Work/python_libs/cv2_helpers.py:34-37; Work/SLAM/application/own/slam2.py:665, 1174
//   calcOpticalFlowPyrLK  slam2.py:381
// OpenCV 2.4 is not vendored; the kernels follow its published method as restated (float32, fixed operation
// order) in oracle/features_np.py -- parity with OpenCV itself is unpinned (no images, no golden output in the
// reference).  Integer stages (pyramid, Scharr derivatives) are bit-exact against the oracle, the corner response is
// bit-exact float32 (no FMA contraction), the tracker agrees to float32 rounding of its window sums.
//
// Mapping to the machine
//   * corner response: one thread per pixel, the 5x5 support read straight through L1/L2 (a VGA frame is 300 KB);
//     two-stage max; candidates (thresholded 3x3 maxima under the mask) compacted with one atomic counter as 64-bit
//     keys (response bits << 32 | ~(y << 16 | x)), sorted by the selection kernel's own workgroup (bitonic network in
//     LDS, 16 384 keys at a time, merged when there are more) -- response descending, position ascending, so the
//     result does not depend on the order the atomics happened in.  (Round 1 called rocPRIM's radix sort over all
//     W x H key slots: eight passes over 307 200 keys for the few thousand candidates of a VGA frame, ~100 us in a
//     dozen launches; the library is gone from the product.)  The response maximum is an atomic max of order-
//     preserving keys inside the response kernel: no separate reduction launches;
//   * minimum-distance selection: the greedy rule (a candidate is a corner unless a stronger corner is closer than
//     min_distance) walks the sorted list, and a candidate depends only on candidates in front of it: the list goes through in
//     groups of 64 -- every candidate's mask of stronger group members within the distance built by all wavefronts (1024
//     candidates per pass), then one wavefront settles the groups in order against the grid of accepted corners (in LDS when it
//     fits) and on wave-wide masks inside a group (round 6; rounds 3-5: a parallel fixed point over pre-filtered survivors);
//   * Lucas-Kanade: one wavefront per feature, lanes strided over the 21 x 21 window (7 pixels each, template and
//     gradients in registers), all pyramid levels and all iterations inside one launch; window sums in fp64.
#include <cstring>
#include "mqs_common.h"
#include "wave_reduce.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// ---------------------------------------------------------------------------------------------------
// goodFeaturesToTrack
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sobel_scaled(const uint8_t *__restrict__ img, int W, int H, int y, int x, float &dx, float &dy)
{
    const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H), xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
    const float a = img[ym * W + xm], b = img[ym * W + x], c = img[ym * W + xp];
    const float d = img[y * W + xm], f = img[y * W + xp];
    const float g = img[yp * W + xm], h = img[yp * W + x], k = img[yp * W + xp];
    const float scale = 1.0f / (4.0f * 3.0f * 255.0f);
    dx = (((c - a) + 2.0f * (f - d)) + (k - g)) * scale;
    dy = (((g - a) + 2.0f * (h - b)) + (k - c)) * scale;
}

// float <-> unsigned key with the same order (negative responses can only come from rounding; they must still lose)
__device__ __forceinline__ unsigned int float_order_key(float v)
{
    const unsigned int b = __float_as_uint(v);
    return b ^ ((unsigned int)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(unsigned int k)
{
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// eig = the smaller eigenvalue of the 3 x 3 box sum of the gradient products; maxkey[0 .. kMaxSlots) (zeroed by the caller) end
// as order keys whose maximum is the frame's largest response: one atomic per workgroup, spread over kMaxSlots words -- 1 200
// atomics on ONE word took 11 of this kernel's 17 us (same-address atomics retire one every ~10 ns).
constexpr int kMaxSlots = 256;
// Under a mask the maximum is that of the UNMASKED pixels (OpenCV 2.4 featureselect.cpp: `minMaxLoc(eig, 0, &maxVal, 0, 0, mask)`;
// the reference always hands a mask over -- slam2.py:1173-1174, 662-665 -- and the strongest corners are the ones it covers).
__global__ __launch_bounds__(kBlock) void min_eig_kernel(const uint8_t *__restrict__ img, int W, int H, float *__restrict__ eig,
                                                        const uint8_t *__restrict__ mask, unsigned int *__restrict__ maxkey)
{
    // A 32 x 8 pixel tile per workgroup.  The gradient products of the tile and its one-pixel halo (34 x 10) are formed once,
    // in LDS: one Sobel evaluation per halo pixel (1.33 per thread) instead of nine per pixel (72 image loads per thread,
    // 17.5 us per VGA frame).  Same float32 expressions in the same order.
    __shared__ float sXX[10][34], sXY[10][34], sYY[10][34];
    __shared__ unsigned int sMax[kBlock / 64];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    // gradient products at halo position (hy, hx) <-> image pixel (reflect(y0 - 1 + hy), reflect(x0 - 1 + hx))
    for (int e = threadIdx.x; e < 10 * 34; e += kBlock) {
        const int hy = e / 34, hx = e % 34;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        float dx = 0.0f, dy = 0.0f;
        if (gy >= -1 && gy <= H && gx >= -1 && gx <= W)           // what some pixel of the image needs (a tile may overhang it)
            sobel_scaled(img, W, H, reflect101(gy, H), reflect101(gx, W), dx, dy);
        sXX[hy][hx] = dx * dx; sXY[hy][hx] = dx * dy; sYY[hy][hx] = dy * dy;
    }
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    unsigned int key = 0u;
    if (x < W && y < H) {
        float rxx[3], rxy[3], ryy[3];
#pragma unroll
        for (int oy = 0; oy < 3; ++oy) {
            rxx[oy] = (sXX[ty + oy][tx] + sXX[ty + oy][tx + 1]) + sXX[ty + oy][tx + 2];
            rxy[oy] = (sXY[ty + oy][tx] + sXY[ty + oy][tx + 1]) + sXY[ty + oy][tx + 2];
            ryy[oy] = (sYY[ty + oy][tx] + sYY[ty + oy][tx + 1]) + sYY[ty + oy][tx + 2];
        }
        const float a = ((rxx[0] + rxx[1]) + rxx[2]) * 0.5f;
        const float b = (rxy[0] + rxy[1]) + rxy[2];
        const float c = ((ryy[0] + ryy[1]) + ryy[2]) * 0.5f;
        const float d = a - c;
        const float e = (a + c) - sqrtf(d * d + b * b);
        eig[y * W + x] = e;
        if (!mask || mask[y * W + x] != 0) key = float_order_key(e);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)key, off, 64);
        key = key > o ? key : o;
    }
    if ((threadIdx.x & 63) == 0) sMax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int m = sMax[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) m = m > sMax[w] ? m : sMax[w];
        atomicMax(maxkey + ((blockIdx.y * gridDim.x + blockIdx.x) & (kMaxSlots - 1)), m);
    }
}

// Candidates (thresholded 3 x 3 maxima under the mask) of a 32 x 8 tile go to the tile's OWN segment of 256 key slots, in
// thread order, and the tile's count to wg_count[tile]: no global counter (3 000 atomics on one word were 13 of this kernel's
// 16 us).  The selection kernel gathers the segments.
__global__ __launch_bounds__(kBlock) void candidates_kernel(const float *__restrict__ eig, int W, int H,
                                                           const unsigned int *__restrict__ maxkey, float quality,
                                                           const uint8_t *__restrict__ mask, unsigned long long *__restrict__ keys,
                                                           unsigned int *__restrict__ wg_count)
{
    __shared__ unsigned int sRed[kBlock / 64];
    __shared__ int sCnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the frame's largest response: maximum over the slots
    unsigned int mk = maxkey[threadIdx.x & (kMaxSlots - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)mk, off, 64);
        mk = mk > o ? mk : o;
    }
    if (lane == 0) sRed[wave] = mk;
    __syncthreads();
#pragma unroll
    for (int w2 = 0; w2 < kBlock / 64; ++w2) mk = mk > sRed[w2] ? mk : sRed[w2];
    const float thr = float_from_order_key(mk) * quality;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    bool cand = false;
    float e = 0.0f;
    if (!(x <= 0 || y <= 0 || x >= W - 1 || y >= H - 1)) {
        e = eig[y * W + x];
        cand = (e > thr) && e != 0.0f && !(mask && mask[y * W + x] == 0);
        if (cand) {
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                for (int ox = -1; ox <= 1; ++ox) {
                    const float v = eig[(y + oy) * W + x + ox];          // in range: (x, y) is not on the border
                    if ((v > thr ? v : 0.0f) > e) cand = false;
                }
        }
    }
    const unsigned long long m = __ballot(cand);
    if (lane == 0) sCnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < kBlock / 64; ++w2) {
        off += w2 < wave ? sCnt[w2] : 0;
        total += sCnt[w2];
    }
    const unsigned int tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (cand)
        // low word: ~(y << 16 | x) -- the same order as the row-major position, and no division to take apart
        keys[(size_t)tile * kBlock + off + __popcll(m & ((1ull << lane) - 1ull))] =
            ((unsigned long long)__float_as_uint(e) << 32) | (unsigned long long)(0xFFFFFFFFu - (((unsigned)y << 16) | (unsigned)x));
    if (threadIdx.x == 0) wg_count[tile] = (unsigned int)total;
}

// Minimum-distance selection over the sorted candidates (one workgroup).  grid: cells x 4 slots (x | y << 16, 0xFFFFFFFF =
// empty) holding the accepted corners, in dynamic LDS (IN_LDS) or in the workspace.  The greedy rule is evaluated as a
// parallel fixed point, 1024 candidates per round -- see the comment inside the kernel.
// First the gather of the tiles' candidate segments (prefix sum of their counts) and the sort, by the same (single) workgroup.  Up to
// kSortChunk keys are sorted in LDS by a bitonic network (descending; the keys are distinct); more than that -- a frame of
// noise -- chunk by chunk into `keys` itself, then merged pairwise between `keys` and `tmp` (merge path: every thread finds
// its share of the output by bisection).  The LDS the network used is then the selection grid.
constexpr int kSelThreads = 1024;
constexpr int kSortChunk = 16384;         // 128 KB of 64-bit keys

__device__ __forceinline__ void bitonic_sort_desc_lds(unsigned long long *sK, int m /* power of two */, int tid)
{
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m / 2; i += kSelThreads) {
                const int lo = 2 * i - (i & (j - 1)), hi = lo + j;         // lo has bit j clear
                const unsigned long long a = sK[lo], b = sK[hi];
                const bool desc = (lo & k) == 0;                            // this run is sorted descending
                if ((a < b) == desc) { sK[lo] = b; sK[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// merges the descending runs src[a0 .. a0 + la) and src[a0 + la .. a0 + la + lb) into dst[a0 ..): all threads
__device__ __forceinline__ void merge_runs_desc(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst,
                                                unsigned int a0, unsigned int la, unsigned int lb, int tid)
{
    const unsigned long long *A = src + a0, *B = src + a0 + la;
    const unsigned int L = la + lb;
    const unsigned int d0 = (unsigned int)(((unsigned long long)L * tid) / kSelThreads);
    const unsigned int d1 = (unsigned int)(((unsigned long long)L * (tid + 1)) / kSelThreads);
    if (d0 == d1) return;
    unsigned int lo = d0 > lb ? d0 - lb : 0u, hi = d0 < la ? d0 : la;      // elements of A among the first d0 outputs
    while (lo < hi) {
        const unsigned int mid = (lo + hi) >> 1;
        if (A[mid] > B[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    unsigned int i = lo, j = d0 - lo;
    for (unsigned int d = d0; d < d1; ++d) {
        const bool takeA = j >= lb || (i < la && A[i] > B[j]);
        dst[a0 + d] = takeA ? A[i] : B[j];
        i += takeA; j += !takeA;
    }
}

// A/B builds only (-DMQS_GFTT_STAMPS): the kernel's phases in 100 MHz ticks, printed by thread 0 (tools/probes: gather, sort, rounds)
#ifdef MQS_GFTT_STAMPS
#define MQS_GSTAMP(i) do { if (threadIdx.x == 0) gst[i] = wall_clock64(); } while (0)
#else
#define MQS_GSTAMP(i) do { } while (0)
#endif
template <bool IN_LDS>
__global__ __launch_bounds__(kSelThreads) void sort_select_kernel(unsigned long long *__restrict__ keys_seg,
                                                                  unsigned long long *__restrict__ keys_io,
                                                                  const unsigned int *__restrict__ wg_count, unsigned int ntiles, int W,
                                                                  int H, float min_distance, int max_corners, int out_capacity,
                                                                  unsigned int *__restrict__ grid_global, float *__restrict__ out_xy,
                                                                  int *__restrict__ out_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned int sGrid[];
    const unsigned long long *keys = keys_io;
    __shared__ unsigned int sScan[kSelThreads / 64];
    __shared__ unsigned int sTotal;
    unsigned long long *tmp = keys_seg;                    // the segments are free once gathered: merge buffer
#ifdef MQS_GFTT_STAMPS
    unsigned long long gst[6] = {0, 0, 0, 0, 0, 0};
    int n_rounds = 0;
#endif
    MQS_GSTAMP(0);
    {
        // gather the tiles' segments into keys_io, tile after tile: thread t owns a contiguous range of tiles
        const int tid = threadIdx.x;
        const unsigned int per = (ntiles + kSelThreads - 1) / kSelThreads;
        const unsigned int t0 = tid * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
        unsigned int mine = 0;
        for (unsigned int t = t0; t < t1; ++t) mine += wg_count[t];
        unsigned int incl = mine;                          // inclusive scan over the workgroup
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = (unsigned int)__shfl_up((int)incl, off, 64);
            if ((tid & 63) >= off) incl += o;
        }
        if ((tid & 63) == 63) sScan[tid >> 6] = incl;
        __syncthreads();
        unsigned int base = 0;
#pragma unroll
        for (int w2 = 0; w2 < kSelThreads / 64; ++w2) base += w2 < (tid >> 6) ? sScan[w2] : 0u;
        if (tid == kSelThreads - 1) sTotal = base + incl;
        unsigned int pos = base + incl - mine;
        for (unsigned int t = t0; t < t1; ++t) {
            const unsigned int c = wg_count[t];
            for (unsigned int k = 0; k < c; ++k) keys_io[pos + k] = keys_seg[(size_t)t * kBlock + k];
            pos += c;
        }
        __syncthreads();
    }
    MQS_GSTAMP(1);
    {
        unsigned long long *sK = reinterpret_cast<unsigned long long *>(sGrid);
        const unsigned int cnt = sTotal;
        const int tid = threadIdx.x;
        for (unsigned int c0 = 0; c0 < cnt; c0 += kSortChunk) {
            const unsigned int len = cnt - c0 < (unsigned)kSortChunk ? cnt - c0 : (unsigned)kSortChunk;
            int m = 2;
            while ((unsigned)m < len) m <<= 1;
            for (int i = tid; i < m; i += kSelThreads) sK[i] = (unsigned)i < len ? keys_io[c0 + i] : 0ull;   // 0 sorts last, is no key
            __syncthreads();
            bitonic_sort_desc_lds(sK, m, tid);
            for (unsigned int i = tid; i < len; i += kSelThreads) keys_io[c0 + i] = sK[i];
            __syncthreads();
        }
        if (cnt > (unsigned)kSortChunk) {
            unsigned long long *src = keys_io, *dst = tmp;
            __threadfence_block();
            for (unsigned int run = kSortChunk; run < cnt; run <<= 1) {
                __syncthreads();
                for (unsigned int a0 = 0; a0 < cnt; a0 += 2 * run) {
                    const unsigned int la = cnt - a0 < run ? cnt - a0 : run;
                    const unsigned int lb = cnt - a0 - la < run ? cnt - a0 - la : run;
                    merge_runs_desc(src, dst, a0, la, lb, tid);
                }
                unsigned long long *t = src; src = dst; dst = t;
            }
            keys = src;
        }
        __syncthreads();                                   // the sorted keys are in global memory; the LDS is free
    }
    MQS_GSTAMP(2);
#ifdef MQS_GFTT_EXPERIMENT_SORT_ONLY                        // (timing only: the gather and the sort, no selection)
    if (threadIdx.x == 0) out_n[0] = (int)sTotal;
    if (W > 0) return;
#endif
    __shared__ unsigned long long sMask[kSelThreads];      // a candidate's stronger candidates of its own group of 64 within min_distance
    __shared__ unsigned int sPos[kSelThreads];
    __shared__ int sAccepted;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int n = sTotal;
    int accepted = 0;                                      // the same value in every thread
    const int limit = (max_corners > 0 && max_corners < out_capacity) ? max_corners : out_capacity;
    if (min_distance < 1.0f) {
        for (unsigned int i = threadIdx.x; i < n && (int)i < limit; i += kSelThreads) {
            const unsigned int pos = 0xFFFFFFFFu - (unsigned int)(keys[i] & 0xFFFFFFFFull);
            out_xy[2 * i] = (float)(pos & 0xFFFFu);
            out_xy[2 * i + 1] = (float)(pos >> 16);
        }
        if (threadIdx.x == 0) out_n[0] = (int)(n < (unsigned)limit ? n : (unsigned)limit);
        return;
    }
    const int cell = (int)rintf(min_distance);
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    auto slot_ptr = [&](int i) -> unsigned int * {
        if constexpr (IN_LDS) return &sGrid[i];
        else return &grid_global[i];
    };
    for (int i = threadIdx.x; i < gw * gh * 4; i += kSelThreads) *slot_ptr(i) = 0xFFFFFFFFu;
    __syncthreads();
    const float md2 = min_distance * min_distance, inv_cell = 1.0f / (float)cell;
    // floor(v / cell) without an integer division (v < 65536 is exact in float; one correction step)
    auto cell_of = [&](int v) {
        int c = (int)((float)v * inv_cell);
        c += ((c + 1) * cell <= v) - (c * cell > v);
        return c;
    };
    // The greedy rule -- a candidate is a corner unless a STRONGER corner lies closer than min_distance -- walks the sorted list, and a
    // candidate's fate depends only on candidates in front of it.  The list goes through in GROUPS OF 64, in order:
    //   A. (all sixteen wavefronts, one group each, 1024 candidates per pass) every candidate finds the stronger candidates OF ITS OWN
    //      GROUP within min_distance: 64 broadcasts of a packed position (v_readlane), one bit each -> a 64-bit mask per candidate;
    //   B. (one wavefront, the pass's groups in order) a candidate closer than min_distance to a corner accepted in an EARLIER group is
    //      out (the grid of accepted corners: four slots per cell, nine cells); the rest of the group is settled on wave-wide masks --
    //      a stronger member of the group accepted -> rejected; all of them rejected, or none -> accepted; else wait -- which ends
    //      after at most 64 looks (the strongest undecided member always decides; on images after two or three); the accepted ones, in
    //      order, up to the limit, go to the output and into the grid, where the next group's look finds them (a wavefront's LDS
    //      operations execute in order: no barrier inside B).
    // (History.  Round 2: one wavefront, seven candidates at a time against the grid: 85 us for 1100 candidates.  Rounds 3-5: a parallel
    // fixed point over 1024 survivors of a pre-filter per round -- hash table of the survivors by cell, per-survivor lists of up to eight
    // stronger neighbours, the states settled by sweeps with barriers, then by every survivor polling its neighbours in LDS: on the example
    // sequence 20-45 us per round BEFORE the first decision and 12-80 us of decisions -- along an edge most candidates have more than eight
    // stronger ones within twelve pixels and walk the buckets at every look -- 25-170 us per detection.  What is looked at here is what
    // the rule needs: accepted corners, at most 36 slots, and the 63 other members of the group.)
    for (unsigned int base = 0; base < n && accepted < limit; base += kSelThreads) {
        const unsigned int ci = base + threadIdx.x;
        const unsigned int cpos = ci < n ? 0xFFFFFFFFu - (unsigned int)(keys[ci] & 0xFFFFFFFFull) : 0u;
        {
            // closeness is symmetric: the wavefront's vote on "closer than min_distance to member j" IS member j's row (no branches, no
            // 64-bit selects: the compare's mask goes into lane j)
            const int px = (int)(cpos & 0xFFFFu), py = (int)(cpos >> 16);
            unsigned int mlo = 0u, mhi = 0u;
            if (base + 64u * wave < n)                         // (wavefront-uniform: a group beyond the list has nothing to find)
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const unsigned int pj = (unsigned int)__builtin_amdgcn_readlane((int)cpos, j);
                const float dx = (float)(px - (int)(pj & 0xFFFFu)), dy = (float)(py - (int)(pj >> 16));
                const unsigned long long row = __builtin_amdgcn_ballot_w64(dx * dx + dy * dy < md2);
                mlo = lane == j ? (unsigned int)row : mlo;
                mhi = lane == j ? (unsigned int)(row >> 32) : mhi;
            }
            sMask[threadIdx.x] = (((unsigned long long)mhi << 32) | mlo) & ((1ull << lane) - 1ull);      // the STRONGER members only
            sPos[threadIdx.x] = cpos;
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long below = (1ull << lane) - 1ull;
            for (int g = 0; g < kSelThreads / 64 && base + 64u * g < n && accepted < limit; ++g) {
                const int t = g * 64 + lane;
                const bool valid = base + t < n;
                const unsigned int pos = sPos[t];
                const unsigned long long m = sMask[t];
                const int px = (int)(pos & 0xFFFFu), py = (int)(pos >> 16);
                const int pcx = cell_of(px), pcy = cell_of(py);
                bool clash = false;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    // (every cell read, unconditionally -- a cell off the grid reads cell 0 and is ignored: 36 loads in flight together)
                    const int xx = pcx + k % 3 - 1, yy = pcy + k / 3 - 1;
                    const bool inside = yy >= 0 && yy < gh && xx >= 0 && xx < gw;
                    const unsigned int *cp = slot_ptr(inside ? (yy * gw + xx) * 4 : 0);
                    unsigned int v[4];
                    if constexpr (IN_LDS) {
                        const uint4 q = *reinterpret_cast<const uint4 *>(cp);
                        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                    } else {
                        // the grid in the workspace: written by this wavefront's atomics (at the L2), read past the vector cache
#pragma unroll
                        for (int slot = 0; slot < 4; ++slot) v[slot] = __hip_atomic_load(cp + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int slot = 0; slot < 4; ++slot) {
                        const float dx = (float)(px - (int)(v[slot] & 0xFFFFu)), dy = (float)(py - (int)(v[slot] >> 16));
                        clash = clash || (inside && v[slot] != 0xFFFFFFFFu && dx * dx + dy * dy < md2);
                    }
                }
                bool rej = !valid || clash, acc = false;
                unsigned long long A = 0ull, R = __builtin_amdgcn_ballot_w64(rej);
                for (int look = 0; look < 64 && (A | R) != ~0ull; ++look) {
                    if (!rej && !acc) {
                        if (m & A) rej = true;
                        else if ((m & ~R) == 0ull) acc = true;
                    }
                    A = __builtin_amdgcn_ballot_w64(acc);
                    R = __builtin_amdgcn_ballot_w64(rej);
                }
                const int arank = accepted + __popcll(A & below);
                if (acc && arank < limit) {
                    out_xy[2 * arank] = (float)px;
                    out_xy[2 * arank + 1] = (float)py;
                    // a free slot of the corner's cell (corners >= cell - 0.5 apart: at most four per cell)
                    unsigned int *slots = slot_ptr((pcy * gw + pcx) * 4);
                    for (int slot = 0; slot < 4; ++slot)
                        if (atomicCAS(slots + slot, 0xFFFFFFFFu, pos) == 0xFFFFFFFFu) break;
                }
                accepted += __popcll(A);
                if (accepted > limit) accepted = limit;
            }
            if (lane == 0) sAccepted = accepted;
        }
        __syncthreads();
        accepted = sAccepted;
#ifdef MQS_GFTT_STAMPS
        n_rounds += 1;
#endif
    }
    if (threadIdx.x == 0) out_n[0] = accepted;
#ifdef MQS_GFTT_STAMPS
    MQS_GSTAMP(3);
    if (threadIdx.x == 0) printf("GFTT candidates %u accepted %d limit %d passes %d  gather %.1f sort %.1f select %.1f us\n", n, accepted, limit, n_rounds,
                                 (gst[1] - gst[0]) / 100.0, (gst[2] - gst[1]) / 100.0, (gst[3] - gst[2]) / 100.0);
#endif
}

// ---------------------------------------------------------------------------------------------------
// Pyramid and derivatives (integer, exact)
// ---------------------------------------------------------------------------------------------------
// One launch per pyramid level instead of three: blockIdx.z = 0 takes the Scharr derivatives of level l of the previous
// image, z = 1 / 2 reduce level l of the previous / next image to level l + 1 (all three only read level l).  The grid is
// sized for the derivative image; the two reductions use its first quarter.
__global__ __launch_bounds__(kBlock) void pyr_level_kernel(const uint8_t *__restrict__ I, const uint8_t *__restrict__ J, int W, int H,
                                                          short2 *__restrict__ dI, uint8_t *__restrict__ Id, uint8_t *__restrict__ Jd,
                                                          int Wd, int Hd)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (blockIdx.z == 0) {
        if (x >= W || y >= H) return;
        const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H), xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
        const int a = I[ym * W + xm], b = I[ym * W + x], c = I[ym * W + xp];
        const int e = I[y * W + xm], f = I[y * W + xp];
        const int g = I[yp * W + xm], h = I[yp * W + x], k = I[yp * W + xp];
        dI[y * W + x] = make_short2((short)(3 * (c - a) + 10 * (f - e) + 3 * (k - g)), (short)(3 * (g - a) + 10 * (h - b) + 3 * (k - c)));
        return;
    }
    if (Id == nullptr || x >= Wd || y >= Hd) return;
    const uint8_t *src = blockIdx.z == 1 ? I : J;
    uint8_t *dst = blockIdx.z == 1 ? Id : Jd;
    const int k5[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int yy = reflect101(2 * y + j - 2, H);
        int row = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) row += k5[i] * (int)src[yy * W + reflect101(2 * x + i - 2, W)];
        acc += k5[j] * row;
    }
    dst[y * Wd + x] = (uint8_t)((acc + 128) >> 8);
}

// ---------------------------------------------------------------------------------------------------
// Pyramidal Lucas-Kanade
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxLevels = 8;
constexpr int kMaxWinPixelsPerLane = 16;      // windows up to 31 x 31 (961 pixels / 64 lanes)

// The tracker reads BORDER-EXTENDED copies of the pyramid levels, as OpenCV 2.4's does (lkpyramid.cpp: buildOpticalFlowPyramid pads
// every level by winSize with BORDER_REFLECT_101, the derivative image with BORDER_CONSTANT zeros): a window may leave the image by
// up to its own size, and every read is a plain one.  I / J / dI point at the INTERIOR origin of a level's padded copy, P is its
// row pitch (W + 2 kLkBorder).
constexpr int kLkBorder = 32;                  // >= the largest window (31) + 1 for the bilinear neighbour
struct LkLevels {
    const uint8_t *I[kMaxLevels];
    const uint8_t *J[kMaxLevels];
    const short2 *dI[kMaxLevels];
    int W[kMaxLevels], H[kMaxLevels], P[kMaxLevels];
    int levels;                                // highest level index
};

// what the pyramid build leaves (tight level images, level 0 the caller's) -> the padded copies: one launch over all levels
struct LkPadJob {
    const uint8_t *I[kMaxLevels], *J[kMaxLevels];
    const short2 *dI[kMaxLevels];
    uint8_t *Ip[kMaxLevels], *Jp[kMaxLevels];  // padded buffers' ORIGINS (top-left of the border)
    short2 *dIp[kMaxLevels];
    int W[kMaxLevels], H[kMaxLevels];
};
__global__ __launch_bounds__(kBlock) void lk_pad_levels_kernel(LkPadJob job)
{
    const int l = blockIdx.z;
    const int W = job.W[l], H = job.H[l], P = W + 2 * kLkBorder, Hp = H + 2 * kLkBorder;
    const int xp = blockIdx.x * 32 + (threadIdx.x & 31), yp = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (xp >= P || yp >= Hp) return;
    const int x = xp - kLkBorder, y = yp - kLkBorder;
    const bool inside = x >= 0 && x < W && y >= 0 && y < H;
    // BORDER_REFLECT_101, one reflection then clamped (levels narrower than the border: the far part of the border is never read by
    // a window that passes the tracker's own test)
    int xr = x < 0 ? -x : x; xr = xr >= W ? 2 * (W - 1) - xr : xr; xr = xr < 0 ? 0 : (xr > W - 1 ? W - 1 : xr);
    int yr = y < 0 ? -y : y; yr = yr >= H ? 2 * (H - 1) - yr : yr; yr = yr < 0 ? 0 : (yr > H - 1 ? H - 1 : yr);
    job.Ip[l][yp * P + xp] = job.I[l][yr * W + xr];
    job.Jp[l][yp * P + xp] = job.J[l][yr * W + xr];
    job.dIp[l][yp * P + xp] = inside ? job.dI[l][y * W + x] : short2{0, 0};
}

// The same pyramid -- every level of both images, the derivative images of the previous one, the border-extended copies -- in ONE
// launch (round 5; the per-level kernels above: four dependent launches + the padding launch, 27 us of a VGA frame's ~210, nearly all
// of it launch latency).  A workgroup owns an 8 x 8 tile of the top level and the tiles under it (16 x 16, 32 x 32, 64 x 64 at level
// 0) and computes the whole chain for them in LDS: level l's patch carries the halo the levels above need, h_top = 1 (the Scharr
// neighbours), h_l = 2 h_(l+1) + 2 (pyrDown's five taps) -- 22 pixels at level 0 of a four-level pyramid, 108 x 108 bytes.  A patch
// holds IN-IMAGE pixels only; every read reflects its coordinate first (BORDER_REFLECT_101, as the per-level kernels do on the whole
// image), and a reflected coordinate lies at most two pixels inside the border, inside the patch of the tile that asks (h_l >= 2 below
// the top level).  Integer arithmetic throughout: bit-identical to the per-level path (tests/test_features.py compares the tracker's
// outputs of both).  The owner of an in-image pixel also writes the border positions that reflect to it (levels at least 33 pixels
// wide and high: one reflection, no clamping; smaller pyramids take the per-level path).
#ifndef MQS_LK_PYRAMID_THREADS
#define MQS_LK_PYRAMID_THREADS 1024
#endif
#ifndef MQS_LK_PYRAMID_TOP_TILE
#define MQS_LK_PYRAMID_TOP_TILE 8
#endif
constexpr int kPyrTopTile = MQS_LK_PYRAMID_TOP_TILE;       // a workgroup's tile of the top level
constexpr int kPyrThreads = MQS_LK_PYRAMID_THREADS;       // sixteen wavefronts on a tile's chain: with four the launch took as long as the five it replaces (28 us)
template <int LTOP>
struct PyrGeom {
    static constexpr int halo(int l) { return l >= LTOP ? 1 : 2 * halo(l + 1) + 2; }
    static constexpr int tile(int l) { return kPyrTopTile << (LTOP - l); }
    static constexpr int side(int l) { return tile(l) + 2 * halo(l); }
    static constexpr int offset(int l) { return l == 0 ? 0 : offset(l - 1) + (side(l - 1) * side(l - 1) + 15) / 16 * 16; }
    static constexpr int total = offset(LTOP) + side(LTOP) * side(LTOP);
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int LTOP, int L>
__device__ __forceinline__ void pyr_down_in_lds(const LkPadJob &job, uint8_t *sP, int tx, int ty, int tid)
{
    using G = PyrGeom<LTOP>;
    constexpr int S = G::side(L), Hh = G::halo(L), T = G::tile(L), Sp = G::side(L - 1), Hp = G::halo(L - 1), Tp = G::tile(L - 1);
    const int W = job.W[L], H = job.H[L], Wp = job.W[L - 1], Hpv = job.H[L - 1];
    const int x0 = tx * T - Hh, y0 = ty * T - Hh, xp0 = tx * Tp - Hp, yp0 = ty * Tp - Hp;
    constexpr int kOffSrc = G::offset(L - 1), kOffDst = G::offset(L);
    const uint8_t *src = sP + kOffSrc;
    uint8_t *dst = sP + kOffDst;
    const int k5[5] = {1, 4, 6, 4, 1};
    for (int e = tid; e < S * S; e += kPyrThreads) {
        const int py = e / S, px = e - py * S, cx = x0 + px, cy = y0 + py;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) continue;             // never read
        int acc = 0, xi[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) xi[i] = clampi(reflect101(2 * cx + i - 2, Wp) - xp0, 0, Sp - 1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint8_t *row = src + clampi(reflect101(2 * cy + j - 2, Hpv) - yp0, 0, Sp - 1) * Sp;
            int r = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) r += k5[i] * (int)row[xi[i]];
            acc += k5[j] * r;
        }
        dst[e] = (uint8_t)((acc + 128) >> 8);
    }
}

template <int LTOP, int L>
__device__ __forceinline__ void pyr_store_level(const LkPadJob &job, const uint8_t *sP, int tx, int ty, int tid, bool prev)
{
    using G = PyrGeom<LTOP>;
    constexpr int S = G::side(L), Hh = G::halo(L), T = G::tile(L);
    const int W = job.W[L], H = job.H[L], P = W + 2 * kLkBorder;
    const int x0 = tx * T - Hh, y0 = ty * T - Hh;
    constexpr int kOff = G::offset(L);
    const uint8_t *lev = sP + kOff;
    uint8_t *out = prev ? job.Ip[L] : job.Jp[L];
    short2 *dout = job.dIp[L];
    for (int e = tid; e < T * T; e += kPyrThreads) {
        const int ly = e / T, lx = e - ly * T, x = tx * T + lx, y = ty * T + ly;
        if (x >= W || y >= H) continue;
        const uint8_t v = lev[(ly + Hh) * S + lx + Hh];
        short2 d = make_short2(0, 0);
        if (prev) {
            const int xm = reflect101(x - 1, W) - x0, xq = reflect101(x + 1, W) - x0, xc = x - x0;
            const uint8_t *rm = lev + (reflect101(y - 1, H) - y0) * S, *rc = lev + (y - y0) * S, *rp = lev + (reflect101(y + 1, H) - y0) * S;
            const int a = rm[xm], b = rm[xc], c = rm[xq], ee = rc[xm], f = rc[xq], g = rp[xm], h = rp[xc], k = rp[xq];
            d = make_short2((short)(3 * (c - a) + 10 * (f - ee) + 3 * (k - g)), (short)(3 * (g - a) + 10 * (h - b) + 3 * (k - c)));
        }
        // the padded positions this pixel fills: its own, and the border's that reflect to it
        int xs[3], ys[3], nx = 1, ny = 1;
        xs[0] = x + kLkBorder; ys[0] = y + kLkBorder;
        if (x >= 1 && x <= kLkBorder) xs[nx++] = kLkBorder - x;
        if (x >= W - 1 - kLkBorder && x <= W - 2) xs[nx++] = kLkBorder + 2 * (W - 1) - x;
        if (y >= 1 && y <= kLkBorder) ys[ny++] = kLkBorder - y;
        if (y >= H - 1 - kLkBorder && y <= H - 2) ys[ny++] = kLkBorder + 2 * (H - 1) - y;
        for (int yi = 0; yi < ny; ++yi)
            for (int xi2 = 0; xi2 < nx; ++xi2) {
                const int pos = ys[yi] * P + xs[xi2];
                out[pos] = v;
                if (prev) dout[pos] = (yi == 0 && xi2 == 0) ? d : make_short2(0, 0);
            }
    }
}

template <int LTOP>
__global__ __launch_bounds__(kPyrThreads) void lk_pyramid_kernel(LkPadJob job)
{
    using G = PyrGeom<LTOP>;
    __shared__ __attribute__((aligned(16))) uint8_t sP[G::total];
    const int tid = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y;
    const bool prev = blockIdx.z == 0;
    {
        constexpr int S = G::side(0), Hh = G::halo(0), T = G::tile(0), kLoads = (S * S + kPyrThreads - 1) / kPyrThreads;
        const uint8_t *src = prev ? job.I[0] : job.J[0];
        const int W = job.W[0], H = job.H[0], x0 = tx * T - Hh, y0 = ty * T - Hh;
        // every load of the patch issued before the first is waited for (unconditional, the index clamped: written as a loop of
        // load + store the compiler waits for each byte's round trip in turn -- twelve of them were 12 us of the launch's 18)
        uint8_t v[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            int e = tid + u * kPyrThreads;
            e = e < S * S ? e : S * S - 1;
            const int py = e / S, px = e - py * S;
            v[u] = src[clampi(reflect101(y0 + py, H), 0, H - 1) * W + clampi(reflect101(x0 + px, W), 0, W - 1)];
        }
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int e = tid + u * kPyrThreads;
            if (e < S * S) sP[e] = v[u];
        }
    }
    __syncthreads();
    if constexpr (LTOP >= 1) { pyr_down_in_lds<LTOP, 1>(job, sP, tx, ty, tid); __syncthreads(); }
    if constexpr (LTOP >= 2) { pyr_down_in_lds<LTOP, 2>(job, sP, tx, ty, tid); __syncthreads(); }
    if constexpr (LTOP >= 3) { pyr_down_in_lds<LTOP, 3>(job, sP, tx, ty, tid); __syncthreads(); }
    pyr_store_level<LTOP, 0>(job, sP, tx, ty, tid, prev);
    if constexpr (LTOP >= 1) pyr_store_level<LTOP, 1>(job, sP, tx, ty, tid, prev);
    if constexpr (LTOP >= 2) pyr_store_level<LTOP, 2>(job, sP, tx, ty, tid, prev);
    if constexpr (LTOP >= 3) pyr_store_level<LTOP, 3>(job, sP, tx, ty, tid, prev);
    // (measured and not kept: a level's stores issued before the next level is computed, + 0.8 us; a reflection-free path for the
    // tiles off the rim, + 0.3 us; 4 x 4 top tiles, + 2.3 us; 512 / 256 threads, + 3 / + 11 us.  An empty launch of this shape: 4.3 us)
}

// wave sums with the butterfly's pairs, without its ds_bpermute round trips (a quarter of a Lucas-Kanade iteration): wave_reduce.h
__device__ __forceinline__ void wave_sum2_d(double x, double y, double &sx, double &sy) { mqs::wave::sum2(x, y, sx, sy); }
__device__ __forceinline__ double wave_sum_d(double v) { return mqs::wave::sum1(v); }

__device__ __forceinline__ float bilinear_u8(const uint8_t *__restrict__ a, int W, int x, int y, float w00, float w01, float w10, float w11)
{
    const uint8_t *p = a + y * W + x;
    return (((float)p[0] * w00 + (float)p[1] * w01) + (float)p[W] * w10) + (float)p[W + 1] * w11;
}

// NW wavefronts per feature (round 5: four): a feature's track is a serial chain of ~80 iterations, each a window sum and a 2 x 2 solve; with
// one wavefront the chain's length IS the launch's (300 features on 256 CUs: one wave per CU, 73 us); four wavefronts take a quarter of the
// window each and meet in LDS once per sum (parity-alternating slots: one workgroup barrier per sum).  Two wavefronts: 40.8 us, four: 38.5,
// eight: 41.4 (with the loads of an iteration batched; 55.8 / 47.1 / 47.4 before) -- the chain is the iteration's own dependent instructions
// (weights, bilinear taps, the wave sum, the 2 x 2 solve).
#ifndef MQS_LK_WAVES
#define MQS_LK_WAVES 4
#endif
constexpr int kLkWaves = MQS_LK_WAVES;
// wave_reduce.h's sum2 without its last step: the sum of x in every lane 0..31, the sum of y in every lane 32..63 (same pairs, same bits)
__device__ __forceinline__ double lk_sum2_halves(double x, double y)
{
    mqs::wave::swap32(x, y);
    double t = x + y;
    double a = t, b = t;
    mqs::wave::swap16(a, b);
    t = a + b;
    t += mqs::wave::xor_lane<8>(t);
    t += mqs::wave::xor_lane<4>(t);
    t += mqs::wave::xor_lane<2>(t);
    t += mqs::wave::xor_lane<1>(t);
    return t;
}

template <int NW>
__device__ __forceinline__ void lk_block_sum2(double x, double y, double &sx, double &sy, double *sRed, int &slot, int wave, int lane)
{
    if (NW == 1) { mqs::wave::sum2(x, y, sx, sy); return; }
    {
        // lanes 0 and 32 hold the wavefront's two totals after the butterfly: they store them themselves (no v_readlane round trip)
        const double t = lk_sum2_halves(x, y);
        double *r = sRed + (slot & 1) * 2 * NW;
        slot += 1;
        if ((lane & 31) == 0) r[2 * wave + (lane >> 5)] = t;
        __syncthreads();
        sx = 0.0; sy = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { sx += r[2 * w]; sy += r[2 * w + 1]; }
    }
}

template <int NW, int kPix /* window pixels per thread: ceil(ww * wh / (64 NW)) -- two for the loop's 21 x 21 window on four wavefronts */>
__global__ __launch_bounds__(64 * NW) void lk_kernel(LkLevels L, const float *__restrict__ prev_pts, int n, const int32_t *__restrict__ n_dev,
                                                    int ww, int wh, int max_iter, float eps, float min_eig_threshold,
                                                    float *__restrict__ next_pts, uint8_t *__restrict__ status, float *__restrict__ err)
{
    __shared__ double sRed[4 * NW];
    int slot = 0;
    const int k = blockIdx.x, lane = threadIdx.x, wave = threadIdx.x >> 6;
    if (n_dev) n = min(n, *n_dev);                  // the device-resident loop: the number of live tracks is device state
    if (k >= n) return;
    const int npix = ww * wh;
    const float halfx = (float)(ww - 1) * 0.5f, halfy = (float)(wh - 1) * 0.5f;
    const float ppx = prev_pts[2 * k], ppy = prev_pts[2 * k + 1];
    const float kFltScale = 1.0f / (float)(1 << 20);
    float nx = 0.0f, ny = 0.0f;
    bool ok = true;
    float errv = 0.0f;
#ifdef MQS_LK_COUNT_ITERS                                  // (A/B builds: err[] returns the feature's iteration count over all levels)
    int iters_total = 0;
#endif
    for (int level = L.levels; level >= 0; --level) {
        const int W = L.W[level], H = L.H[level];
        const float sc = 1.0f / (float)(1 << level);
        float px = ppx * sc, py = ppy * sc;
        if (level == L.levels) { nx = px; ny = py; } else { nx = nx * 2.0f; ny = ny * 2.0f; }
        px -= halfx; py -= halfy;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= W || ipy < -wh || ipy >= H) {     // lkpyramid.cpp's test: the window may leave the image by its own size
            if (level == 0) { ok = false; errv = 0.0f; }
            continue;
        }
        const int P = L.P[level];
        float Iw[kPix], Ixw[kPix], Iyw[kPix];
        // Every window load below is issued UNCONDITIONALLY (a lane's pixel index beyond the window is clamped to the window's last pixel and
        // its weight is zero): a load inside a divergent branch makes the wait-count bookkeeping serialise the seven pixel groups of an
        // iteration -- seven dependent L1 round trips where one suffices (round 5; found on the resident adjuster).  The pixel's offset inside
        // the padded level (two integer divisions by the runtime window width) is computed once per level, not once per iteration.
        int woff[kPix];
        bool win[kPix];
#pragma unroll
        for (int t = 0; t < kPix; ++t) {
            const int p = lane + 64 * NW * t;
            win[t] = p < npix;
            const int pc = win[t] ? p : npix - 1;
            woff[t] = (pc / ww) * P + pc % ww;
        }
        double a11 = 0.0, a12 = 0.0, a22 = 0.0;
        {
            const float a = px - (float)ipx, b = py - (float)ipy;
            const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
            const uint8_t *I = L.I[level] + ipy * P + ipx;
            const short2 *dI = L.dI[level] + ipy * P + ipx;
            // (as in the iterations below: every load of the set-up in flight before the first is used)
            int i00[kPix], i01[kPix], i10[kPix], i11[kPix];
            int e00[kPix], e01[kPix], e10[kPix], e11[kPix];        // the four derivative pairs as their 32-bit words
#pragma unroll
            for (int t = 0; t < kPix; ++t) {
                const uint8_t *q = I + woff[t];
                i00[t] = q[0]; i01[t] = q[1]; i10[t] = q[P]; i11[t] = q[P + 1];
                const int *dw = reinterpret_cast<const int *>(dI + woff[t]);
                e00[t] = dw[0]; e01[t] = dw[1]; e10[t] = dw[P]; e11[t] = dw[P + 1];
            }
#pragma unroll
            for (int t = 0; t < kPix; ++t)
                asm volatile("" : "+v"(i00[t]), "+v"(i01[t]), "+v"(i10[t]), "+v"(i11[t]), "+v"(e00[t]), "+v"(e01[t]), "+v"(e10[t]), "+v"(e11[t]));
#pragma unroll
            for (int t = 0; t < kPix; ++t) {
                const float iv = ((((float)i00[t] * w00 + (float)i01[t] * w01) + (float)i10[t] * w10) + (float)i11[t] * w11) * 32.0f;
                const short2 d00 = make_short2((short)(e00[t] & 0xffff), (short)(e00[t] >> 16)), d01 = make_short2((short)(e01[t] & 0xffff), (short)(e01[t] >> 16));
                const short2 d10 = make_short2((short)(e10[t] & 0xffff), (short)(e10[t] >> 16)), d11 = make_short2((short)(e11[t] & 0xffff), (short)(e11[t] >> 16));
                const float ix = (((float)d00.x * w00 + (float)d01.x * w01) + (float)d10.x * w10) + (float)d11.x * w11;
                const float iy = (((float)d00.y * w00 + (float)d01.y * w01) + (float)d10.y * w10) + (float)d11.y * w11;
                Iw[t] = win[t] ? iv : 0.0f; Ixw[t] = win[t] ? ix : 0.0f; Iyw[t] = win[t] ? iy : 0.0f;
                a11 += (double)Ixw[t] * (double)Ixw[t];
                a12 += (double)Ixw[t] * (double)Iyw[t];
                a22 += (double)Iyw[t] * (double)Iyw[t];
            }
        }
        double s11, s12, s22, unused;
        lk_block_sum2<NW>(a11, a12, s11, s12, sRed, slot, wave, lane & 63);
        lk_block_sum2<NW>(a22, 0.0, s22, unused, sRed, slot, wave, lane & 63);
        const float A11 = (float)s11 * kFltScale, A12 = (float)s12 * kFltScale, A22 = (float)s22 * kFltScale;
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.0f * A12 * A12)) / (float)(2 * ww * wh);
        if (min_eig < min_eig_threshold || D < 1.1920929e-07f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.0f / D;
        nx -= halfx; ny -= halfy;
        float pdx = 0.0f, pdy = 0.0f;
        const uint8_t *J = L.J[level];
        for (int j = 0; j < max_iter; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                if (level == 0) ok = false;
                break;
            }
            const float a = nx - (float)inx, b = ny - (float)iny;
            const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
            const uint8_t *Jw = J + iny * P + inx;
            double b1 = 0.0, b2 = 0.0;
            // all of the iteration's window bytes requested before the first is used: the compiler otherwise sinks each pixel's loads into
            // the branch its `win` select becomes and waits for them there, one L1 round trip per pixel of the thread (ISA listing, round 5)
            int j00[kPix], j01[kPix], j10[kPix], j11[kPix];
#pragma unroll
            for (int t = 0; t < kPix; ++t) {
                const uint8_t *q = Jw + woff[t];
                j00[t] = q[0]; j01[t] = q[1]; j10[t] = q[P]; j11[t] = q[P + 1];
            }
#pragma unroll
            for (int t = 0; t < kPix; ++t) asm volatile("" : "+v"(j00[t]), "+v"(j01[t]), "+v"(j10[t]), "+v"(j11[t]));
#pragma unroll
            for (int t = 0; t < kPix; ++t) {
                const float jv = (((float)j00[t] * w00 + (float)j01[t] * w01) + (float)j10[t] * w10) + (float)j11[t] * w11;
                const float diff = win[t] ? jv * 32.0f - Iw[t] : 0.0f;
                b1 += (double)diff * (double)Ixw[t];
                b2 += (double)diff * (double)Iyw[t];
            }
            double sb1, sb2;
            lk_block_sum2<NW>(b1, b2, sb1, sb2, sRed, slot, wave, lane & 63);
            const float B1 = (float)sb1 * kFltScale, B2 = (float)sb2 * kFltScale;
            const float dx = (A12 * B2 - A22 * B1) * D, dy = (A12 * B1 - A11 * B2) * D;
            nx += dx; ny += dy;
#ifdef MQS_LK_COUNT_ITERS
            iters_total += 1;
#endif
            if (dx * dx + dy * dy <= eps * eps) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        nx += halfx; ny += halfy;
        if (ok && level == 0) {
            const float qx = nx - halfx, qy = ny - halfy;
            const int inx = (int)floorf(qx), iny = (int)floorf(qy);
            if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                ok = false;
            } else {
                const float a = qx - (float)inx, b = qy - (float)iny;
                const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
                const uint8_t *Jw = J + iny * P + inx;
                double e = 0.0;
#pragma unroll
                for (int t = 0; t < kPix; ++t) {
                    const uint8_t *q = Jw + woff[t];
                    const float jv = (((float)q[0] * w00 + (float)q[1] * w01) + (float)q[P] * w10) + (float)q[P + 1] * w11;
                    if (win[t]) e += fabs((double)(jv * 32.0f - Iw[t]));
                }
                double se, unused2;
                lk_block_sum2<NW>(e, 0.0, se, unused2, sRed, slot, wave, lane & 63);
                errv = (float)(se / (32.0 * (double)ww * (double)wh));
            }
        }
    }
    if (lane == 0) {
        next_pts[2 * k] = nx;
        next_pts[2 * k + 1] = ny;
        status[k] = ok ? 1 : 0;
        err[k] = ok ? errv : 0.0f;
#ifdef MQS_LK_COUNT_ITERS
        err[k] = (float)iters_total;
#endif
    }
}

// ---------------------------------------------------------------------------------------------------
// FAST-9/16 (FastFeatureDetector, slam.py:34, 62): integer, exact against the oracle
// ---------------------------------------------------------------------------------------------------
__constant__ int kFastDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int kFastDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// score[y][x] = 0 (no corner / border) or max over the 16 arcs of 9 contiguous ring pixels of the arc's smallest
// sign-consistent |centre - ring| difference, minus 1 (OpenCV 2.4 cornerScore<16>), when that exceeds the threshold
__global__ __launch_bounds__(kBlock) void fast_score_kernel(const uint8_t *__restrict__ img, int W, int H, int threshold,
                                                           int *__restrict__ score)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    int out = 0;
    if (x >= 3 && y >= 3 && x < W - 3 && y < H - 3) {
        const int c = img[y * W + x];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = c - (int)img[(y + kFastDy[k]) * W + x + kFastDx[k]];
        int best = -1000000;
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            int lo = 1000000, hi = 1000000;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int v = d[(s0 + k) & 15];
                lo = min(lo, v);
                hi = min(hi, -v);
            }
            best = max(best, max(lo, hi));
        }
        if (best > threshold) out = best - 1;
    }
    score[y * W + x] = out;
}

__device__ __forceinline__ bool fast_keep(const int *__restrict__ score, int W, int H, int x, int y, int nonmax)
{
    const int s = score[y * W + x];
    if (s <= 0) return false;
    if (!nonmax) return true;
    for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox) {
            if (!oy && !ox) continue;
            const int yy = y + oy, xx = x + ox;
            const int v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? score[yy * W + xx] : 0;
            if (!(s > v)) return false;
        }
    return true;
}

// one workgroup per image row: number of kept corners in the row
__global__ __launch_bounds__(kBlock) void fast_row_count_kernel(const int *__restrict__ score, int W, int H, int nonmax,
                                                               int *__restrict__ row_count)
{
    __shared__ int s[kBlock];
    const int y = blockIdx.x;
    int c = 0;
    for (int x = threadIdx.x; x < W; x += kBlock) c += fast_keep(score, W, H, x, y, nonmax) ? 1 : 0;
    s[threadIdx.x] = c;
    __syncthreads();
    for (int h = kBlock / 2; h >= 1; h >>= 1) {
        if (threadIdx.x < h) s[threadIdx.x] += s[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_count[y] = s[0];
}

// exclusive scan of the row counts (one workgroup); total -> out_n
__global__ __launch_bounds__(kBlock) void fast_row_scan_kernel(const int *__restrict__ row_count, int H, int *__restrict__ row_offset,
                                                              int *__restrict__ out_n)
{
    __shared__ int s[kBlock];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int y0 = 0; y0 < H; y0 += kBlock) {
        const int y = y0 + threadIdx.x;
        const int v = y < H ? row_count[y] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int h = 1; h < kBlock; h <<= 1) {                 // inclusive Hillis-Steele scan
            const int t = threadIdx.x >= h ? s[threadIdx.x - h] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (y < H) row_offset[y] = base + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) base += s[kBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_n[0] = base;
}

// one workgroup per row: corners written in x order at the row's offset (row-major scan order overall)
__global__ __launch_bounds__(kBlock) void fast_emit_kernel(const int *__restrict__ score, int W, int H, int nonmax,
                                                          const int *__restrict__ row_offset, int capacity,
                                                          float *__restrict__ out_xy, int *__restrict__ out_score)
{
    __shared__ int s[kBlock];
    __shared__ int base;
    const int y = blockIdx.x;
    if (threadIdx.x == 0) base = row_offset[y];
    __syncthreads();
    for (int x0 = 0; x0 < W; x0 += kBlock) {
        const int x = x0 + threadIdx.x;
        const int f = (x < W && fast_keep(score, W, H, x, y, nonmax)) ? 1 : 0;
        s[threadIdx.x] = f;
        __syncthreads();
        for (int h = 1; h < kBlock; h <<= 1) {
            const int t = threadIdx.x >= h ? s[threadIdx.x - h] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const int pos = base + s[threadIdx.x] - f;
        if (f && pos < capacity) {
            out_xy[2 * pos] = (float)x;
            out_xy[2 * pos + 1] = (float)y;
            if (out_score) out_score[pos] = score[y * W + x];
        }
        __syncthreads();
        if (threadIdx.x == 0) base += s[kBlock - 1];
        __syncthreads();
    }
}

dim3 grid2d(int W, int H) { return dim3((unsigned)((W + 31) / 32), (unsigned)((H + 7) / 8)); }

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

extern "C" {

int64_t mqs_gftt_workspace_bytes(int W, int H)
{
    if (W < 1 || H < 1) return 0;
    const size_t npx = (size_t)W * H;
    const size_t ntiles = (size_t)((W + 31) / 32) * ((H + 7) / 8);
    // response, the tiles' candidate segments (256 slots each), the gathered keys, slots of the maximum + tile counts,
    // selection grid (worst case: cell = 1)
    return (int64_t)(align_up(npx * 4) + align_up(ntiles * kBlock * 8) + align_up(npx * 8) + align_up((kMaxSlots + ntiles) * 4) +
                     align_up(npx * 16));
}

int mqs_good_features_to_track_dev(const uint8_t *img, int W, int H, int max_corners, double quality_level,
                                   double min_distance, const uint8_t *mask, float *out_xy, int out_capacity, int32_t *out_n,
                                   void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(img && out_xy && out_n && workspace, "pointers must not be null");
    MQS_ARG_CHECK(W >= 3 && H >= 3 && W < 65536 && H < 65536, "3 <= W, H < 65536");
    MQS_ARG_CHECK(max_corners >= 0 && out_capacity >= 1 && quality_level > 0.0 && min_distance >= 0.0, "parameter ranges");
    MQS_ARG_CHECK(workspace_bytes >= mqs_gftt_workspace_bytes(W, H), "workspace too small (mqs_gftt_workspace_bytes)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t npx = (size_t)W * H;
    const size_t ntiles = (size_t)((W + 31) / 32) * ((H + 7) / 8);
    char *w = static_cast<char *>(workspace);
    float *eig = reinterpret_cast<float *>(w); w += align_up(npx * 4);
    unsigned long long *keys_seg = reinterpret_cast<unsigned long long *>(w); w += align_up(ntiles * kBlock * 8);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(w); w += align_up(npx * 8);
    unsigned int *maxkey = reinterpret_cast<unsigned int *>(w);          // [kMaxSlots] order keys of the largest response
    unsigned int *wg_count = maxkey + kMaxSlots; w += align_up((kMaxSlots + ntiles) * 4);
    unsigned int *grid = reinterpret_cast<unsigned int *>(w);

    // four launches: clear, response (+ maximum), candidates, gather + sort + selection
    MQS_HIP_CHECK(hipMemsetAsync(maxkey, 0, kMaxSlots * 4, stream));
    hipLaunchKernelGGL(min_eig_kernel, grid2d(W, H), dim3(kBlock), 0, stream, img, W, H, eig, mask, maxkey);
    hipLaunchKernelGGL(candidates_kernel, grid2d(W, H), dim3(kBlock), 0, stream, eig, W, H, maxkey, (float)quality_level, mask,
                       keys_seg, wg_count);
    const int cell = min_distance >= 1.0 ? (int)rint(min_distance) : 1;
    const size_t cells = (size_t)((W + cell - 1) / cell) * ((H + cell - 1) / cell);
    const size_t sort_lds = (size_t)kSortChunk * sizeof(unsigned long long);
    const size_t grid_lds = cells * 16;
    const bool in_lds = min_distance < 1.0 || grid_lds <= sort_lds;      // the grid takes the sort buffer's place
    const size_t lds = sort_lds;
    if (in_lds) {
        static mqs_lds_opt_in opt;                           // per device
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(sort_select_kernel<true>), (int)sort_lds));
        hipLaunchKernelGGL(sort_select_kernel<true>, dim3(1), dim3(kSelThreads), lds, stream, keys_seg, keys, wg_count, (unsigned int)ntiles, W,
                           H, (float)min_distance, max_corners, out_capacity, grid, out_xy, out_n);
    } else {
        static mqs_lds_opt_in opt;                           // per device
        MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(sort_select_kernel<false>), (int)sort_lds));
        hipLaunchKernelGGL(sort_select_kernel<false>, dim3(1), dim3(kSelThreads), sort_lds, stream, keys_seg, keys, wg_count,
                           (unsigned int)ntiles, W, H, (float)min_distance, max_corners, out_capacity, grid, out_xy, out_n);
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int64_t mqs_lk_workspace_bytes(int W, int H, int max_level)
{
    if (W < 1 || H < 1 || max_level < 0 || max_level >= kMaxLevels) return 0;
    size_t total = 0;
    int w = W, h = H;
    for (int l = 0; l <= max_level; ++l) {
        // prev / next level images (level 0 is the caller's) and the derivative image of prev
        if (l > 0) total += 2 * align_up((size_t)w * h);
        total += align_up((size_t)w * h * 4);
        // ... and their border-extended copies (what the tracker reads)
        const size_t padded = (size_t)(w + 2 * kLkBorder) * (h + 2 * kLkBorder);
        total += 2 * align_up(padded) + align_up(padded * 4);
        if (w <= 2 || h <= 2) break;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    return (int64_t)total;
}

int mqs_calc_optical_flow_pyr_lk_dev(const uint8_t *prev_img, const uint8_t *next_img, int W, int H, const float *prev_pts,
                                     int n, int win_w, int win_h, int max_level, int max_iter, double eps,
                                     double min_eig_threshold, float *next_pts, uint8_t *status, float *err, void *workspace,
                                     int64_t workspace_bytes, void *stream_)
{
    return mqs_lk_launch(prev_img, next_img, W, H, prev_pts, n, nullptr, win_w, win_h, max_level, max_iter, eps, min_eig_threshold,
                         next_pts, status, err, workspace, workspace_bytes, static_cast<hipStream_t>(stream_), 3);
}

}  // extern "C"

// n = capacity of the point arrays (one wavefront each is launched); n_dev (device, may be null): the live count
// phases: bit 0 -- the pyramid of the image pair (levels, derivatives, border-extended copies) into the workspace; bit 1 -- the tracker on
// the pyramid the workspace holds.  3 = both (one call); the device loop builds the NEXT pair's pyramid on a side stream under the current
// frame's pose kernels (mqs_slam_prepare_next) and then runs the tracker alone.
int mqs_lk_launch(const uint8_t *prev_img, const uint8_t *next_img, int W, int H, const float *prev_pts, int n, const int32_t *n_dev,
                  int win_w, int win_h, int max_level, int max_iter, double eps, double min_eig_threshold, float *next_pts,
                  uint8_t *status, float *err, void *workspace, int64_t workspace_bytes, hipStream_t stream, int phases)
{
    MQS_ARG_CHECK(phases >= 1 && phases <= 3, "phases: 1 pyramid, 2 tracker, 3 both");
    if (!(phases & 2)) n = 0;
    MQS_ARG_CHECK(prev_img && next_img && workspace, "pointers must not be null");
    MQS_ARG_CHECK(n >= 0 && (n == 0 || (prev_pts && next_pts && status && err)), "point arrays must not be null");
    MQS_ARG_CHECK(W >= 3 && H >= 3 && max_level >= 0 && max_level < kMaxLevels, "W, H >= 3; 0 <= max_level < 8");
    MQS_ARG_CHECK(win_w >= 3 && win_h >= 3 && win_w * win_h <= 64 * kMaxWinPixelsPerLane, "3 <= window, at most 1024 pixels");
    MQS_ARG_CHECK(max_iter >= 1 && eps >= 0.0, "max_iter >= 1, eps >= 0");
    MQS_ARG_CHECK(workspace_bytes >= mqs_lk_workspace_bytes(W, H, max_level), "workspace too small (mqs_lk_workspace_bytes)");
    MQS_ARG_CHECK(win_w + 1 <= kLkBorder && win_h + 1 <= kLkBorder, "window larger than the border the pyramid copies carry (31 x 31)");
    LkLevels L;
    LkPadJob job;
    char *wsp = static_cast<char *>(workspace);
    int w = W, h = H;
    L.levels = 0;
    job.I[0] = prev_img; job.J[0] = next_img;
    uint8_t *down_i[kMaxLevels] = {}, *down_j[kMaxLevels] = {};
    short2 *deriv[kMaxLevels] = {};
    for (int l = 0; l <= max_level; ++l) {
        L.W[l] = w; L.H[l] = h; L.P[l] = w + 2 * kLkBorder;
        job.W[l] = w; job.H[l] = h;
        short2 *d = reinterpret_cast<short2 *>(wsp); wsp += align_up((size_t)w * h * 4);
        job.dI[l] = d; deriv[l] = d;
        {
            const size_t padded = (size_t)(w + 2 * kLkBorder) * (h + 2 * kLkBorder), origin = (size_t)kLkBorder * L.P[l] + kLkBorder;
            job.Ip[l] = reinterpret_cast<uint8_t *>(wsp); wsp += align_up(padded);
            job.Jp[l] = reinterpret_cast<uint8_t *>(wsp); wsp += align_up(padded);
            job.dIp[l] = reinterpret_cast<short2 *>(wsp); wsp += align_up(padded * 4);
            L.I[l] = job.Ip[l] + origin; L.J[l] = job.Jp[l] + origin; L.dI[l] = job.dIp[l] + origin;
        }
        L.levels = l;
        const bool last = l == max_level || w <= 2 || h <= 2;
        if (last) break;
        const int wd = (w + 1) / 2, hd = (h + 1) / 2;
        down_i[l] = reinterpret_cast<uint8_t *>(wsp); wsp += align_up((size_t)wd * hd);
        down_j[l] = reinterpret_cast<uint8_t *>(wsp); wsp += align_up((size_t)wd * hd);
        job.I[l + 1] = down_i[l]; job.J[l + 1] = down_j[l];
        w = wd; h = hd;
    }
    // one launch for the whole pyramid where the top level is at least a border wide and high (a VGA frame's four levels: 80 x 60);
    // MQS_LK_PYRAMID_PER_LEVEL=1 keeps the per-level launches (A/B, tests)
    const char *per_level = getenv("MQS_LK_PYRAMID_PER_LEVEL");
    const bool fused = L.levels >= 1 && L.levels <= 3 && L.W[L.levels] > kLkBorder && L.H[L.levels] > kLkBorder && !(per_level && per_level[0] == '1');
    if (!(phases & 1)) {
        // the pyramid is in the workspace already
    } else if (fused) {
        const dim3 g((unsigned)((L.W[L.levels] + kPyrTopTile - 1) / kPyrTopTile), (unsigned)((L.H[L.levels] + kPyrTopTile - 1) / kPyrTopTile), 2);
        if (L.levels == 1) hipLaunchKernelGGL(lk_pyramid_kernel<1>, g, dim3(kPyrThreads), 0, stream, job);
        else if (L.levels == 2) hipLaunchKernelGGL(lk_pyramid_kernel<2>, g, dim3(kPyrThreads), 0, stream, job);
        else hipLaunchKernelGGL(lk_pyramid_kernel<3>, g, dim3(kPyrThreads), 0, stream, job);
    } else {
        for (int l = 0; l <= L.levels; ++l) {
            // derivatives of this level and, beside them, both images' next level: one launch
            const bool last = l == L.levels;
            dim3 g = grid2d(L.W[l], L.H[l]);
            g.z = last ? 1 : 3;
            hipLaunchKernelGGL(pyr_level_kernel, g, dim3(kBlock), 0, stream, job.I[l], job.J[l], L.W[l], L.H[l], deriv[l], last ? nullptr : down_i[l],
                               last ? nullptr : down_j[l], last ? 0 : L.W[l + 1], last ? 0 : L.H[l + 1]);
        }
        dim3 g = grid2d(W + 2 * kLkBorder, H + 2 * kLkBorder);            // sized for level 0; the other levels use its first part
        g.z = (unsigned)(L.levels + 1);
        hipLaunchKernelGGL(lk_pad_levels_kernel, g, dim3(kBlock), 0, stream, job);
    }
    if (n > 0) {
        // (a thread's pixels beyond the window carry weight zero: instantiated per count so that the loop's window does not pay for the largest)
        const int per_thread = (win_w * win_h + 64 * kLkWaves - 1) / (64 * kLkWaves);
#define MQS_LK_LAUNCH(KP)                                                                                                             \
        hipLaunchKernelGGL((lk_kernel<kLkWaves, KP>), dim3(n), dim3(64 * kLkWaves), 0, stream, L, prev_pts, n, n_dev, win_w, win_h, max_iter,   \
                           (float)eps, (float)min_eig_threshold, next_pts, status, err)
        constexpr int kPixMax = (kMaxWinPixelsPerLane + kLkWaves - 1) / kLkWaves;
        constexpr int kPix4 = 4 < kPixMax ? 4 : kPixMax;
        if (per_thread <= 1) MQS_LK_LAUNCH(1);
        else if (per_thread <= 2) MQS_LK_LAUNCH(2);
        else if (per_thread <= 4) MQS_LK_LAUNCH(kPix4);
        else MQS_LK_LAUNCH(kPixMax);
#undef MQS_LK_LAUNCH
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

extern "C" {

int64_t mqs_fast_workspace_bytes(int W, int H)
{
    if (W < 1 || H < 1) return 0;
    return (int64_t)(align_up((size_t)W * H * 4) + 2 * align_up((size_t)H * 4));
}

int mqs_fast_detect_dev(const uint8_t *img, int W, int H, int threshold, int nonmax, float *out_xy, int32_t *out_score,
                        int out_capacity, int32_t *out_n, void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(img && out_xy && out_n && workspace, "pointers must not be null");
    MQS_ARG_CHECK(W >= 1 && H >= 1 && W < 65536 && H < 65536 && threshold >= 0 && out_capacity >= 1, "parameter ranges");
    MQS_ARG_CHECK(workspace_bytes >= mqs_fast_workspace_bytes(W, H), "workspace too small (mqs_fast_workspace_bytes)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *w = static_cast<char *>(workspace);
    int *score = reinterpret_cast<int *>(w); w += align_up((size_t)W * H * 4);
    int *row_count = reinterpret_cast<int *>(w); w += align_up((size_t)H * 4);
    int *row_offset = reinterpret_cast<int *>(w);
    hipLaunchKernelGGL(fast_score_kernel, grid2d(W, H), dim3(kBlock), 0, stream, img, W, H, threshold, score);
    hipLaunchKernelGGL(fast_row_count_kernel, dim3(H), dim3(kBlock), 0, stream, score, W, H, nonmax, row_count);
    hipLaunchKernelGGL(fast_row_scan_kernel, dim3(1), dim3(kBlock), 0, stream, row_count, H, row_offset, out_n);
    hipLaunchKernelGGL(fast_emit_kernel, dim3(H), dim3(kBlock), 0, stream, score, W, H, nonmax, row_offset, out_capacity, out_xy,
                       out_score);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// ---- host-pointer wrappers ----
int mqs_fast_detect(mqs_ctx *ctx, const uint8_t *img, int W, int H, int threshold, int nonmax, float *out_xy,
                    int32_t *out_score, int out_capacity, int32_t *out_n)
{
    MQS_ARG_CHECK(ctx && img && out_xy && out_n, "pointers must not be null");
    MQS_ARG_CHECK(W >= 1 && H >= 1 && out_capacity >= 1, "W, H, out_capacity >= 1");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)W * H, wsb = (size_t)mqs_fast_workspace_bytes(W, H);
    const size_t o_img = 0, o_xy = align_up(npx), o_sc = o_xy + align_up((size_t)out_capacity * 8),
                 o_n = o_sc + align_up((size_t)out_capacity * 4), o_ws = o_n + 256, total = o_ws + wsb;
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    hipStream_t s = ctx->stream;
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_img, img, npx, hipMemcpyHostToDevice, s));
    rc = mqs_fast_detect_dev((uint8_t *)(d + o_img), W, H, threshold, nonmax, (float *)(d + o_xy), (int32_t *)(d + o_sc),
                             out_capacity, (int32_t *)(d + o_n), d + o_ws, (int64_t)wsb, s);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(out_n, d + o_n, 4, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    int n = out_n[0] < out_capacity ? out_n[0] : out_capacity;
    if (n > 0) {
        MQS_HIP_CHECK(hipMemcpyAsync(out_xy, d + o_xy, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        if (out_score) MQS_HIP_CHECK(hipMemcpyAsync(out_score, d + o_sc, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        MQS_HIP_CHECK(hipStreamSynchronize(s));
    }
    return MQS_OK;
}

int mqs_good_features_to_track(mqs_ctx *ctx, const uint8_t *img, int W, int H, int max_corners, double quality_level,
                               double min_distance, const uint8_t *mask, float *out_xy, int out_capacity, int32_t *out_n)
{
    MQS_ARG_CHECK(ctx && img && out_xy && out_n, "pointers must not be null");
    MQS_ARG_CHECK(W >= 3 && H >= 3 && out_capacity >= 1, "W, H >= 3, out_capacity >= 1");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)W * H, wsb = (size_t)mqs_gftt_workspace_bytes(W, H);
    const size_t o_img = 0, o_mask = align_up(npx), o_xy = o_mask + align_up(npx), o_n = o_xy + align_up((size_t)out_capacity * 8),
                 o_ws = o_n + 256, total = o_ws + wsb;
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    hipStream_t s = ctx->stream;
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_img, img, npx, hipMemcpyHostToDevice, s));
    if (mask) MQS_HIP_CHECK(hipMemcpyAsync(d + o_mask, mask, npx, hipMemcpyHostToDevice, s));
    rc = mqs_good_features_to_track_dev((uint8_t *)(d + o_img), W, H, max_corners, quality_level, min_distance,
                                        mask ? (uint8_t *)(d + o_mask) : nullptr, (float *)(d + o_xy), out_capacity,
                                        (int32_t *)(d + o_n), d + o_ws, (int64_t)wsb, s);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(out_n, d + o_n, 4, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipMemcpyAsync(out_xy, d + o_xy, (size_t)out_capacity * 8, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

int mqs_calc_optical_flow_pyr_lk(mqs_ctx *ctx, const uint8_t *prev_img, const uint8_t *next_img, int W, int H,
                                 const float *prev_pts, int n, int win_w, int win_h, int max_level, int max_iter, double eps,
                                 double min_eig_threshold, float *next_pts, uint8_t *status, float *err)
{
    MQS_ARG_CHECK(ctx && prev_img && next_img, "pointers must not be null");
    MQS_ARG_CHECK(W >= 3 && H >= 3 && n >= 0, "W, H >= 3, n >= 0");
    if (n == 0) return MQS_OK;
    MQS_ARG_CHECK(prev_pts && next_pts && status && err, "point arrays must not be null");
    MQS_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)W * H, wsb = (size_t)mqs_lk_workspace_bytes(W, H, max_level);
    MQS_ARG_CHECK(wsb > 0, "0 <= max_level < 8");
    const size_t o_i = 0, o_j = align_up(npx), o_p = o_j + align_up(npx), o_q = o_p + align_up((size_t)n * 8),
                 o_s = o_q + align_up((size_t)n * 8), o_e = o_s + align_up((size_t)n), o_ws = o_e + align_up((size_t)n * 4),
                 total = o_ws + wsb;
    int rc = mqs_ctx_reserve(ctx, total);
    if (rc != MQS_OK) return rc;
    char *d = static_cast<char *>(ctx->dbuf);
    hipStream_t s = ctx->stream;
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_i, prev_img, npx, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_j, next_img, npx, hipMemcpyHostToDevice, s));
    MQS_HIP_CHECK(hipMemcpyAsync(d + o_p, prev_pts, (size_t)n * 8, hipMemcpyHostToDevice, s));
    rc = mqs_calc_optical_flow_pyr_lk_dev((uint8_t *)(d + o_i), (uint8_t *)(d + o_j), W, H, (float *)(d + o_p), n, win_w, win_h,
                                          max_level, max_iter, eps, min_eig_threshold, (float *)(d + o_q), (uint8_t *)(d + o_s),
                                          (float *)(d + o_e), d + o_ws, (int64_t)wsb, s);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(next_pts, d + o_q, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipMemcpyAsync(status, d + o_s, (size_t)n, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipMemcpyAsync(err, d + o_e, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MQS_HIP_CHECK(hipStreamSynchronize(s));
    return MQS_OK;
}

}  // extern "C"
