// Set-up of the sparse bundle adjustment on the device: the list of observation pairs grouped by pose pair.
//
// The reduced camera system of bundle_adjust.cpp:268-298's graph gets one 6 x 6 block per pair of poses that share a
// landmark; mqs_sba_linearize_grouped_dev (ba_sparse.hip) sums a block from the GROUP of (landmark, observation a, observation
// b) pairs that feed it, one wavefront per group, in list order -- so the list must be sorted by (pose of a, pose of b) with
// the pairs of one group in landmark order (a stable sort of the landmark-major enumeration).  Round 2 built and sorted the
// list with numpy: 0.2 s for the 2.0 M pairs of the ICL-kt2 shape, 56 x the Levenberg-Marquardt solve it prepares.  Here:
//   pairs_generate_kernel   one thread per pair: landmark by binary search in the per-landmark pair offsets, (row, column) of
//                           the landmark's upper triangle from the local index, key = pose(a) * P + pose(b), value = pair index
//   radix passes            least-significant-digit radix sort, 6-bit digits, STABLE: per tile of 4096 pairs a histogram
//                           (radix_hist_kernel), one exclusive scan over (digit, tile) (radix_scan_kernel), and a scatter in
//                           which every thread owns 16 CONSECUTIVE pairs and ranks them after the lower threads' pairs of the
//                           same digit (radix_scatter_kernel) -- ceil(log2(P^2) / 6) passes, no atomics on the data path
//   gather_flags / group_scan / group_write   the sorted pair lists gathered; group boundaries (where the key changes) counted
//                           per tile, scanned, written in order
// Same output as sparse_ba.group_pairs (numpy stable argsort), element for element.
#include "mqs_common.h"

namespace {

constexpr int kDigitBits = 6, kDigits = 1 << kDigitBits;
constexpr int kThreads = 256, kItems = 16, kTile = kThreads * kItems;

__global__ __launch_bounds__(kThreads) void pairs_generate_kernel(const int64_t *__restrict__ obs_ptr, const int32_t *__restrict__ obs_pose,
                                                                  int64_t N, const int64_t *__restrict__ pair_off, int64_t Q,
                                                                  unsigned long long P, int64_t *__restrict__ pa,
                                                                  int64_t *__restrict__ pb, unsigned long long *__restrict__ key,
                                                                  unsigned int *__restrict__ val)
{
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= Q) return;
    // landmark i with pair_off[i] <= q < pair_off[i + 1]
    int64_t lo = 0, hi = N;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (pair_off[mid] <= q) lo = mid; else hi = mid;
    }
    const int64_t k0 = obs_ptr[lo], k = obs_ptr[lo + 1] - k0, e = q - pair_off[lo];
    // row r of the upper triangle (rows r = 0 .. k-1 hold k - r entries): the largest r with r k - r (r - 1) / 2 <= e
    const double kk = (double)k + 0.5;
    int64_t r = (int64_t)(kk - sqrt(kk * kk - 2.0 * (double)e));
    if (r < 0) r = 0;
    if (r > k - 1) r = k - 1;
    while (r > 0 && r * k - r * (r - 1) / 2 > e) --r;
    while (r + 1 < k && (r + 1) * k - (r + 1) * r / 2 <= e) ++r;
    const int64_t c = r + (e - (r * k - r * (r - 1) / 2));
    const int64_t a = k0 + r, b = k0 + c;
    pa[q] = a; pb[q] = b;
    key[q] = (unsigned long long)obs_pose[a] * P + (unsigned long long)obs_pose[b];
    val[q] = (unsigned int)q;
}

__global__ __launch_bounds__(kThreads) void radix_hist_kernel(const unsigned long long *__restrict__ key, int64_t Q, int shift, int tiles,
                                                              unsigned int *__restrict__ hist /*[kDigits][tiles]*/)
{
    __shared__ unsigned int sH[kDigits];
    const int tid = threadIdx.x, tile = blockIdx.x;
    if (tid < kDigits) sH[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * kTile;
    for (int j = 0; j < kItems; ++j) {
        const int64_t i = base + (int64_t)j * kThreads + tid;       // any order: only the counts matter here
        if (i < Q) atomicAdd(&sH[(unsigned int)(key[i] >> shift) & (kDigits - 1)], 1u);
    }
    __syncthreads();
    if (tid < kDigits) hist[(size_t)tid * tiles + tile] = sH[tid];
}

// exclusive scan of hist in (digit, tile) order, in place; one workgroup
__global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned int *__restrict__ hist, int n)
{
    __shared__ unsigned int sPart[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024, lo = tid * per, hi = min(n, lo + per);
    unsigned int s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    sPart[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned int v = tid >= off ? sPart[tid - off] : 0u;
        __syncthreads();
        sPart[tid] += v;
        __syncthreads();
    }
    unsigned int run = sPart[tid] - s;
    for (int i = lo; i < hi; ++i) { const unsigned int h = hist[i]; hist[i] = run; run += h; }
}

__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(const unsigned long long *__restrict__ key_in, const unsigned int *__restrict__ val_in,
                                                                 int64_t Q, int shift, int tiles, const unsigned int *__restrict__ hist,
                                                                 unsigned long long *__restrict__ key_out, unsigned int *__restrict__ val_out)
{
    // thread t owns items [t * kItems, (t + 1) * kItems) of the tile: thread order = item order, so ranking a thread's items
    // behind the lower threads' items of the same digit keeps the sort stable
    __shared__ unsigned short sCnt[kDigits][kThreads];
    __shared__ unsigned int sBase[kDigits];
    const int tid = threadIdx.x, tile = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)tile * kTile + (int64_t)tid * kItems;
    unsigned long long k[kItems];
    unsigned int v[kItems];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const int64_t i = base + j;
        if (i < Q) { k[j] = key_in[i]; v[j] = val_in[i]; cnt = j + 1; }
    }
    for (int dgt = 0; dgt < kDigits; ++dgt) sCnt[dgt][tid] = 0;
    if (tid < kDigits) sBase[tid] = hist[(size_t)tid * tiles + tile];
#pragma unroll
    for (int j = 0; j < kItems; ++j)
        if (j < cnt) sCnt[(unsigned int)(k[j] >> shift) & (kDigits - 1)][tid] += 1;
    __syncthreads();
    // exclusive scan over the threads, per digit: wave w takes digits w, w + 4, ...; lane l scans threads 4 l .. 4 l + 3
    for (int dgt = wave; dgt < kDigits; dgt += 4) {
        const unsigned int c0 = sCnt[dgt][4 * lane], c1 = sCnt[dgt][4 * lane + 1], c2 = sCnt[dgt][4 * lane + 2], c3 = sCnt[dgt][4 * lane + 3];
        const unsigned int tot = c0 + c1 + c2 + c3;
        unsigned int inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int up = __shfl_up(inc, off);
            if (lane >= off) inc += up;
        }
        const unsigned int ex = inc - tot;
        sCnt[dgt][4 * lane] = (unsigned short)ex;
        sCnt[dgt][4 * lane + 1] = (unsigned short)(ex + c0);
        sCnt[dgt][4 * lane + 2] = (unsigned short)(ex + c0 + c1);
        sCnt[dgt][4 * lane + 3] = (unsigned short)(ex + c0 + c1 + c2);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        if (j < cnt) {
            const unsigned int dgt = (unsigned int)(k[j] >> shift) & (kDigits - 1);
            const unsigned int pos = sBase[dgt] + sCnt[dgt][tid];
            sCnt[dgt][tid] += 1;                                    // this thread's next item of the digit goes behind it
            key_out[pos] = k[j];
            val_out[pos] = v[j];
        }
    }
}

// sorted pair lists gathered; group boundaries (where the key changes) counted per tile
__global__ __launch_bounds__(kThreads) void gather_flags_kernel(const unsigned long long *__restrict__ key, const unsigned int *__restrict__ val,
                                                                int64_t Q, const int64_t *__restrict__ pa, const int64_t *__restrict__ pb,
                                                                int64_t *__restrict__ pair_a, int64_t *__restrict__ pair_b,
                                                                unsigned int *__restrict__ tile_count)
{
    __shared__ unsigned int sC;
    const int tid = threadIdx.x;
    if (tid == 0) sC = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    unsigned int c = 0;
    for (int j = 0; j < kItems; ++j) {
        const int64_t i = base + (int64_t)j * kThreads + tid;       // coalesced; only the count matters here
        if (i < Q) {
            c += (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
            const unsigned int src = val[i];
            pair_a[i] = pa[src];
            pair_b[i] = pb[src];
        }
    }
    atomicAdd(&sC, c);
    __syncthreads();
    if (tid == 0) tile_count[blockIdx.x] = sC;
}

// exclusive scan of the tile counts (one workgroup), the number of groups, and the closing entry of group_ptr
__global__ __launch_bounds__(1024) void group_scan_kernel(unsigned int *__restrict__ tile_count, int tiles, int64_t Q,
                                                          int64_t *__restrict__ group_ptr, int64_t cap, int64_t *__restrict__ n_groups)
{
    __shared__ unsigned int sPart[1024];
    const int tid = threadIdx.x;
    const int per = (tiles + 1023) / 1024, lo = tid * per, hi = min(tiles, lo + per);
    unsigned int s = 0;
    for (int i = lo; i < hi; ++i) s += tile_count[i];
    sPart[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned int v = tid >= off ? sPart[tid - off] : 0u;
        __syncthreads();
        sPart[tid] += v;
        __syncthreads();
    }
    unsigned int run = sPart[tid] - s;
    for (int i = lo; i < hi; ++i) { const unsigned int h = tile_count[i]; tile_count[i] = run; run += h; }
    if (tid == 1023) {
        const int64_t G = (int64_t)sPart[1023];
        n_groups[0] = G;
        if (G < cap) group_ptr[G] = Q;
    }
}

// group_ptr: every thread owns 16 consecutive entries of the sorted list, ranks its boundaries behind the lower threads'
__global__ __launch_bounds__(kThreads) void group_write_kernel(const unsigned long long *__restrict__ key, int64_t Q,
                                                               const unsigned int *__restrict__ tile_base, int64_t *__restrict__ group_ptr,
                                                               int64_t cap)
{
    __shared__ unsigned int sWave[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)tid * kItems;
    unsigned int mask = 0, c = 0;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const int64_t i = base + j;
        if (i < Q && (i == 0 || key[i] != key[i - 1])) { mask |= 1u << j; ++c; }
    }
    unsigned int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int up = __shfl_up(inc, off);
        if (lane >= off) inc += up;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    unsigned int before = inc - c;
    for (int w = 0; w < wave; ++w) before += sWave[w];
    int64_t g = (int64_t)tile_base[blockIdx.x] + before;
#pragma unroll
    for (int j = 0; j < kItems; ++j)
        if (mask & (1u << j)) { if (g < cap) group_ptr[g] = base + j; ++g; }
}

// ---- observations sorted by pose inside every landmark (the order the pair enumeration above and the linearisers assume) ----
// key = landmark * P + pose (the CSR already groups the observations by landmark: the landmark by binary search in obs_ptr),
// value = the observation's index; the radix passes above sort (stable); the poses and the measurements are then gathered.
__global__ __launch_bounds__(kThreads) void obs_keys_kernel(const int64_t *__restrict__ obs_ptr, const int32_t *__restrict__ obs_pose,
                                                            int64_t N, int64_t M, unsigned long long P,
                                                            unsigned long long *__restrict__ key, unsigned int *__restrict__ val)
{
    const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    int64_t lo = 0, hi = N;                              // landmark lo with obs_ptr[lo] <= m < obs_ptr[lo + 1] (empty landmarks skipped)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (obs_ptr[mid] <= m) lo = mid; else hi = mid;
    }
    key[m] = (unsigned long long)lo * P + (unsigned long long)obs_pose[m];
    val[m] = (unsigned int)m;
}

__global__ __launch_bounds__(kThreads) void obs_gather_kernel(const unsigned int *__restrict__ val, int64_t M, const int32_t *__restrict__ pose_in,
                                                              const double2 *__restrict__ uv_in, int32_t *__restrict__ pose_out,
                                                              double2 *__restrict__ uv_out, int32_t *__restrict__ order_out)
{
    const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    const unsigned int src = val[m];
    pose_out[m] = pose_in[src];
    uv_out[m] = uv_in[src];
    if (order_out) order_out[m] = (int32_t)src;
}

int64_t tiles_of(int64_t Q) { return (Q + kTile - 1) / kTile; }

}  // namespace

extern "C" {

int64_t mqs_sba_group_pairs_workspace_bytes(int64_t Q)
{
    if (Q < 0) return 0;
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    // two (key, value) buffers, the unsorted pair lists, the (digit, tile) histogram
    return 2 * (up(Q * 8) + up(Q * 4)) + 2 * up(Q * 8) + up((int64_t)kDigits * tiles_of(Q > 0 ? Q : 1) * 4) + 256;
}

// obs_ptr [N + 1], obs_pose [M] (sorted by pose inside every landmark), pair_off [N + 1] = exclusive prefix sums of
// k (k + 1) / 2 over the landmarks' observation counts k (pair_off[N] = Q): device pointers.  Writes the Q pairs sorted by
// (pose of a, pose of b), stably, to pair_a / pair_b [Q], the group offsets to group_ptr [<= group_cap] and the number of
// groups G to n_groups[0] (device; group_ptr[G] = Q).  G <= min(Q, P (P + 1) / 2): a group_cap of that + 1 always suffices.
int mqs_sba_group_pairs_dev(const int64_t *obs_ptr, const int32_t *obs_pose, int64_t N, const int64_t *pair_off, int64_t Q, int64_t P,
                            int64_t *pair_a, int64_t *pair_b, int64_t *group_ptr, int64_t group_cap, int64_t *n_groups,
                            void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(N >= 0 && Q >= 0 && P >= 1 && Q < 0xffffffffll && P < (1ll << 31), "sizes (Q < 2^32 pairs)");
    MQS_ARG_CHECK(group_ptr && n_groups && group_cap >= 1, "group_ptr, n_groups must not be null");
    MQS_ARG_CHECK(Q == 0 || (obs_ptr && obs_pose && pair_off && pair_a && pair_b && workspace), "pointers must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_group_pairs_workspace_bytes(Q), "workspace too small (mqs_sba_group_pairs_workspace_bytes)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (Q == 0) {
        MQS_HIP_CHECK(hipMemsetAsync(group_ptr, 0, 8, stream));
        MQS_HIP_CHECK(hipMemsetAsync(n_groups, 0, 8, stream));
        return MQS_OK;
    }
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    char *w = static_cast<char *>(workspace);
    unsigned long long *key[2];
    unsigned int *val[2];
    key[0] = reinterpret_cast<unsigned long long *>(w); w += up(Q * 8);
    key[1] = reinterpret_cast<unsigned long long *>(w); w += up(Q * 8);
    val[0] = reinterpret_cast<unsigned int *>(w); w += up(Q * 4);
    val[1] = reinterpret_cast<unsigned int *>(w); w += up(Q * 4);
    int64_t *pa = reinterpret_cast<int64_t *>(w); w += up(Q * 8);
    int64_t *pb = reinterpret_cast<int64_t *>(w); w += up(Q * 8);
    unsigned int *hist = reinterpret_cast<unsigned int *>(w);
    const int tiles = (int)tiles_of(Q);
    hipLaunchKernelGGL(pairs_generate_kernel, dim3((unsigned)((Q + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, obs_ptr, obs_pose, N,
                       pair_off, Q, (unsigned long long)P, pa, pb, key[0], val[0]);
    int bits = 0;
    for (unsigned long long m = (unsigned long long)P * (unsigned long long)P - 1ull; m; m >>= 1) ++bits;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += kDigitBits) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], Q, shift, tiles, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, kDigits * tiles);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], val[cur], Q, shift, tiles, hist, key[1 - cur],
                           val[1 - cur]);
        cur = 1 - cur;
    }
    // the (digit, tile) histogram's space now holds the boundary counts per tile
    hipLaunchKernelGGL(gather_flags_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], val[cur], Q, pa, pb, pair_a, pair_b, hist);
    hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, tiles, Q, group_ptr, group_cap, n_groups);
    hipLaunchKernelGGL(group_write_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], Q, hist, group_ptr, group_cap);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int64_t mqs_sba_sort_observations_workspace_bytes(int64_t M)
{
    if (M < 0) return 0;
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    return 2 * (up(M * 8) + up(M * 4)) + up((int64_t)kDigits * tiles_of(M > 0 ? M : 1) * 4) + 256;
}

// The observations of a CSR-by-landmark problem (obs_ptr [N + 1]; obs_pose_in [M], obs_uv_in [M][2] in ANY order inside a
// landmark) sorted by pose index inside every landmark, stably: obs_pose_out / obs_uv_out (distinct from the inputs), and --
// order_out [M] int32, may be NULL -- which input observation each output one is.  Same result as a stable per-landmark
// argsort (sparse_ba.sort_observations_by_pose, the numpy statement of it: 7 ms of host time at 231 k observations).
int mqs_sba_sort_observations_dev(const int64_t *obs_ptr, const int32_t *obs_pose_in, const double *obs_uv_in, int64_t N, int64_t M, int64_t P,
                                  int32_t *obs_pose_out, double *obs_uv_out, int32_t *order_out, void *workspace, int64_t workspace_bytes,
                                  void *stream_)
{
    MQS_ARG_CHECK(N >= 0 && M >= 0 && P >= 1 && M < 0x7fffffffll && P < (1ll << 31), "sizes (M < 2^31 observations)");
    if (M == 0) return MQS_OK;
    MQS_ARG_CHECK(obs_ptr && obs_pose_in && obs_uv_in && obs_pose_out && obs_uv_out && workspace, "pointers must not be null");
    MQS_ARG_CHECK(obs_pose_out != obs_pose_in && obs_uv_out != obs_uv_in, "the sort is not in place");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_sort_observations_workspace_bytes(M), "workspace too small (mqs_sba_sort_observations_workspace_bytes)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto up = [](int64_t v) { return (v + 255) & ~int64_t(255); };
    char *w = static_cast<char *>(workspace);
    unsigned long long *key[2];
    unsigned int *val[2];
    key[0] = reinterpret_cast<unsigned long long *>(w); w += up(M * 8);
    key[1] = reinterpret_cast<unsigned long long *>(w); w += up(M * 8);
    val[0] = reinterpret_cast<unsigned int *>(w); w += up(M * 4);
    val[1] = reinterpret_cast<unsigned int *>(w); w += up(M * 4);
    unsigned int *hist = reinterpret_cast<unsigned int *>(w);
    const int tiles = (int)tiles_of(M);
    const unsigned grid = (unsigned)((M + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(obs_keys_kernel, dim3(grid), dim3(kThreads), 0, stream, obs_ptr, obs_pose_in, N, M, (unsigned long long)P, key[0], val[0]);
    int bits = 0;
    for (unsigned long long m = (unsigned long long)(N > 0 ? N : 1) * (unsigned long long)P - 1ull; m; m >>= 1) ++bits;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += kDigitBits) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], M, shift, tiles, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, kDigits * tiles);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(kThreads), 0, stream, key[cur], val[cur], M, shift, tiles, hist, key[1 - cur],
                           val[1 - cur]);
        cur = 1 - cur;
    }
    hipLaunchKernelGGL(obs_gather_kernel, dim3(grid), dim3(kThreads), 0, stream, val[cur], M, obs_pose_in, reinterpret_cast<const double2 *>(obs_uv_in),
                       obs_pose_out, reinterpret_cast<double2 *>(obs_uv_out), order_out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // extern "C"
