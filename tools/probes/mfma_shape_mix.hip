// Round-5 probe for the matcher (csrc/match.hip, knn2_mfma_kernel<F16Path, 16, 2, 8>): does the 16x16x32 fp16 matrix instruction deliver more
// than the 32x32x16 one under the MATCHER'S instruction mix -- per 32 train rows x 32 queries x 256 dimensions: the train fragments read from
// LDS (ds_read_b128, one per contraction step and row half, shared by two query tiles), the MFMAs, and the top-2 scan of the previous block's
// 16 values per lane (v_or3 + v_med3 + v_min: 48 vector instructions) -- on {0,1} fp16 descriptors, 8 waves per workgroup, one workgroup per
// CU, everything else of the kernel (LDS-DMA stage fill, windows, result merge) left out?  Same MACs, same LDS bytes, same scan work per
// block in both forms; only the instruction shape differs.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/mfma_shape_mix tools/probes/mfma_shape_mix.hip ; ./mfma_shape_mix [blocks=4096] [reps=40]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f16v = __attribute__((ext_vector_type(16))) float;
using f4v = __attribute__((ext_vector_type(4))) float;
constexpr int KS = 16;                     // 256 dimensions / 16
constexpr int NW = 8;
constexpr int kTiles = 8;                  // train tiles (32 rows) resident in LDS per stage
constexpr int kRowBytes = (2 * KS + 1) * 16;

__device__ __forceinline__ void scan(unsigned &best, unsigned &second, float v, unsigned rowbits)
{
    const unsigned key = __float_as_uint(v) | rowbits;
    unsigned m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best), "v"(second), "v"(key));
    second = m;
    best = min(best, key);
}

template <int SHAPE>
__global__ __launch_bounds__(NW * 64) void mix_kernel(const _Float16 *__restrict__ train, const _Float16 *__restrict__ query, unsigned *__restrict__ out, int stages)
{
    __shared__ __attribute__((aligned(16))) unsigned char sTile[kTiles * 32 * kRowBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int v = tid; v < kTiles * 32 * (2 * KS + 1); v += NW * 64) {
        const int row = v / (2 * KS + 1), col = v % (2 * KS + 1);
        reinterpret_cast<uint4 *>(sTile)[v] = reinterpret_cast<const uint4 *>(train)[(size_t)(blockIdx.x % 64) * kTiles * 32 * 2 * KS + row * 2 * KS + (col < 2 * KS ? col : 0)];
    }
    unsigned best[2] = {0xffffffffu, 0xffffffffu}, second[2] = {0xffffffffu, 0xffffffffu};
    if (SHAPE == 32) {
        const int r = lane & 31, h = lane >> 5;
        half8 qf[2][KS];
        for (int qt = 0; qt < 2; ++qt)
            for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = reinterpret_cast<const half8 *>(query + ((size_t)(wave * 2 + qt) * 32 + r) * 256)[2 * ks + h];
        __syncthreads();
        f16v acc[2], prev[2] = {};
        for (int s = 0; s < stages; ++s)
            for (int tt = 0; tt < kTiles; ++tt) {
                const unsigned char *arow = sTile + (tt * 32 + r) * kRowBytes + 16 * h;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 a = *reinterpret_cast<const half8 *>(arow + 32 * ks);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[qt][ks], ks == 0 ? f16v{} + (float)tt : acc[qt], 0, 0, 0);
                    // the previous block's values: 2 x 16 per lane over the 16 steps
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) scan(best[qt], second[qt], prev[qt][ks], (unsigned)(4 * h + ks));
                }
                prev[0] = acc[0]; prev[1] = acc[1];
            }
    } else {
        // 16x16x32: a[i = lane % 16][k = 8 (lane / 16) + j], b[k = 8 (lane / 16) + j][col = lane % 16], d[i = 4 (lane / 16) + v][col = lane % 16]
        const int r = lane & 15, g = lane >> 4;
        constexpr int K2 = KS / 2;             // 8 steps of 32
        half8 qf[2][2][K2];                    // query tile, column half, step
        for (int qt = 0; qt < 2; ++qt)
            for (int ch = 0; ch < 2; ++ch)
                for (int ks = 0; ks < K2; ++ks) qf[qt][ch][ks] = reinterpret_cast<const half8 *>(query + ((size_t)(wave * 2 + qt) * 32 + 16 * ch + r) * 256)[4 * ks + g];
        __syncthreads();
        f4v acc[2][2][2], prev[2][2][2] = {};  // query tile, row half, column half
        for (int s = 0; s < stages; ++s)
            for (int tt = 0; tt < kTiles; ++tt) {
#pragma unroll
                for (int ks = 0; ks < K2; ++ks) {
                    half8 a[2];
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) a[rh] = *reinterpret_cast<const half8 *>(sTile + (tt * 32 + 16 * rh + r) * kRowBytes + 64 * ks + 16 * g);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                acc[qt][rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rh], qf[qt][ch][ks], ks == 0 ? f4v{} + (float)tt : acc[qt][rh][ch], 0, 0, 0);
                    // 2 x 16 values per lane over the 8 steps: four per step and query tile
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = 4 * (ks & 3) + e;            // rh, ch, v
                            const f4v &p = prev[qt][(ks >> 1) & 1][ks & 1];
                            scan(best[qt], second[qt], p[e], (unsigned)(idx + 4 * g));
                        }
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) prev[qt][rh][ch] = acc[qt][rh][ch];
            }
    }
    out[(size_t)blockIdx.x * NW * 64 + tid] = best[0] ^ second[0] ^ best[1] ^ second[1];
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 40, stages = 64;
    const size_t nt = (size_t)64 * kTiles * 32 * 256, nq = (size_t)NW * 2 * 32 * 256;
    std::vector<_Float16> ht(nt), hq(nq);
    unsigned x = 12345;
    for (auto &v : ht) { x = x * 1664525u + 1013904223u; v = (_Float16)((x >> 16) & 1); }
    for (auto &v : hq) { x = x * 1664525u + 1013904223u; v = (_Float16)(-2.0f * ((x >> 16) & 1)); }
    _Float16 *dt, *dq; unsigned *dout;
    hipMalloc(&dt, nt * 2); hipMalloc(&dq, nq * 2); hipMalloc(&dout, (size_t)blocks * NW * 64 * 4);
    hipMemcpy(dt, ht.data(), nt * 2, hipMemcpyHostToDevice); hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * blocks * NW * 2.0 * 32 * 32 * 256 * kTiles * stages;
    auto run = [&](int shape) {
        float ms = 0;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
            if (shape == 32) hipLaunchKernelGGL(mix_kernel<32>, dim3(blocks), dim3(NW * 64), 0, 0, dt, dq, dout, stages);
            else hipLaunchKernelGGL(mix_kernel<16>, dim3(blocks), dim3(NW * 64), 0, 0, dt, dq, dout, stages);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        return flop * reps / (ms * 1e-3) / 1e12;
    };
    run(32); run(16);                       // warm-up, clock
    printf("{\"blocks\": %d, \"waves_per_cu\": %d, \"rounds\": [", blocks, NW);
    for (int round = 0; round < 4; ++round) {
        const double a = run(32), b = run(16);
        printf("%s{\"mfma_32x32x16_TFLOPs\": %.1f, \"mfma_16x16x32_TFLOPs\": %.1f, \"ratio\": %.3f}", round ? ", " : "", a, b, b / a);
    }
    printf("], \"peak_TFLOPs\": 2500}\n");
    return 0;
}
