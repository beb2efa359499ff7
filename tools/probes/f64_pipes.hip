// Probe: do v_mfma_f64_16x16x4_f64 and fp64 VALU FMAs overlap on gfx950, and what does each cost?
// Measured on MI355X (round 1): they do NOT overlap -- "both" costs the sum of the two -- and the matrix
// instruction sustains less (47 TF) than plain FMAs (66 TF), so an f64 SYRK on the matrix pipe cannot
// take work off the fp64-bound BA kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/f64_pipes.hip -o build/f64_pipes ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

template <int MODE>   // 0: MFMA only, 1: FMA only, 2: both interleaved (1 MFMA : R FMAs)
__global__ __launch_bounds__(256) void probe(double *out, int iters, double seed)
{
    f64x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0;
    double a = seed + threadIdx.x, b = seed * 0.5;
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, d2, 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], 1.0000001, 0.5);
        }
    }
    double s = d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3] + d2[0] + d2[1] + d2[2] + d2[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> float run(double *out, int grid, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    double *out; CK(hipMalloc(&out, 4096 * 256 * 8));
    const int iters = 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {           // workgroups per CU (4 waves each -> waves per SIMD)
        const int grid = 256 * wpc;
        float m = run<0>(out, grid, iters), f = run<1>(out, grid, iters), b = run<2>(out, grid, iters);
        // per iteration per wave: 3 MFMAs (3 x 2048 flop) and 48 FMAs (48 x 128 flop)
        const double waves = grid * 4.0;
        printf("waves/SIMD=%d  mfma-only %.3f ms (%.1f TF)  fma-only %.3f ms (%.1f TF)  both %.3f ms  (sum %.3f, max %.3f)\n", wpc, m,
               waves * iters * 3 * 2048.0 / m * 1e-9, f, waves * iters * 48 * 128.0 / f * 1e-9, b, m + f, m > f ? m : f);
    }
    return 0;
}
