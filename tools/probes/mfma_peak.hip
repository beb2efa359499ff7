// Round-6 probe: what the matrix pipe delivers to a kernel that does NOTHING but matrix instructions on register operands -- the ceiling the
// matcher's fractions of "peak" should be read against (clock under load included).  Two forms: v_mfma_f32_32x32x16_f16 (fp16 path) and
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands, no scales (FP4 path); W waves per workgroup (one workgroup per CU), A independent
// accumulators per wave.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/mfma_peak tools/probes/mfma_peak.hip ; ./tools/probes/mfma_peak [iters=20000] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f16v = __attribute__((ext_vector_type(16))) float;
using int8v = __attribute__((ext_vector_type(8))) int;

template <int FORM, int A>
__global__ __launch_bounds__(512) void peak_kernel(float *out, int iters)
{
    f16v acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = (float)(threadIdx.x + a + e);
    half8 ha, hb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)((threadIdx.x + j) & 1); hb[j] = (_Float16)((threadIdx.x >> 1) & 1); }
    int8v ia = {0x22222222, 0x02020202, 0x20202020, 0x22002200, 0, 0, 0, 0}, ib = {0x0C0C0C0C, (int)0xC0C0C0C0, (int)0xCCCC0000, 0x0000CCCC, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (FORM == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, acc[a], 4, 4, 0, 0, 0, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int FORM, int A>
void run(const char *name, int waves, int iters, int reps, float *out, int cus)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((peak_kernel<FORM, A>), dim3(cus), dim3(waves * 64), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((peak_kernel<FORM, A>), dim3(cus), dim3(waves * 64), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * (FORM == 0 ? 16 : 64) * (double)A * iters * waves * cus * reps;
    const double pf = flop / (ms * 1e-3) / 1e15;
    printf("{\"form\": \"%s\", \"waves_per_cu\": %d, \"accumulators_per_wave\": %d, \"PFLOPs\": %.3f, \"frac_of_nominal_peak\": %.3f, \"cycles_per_mfma_per_simd_at_2.4GHz\": %.1f}\n",
           name, waves, A, pf, pf / (FORM == 0 ? 2.5 : 10.0), 2.4e9 * (ms * 1e-3) / ((double)A * iters * reps * waves / 4.0));
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, reps = argc > 2 ? atoi(argv[2]) : 20;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    float *out;
    (void)hipMalloc(&out, 4096);
    for (int pass = 0; pass < 2; ++pass) {
        run<0, 1>("f16 32x32x16", 4, iters, reps, out, p.multiProcessorCount);
        run<0, 2>("f16 32x32x16", 4, iters, reps, out, p.multiProcessorCount);
        run<0, 4>("f16 32x32x16", 4, iters, reps, out, p.multiProcessorCount);
        run<0, 1>("f16 32x32x16", 8, iters, reps, out, p.multiProcessorCount);
        run<0, 2>("f16 32x32x16", 8, iters, reps, out, p.multiProcessorCount);
        run<1, 1>("fp4 32x32x64", 4, iters, reps, out, p.multiProcessorCount);
        run<1, 2>("fp4 32x32x64", 4, iters, reps, out, p.multiProcessorCount);
        run<1, 4>("fp4 32x32x64", 4, iters, reps, out, p.multiProcessorCount);
        run<1, 1>("fp4 32x32x64", 8, iters, reps, out, p.multiProcessorCount);
        run<1, 2>("fp4 32x32x64", 8, iters, reps, out, p.multiProcessorCount);
    }
    return 0;
}
