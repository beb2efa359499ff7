// Probe: what does ONE wave per SIMD (and two) pay per vector instruction on gfx950, by instruction class?
// Every kernel runs `iters` x 64 instructions of one class per wave on all 1024 SIMDs; the figure printed is the time per
// wave-instruction (1.67 ns = 4 cycles at 2.4 GHz).  Classes:
//   fma_s    v_fma_f64 with one VGPR source (the others scalar), dependent chain / 8 independent chains
//   fma_v    v_fma_f64 with three distinct VGPR-pair sources, 8 chains
//   mul_v    v_mul_f64 with two VGPR-pair sources, 8 chains
//   add_v    v_add_f64, 8 chains
//   acc_rw   v_accvgpr_write_b32 + v_accvgpr_read_b32 pairs
//   cnd      v_cndmask_b32
//   swap32   v_permlane32_swap_b32
//   dpp      v_mov_b32 with a DPP row rotation
//   mix      fma_v interleaved 1:1 with v_accvgpr_read (independent)
//   cnd_s    v_cndmask_b32 with the mask in an SGPR pair (VOP3);  cnd_far  v_cndmask_b32 ... vcc whose sources were written 8 instructions ago
//   bfi      v_bfi_b32 (select through a VGPR mask);  and_b  v_and_b32
// Build: hipcc --offload-arch=gfx950 -O3 -o build/valu_issue tools/probes/valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

enum { FMA_S1, FMA_S8, FMA_V, MUL_V, ADD_V, ACC_RW, CND, SWAP32, DPP, MIX, CND_S, BFI, CND_FAR, AND_B, CND_E64_VCC, CMP_VCC, CMP_SGPR, CMP_CND_VCC, CMP_CND_SGPR };

template <int KIND, int W>
__global__ __launch_bounds__(256, W) void k(double *out, int iters, double a, double b)
{
    extern __shared__ char pad[];
    double v[8], x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; x[i] = 1.0 + 1e-9 * (threadIdx.x + i); y[i] = 1e-9 * i; }
    int ai[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    int acc[8];
    int bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long smask = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
    const int vmask = (threadIdx.x & 1) ? -1 : 0;
    unsigned long long sm[2] = {smask, smask};
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[i]) : "v"(ai[i]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == FMA_S1) v[0] = fma(v[0], a, b);
                else if (KIND == FMA_S8) v[i] = fma(v[i], a, b);
                else if (KIND == FMA_V) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x[i]), "v"(y[i]));
                else if (KIND == MUL_V) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(x[i]));
                else if (KIND == ADD_V) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(y[i]));
                else if (KIND == ACC_RW) {
                    if (u & 1) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[i]) : "v"(ai[i]));
                    else asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ai[i]) : "a"(acc[i]));
                } else if (KIND == CND) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]));
                else if (KIND == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ai[i]), "+v"(ai[(i + 4) & 7]));
                else if (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]));
                else if (KIND == CND_S) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]), "s"(smask));
                else if (KIND == CND_FAR) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(bi[i]) : "v"(ai[i]), "v"(ai[(i + 1) & 7]));
                else if (KIND == BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(ai[i]) : "v"(vmask), "v"(ai[(i + 1) & 7]));
                else if (KIND == AND_B) asm volatile("v_and_b32 %0, %1, %0" : "+v"(ai[i]) : "v"(vmask));
                else if (KIND == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]));
                else if (KIND == CMP_VCC) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(ai[i]), "v"(ai[(i + 1) & 7]) : "vcc");
                else if (KIND == CMP_SGPR) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(sm[i & 1]) : "v"(ai[i]), "v"(ai[(i + 1) & 7]));
                else if (KIND == CMP_CND_VCC) {
                    if (i & 1) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]) : "vcc");
                    else asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(ai[i]), "v"(ai[(i + 1) & 7]) : "vcc");
                } else if (KIND == CMP_CND_SGPR) {
                    if (i & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(ai[i]) : "v"(ai[(i + 1) & 7]), "s"(sm[0]));
                    else asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(sm[0]) : "v"(ai[i]), "v"(ai[(i + 1) & 7]));
                }
                else if (KIND == MIX) {
                    if (i & 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ai[i]) : "a"(acc[i]));
                    else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(x[i]), "v"(y[i]));
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + ai[i] + bi[i];
    s += (double)(sm[0] ^ sm[1]);
    if (s == 12345.678) out[threadIdx.x] = s + pad[0];
}

template <int KIND, int W>
float run(int iters, double *d)
{
    const int blocks = 256 * W;
    const size_t lds = W == 1 ? 150000 : 70000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, W>), dim3(blocks), dim3(256), lds, 0, d, iters, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, W>), dim3(blocks), dim3(256), lds, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv)
{
    double *d; hipMalloc(&d, 4096);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;       // under rocprofv3 --pmc: 100000 (dispatches long enough for GRBM_GUI_ACTIVE to give the clock)
    const double insts = 64.0 * iters;
    const char *names[] = {"fma_s 1 chain", "fma_s 8 chains", "fma_v 8 chains", "mul_v 8 chains", "add_v 8 chains", "accvgpr r/w", "cndmask", "permlane32_swap", "mov dpp", "fma_v + accvgpr_read 1:1", "cndmask sgpr-pair mask", "bfi", "cndmask vcc, far sources, separate dst", "and_b32", "cndmask e64 encoding, vcc mask", "v_cmp e32 (writes vcc)", "v_cmp e64 (writes an sgpr pair)", "v_cmp e32 -> v_cndmask e32 through vcc, 1:1", "v_cmp e64 -> v_cndmask e64 through an sgpr pair, 1:1"};
    float one[] = {run<FMA_S1, 1>(iters, d), run<FMA_S8, 1>(iters, d), run<FMA_V, 1>(iters, d), run<MUL_V, 1>(iters, d), run<ADD_V, 1>(iters, d),
                   run<ACC_RW, 1>(iters, d), run<CND, 1>(iters, d), run<SWAP32, 1>(iters, d), run<DPP, 1>(iters, d), run<MIX, 1>(iters, d), run<CND_S, 1>(iters, d), run<BFI, 1>(iters, d), run<CND_FAR, 1>(iters, d), run<AND_B, 1>(iters, d), run<CND_E64_VCC, 1>(iters, d), run<CMP_VCC, 1>(iters, d), run<CMP_SGPR, 1>(iters, d), run<CMP_CND_VCC, 1>(iters, d), run<CMP_CND_SGPR, 1>(iters, d)};
    float two[] = {run<FMA_S1, 2>(iters, d), run<FMA_S8, 2>(iters, d), run<FMA_V, 2>(iters, d), run<MUL_V, 2>(iters, d), run<ADD_V, 2>(iters, d),
                   run<ACC_RW, 2>(iters, d), run<CND, 2>(iters, d), run<SWAP32, 2>(iters, d), run<DPP, 2>(iters, d), run<MIX, 2>(iters, d), run<CND_S, 2>(iters, d), run<BFI, 2>(iters, d), run<CND_FAR, 2>(iters, d), run<AND_B, 2>(iters, d), run<CND_E64_VCC, 2>(iters, d), run<CMP_VCC, 2>(iters, d), run<CMP_SGPR, 2>(iters, d), run<CMP_CND_VCC, 2>(iters, d), run<CMP_CND_SGPR, 2>(iters, d)};
    printf("{\n");
    for (int i = 0; i < 19; ++i)
        printf(" \"%s\": {\"ns_per_wave_instruction_1_wave_per_simd\": %.2f, \"ns_per_simd_instruction_2_waves_per_simd\": %.2f}%s\n", names[i],
               one[i] * 1e6 / insts, two[i] * 1e6 / insts / 2, i < 18 ? "," : "");
    printf("}\n");
    return 0;
}
