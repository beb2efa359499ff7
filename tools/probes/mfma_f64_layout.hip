// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment: A = one-hot rows / B = one-hot columns, and
// the position of every product in the 4 result registers of every lane is printed as a formula check.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout tools/probes/mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using double4v = __attribute__((ext_vector_type(4))) double;

__global__ void probe(const double *A /*16x4*/, const double *B /*4x16*/, double *D /*64 lanes x 4*/, int a_mode)
{
    const int lane = threadIdx.x;
    // hypothesis: a[i = lane % 16][k = lane / 16], b[k = lane / 16][j = lane % 16]
    const double a = A[(lane % 16) * 4 + lane / 16];
    const double b = B[(lane / 16) * 16 + lane % 16];
    double4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[lane * 4 + v] = c[v];
}

int main()
{
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 2.0 + 0.5 * j + 0.003 * k * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
        ref[i * 16 + j] = s;
    }
    double *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, 0);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // which (i, j) does lane l, register v hold?
    int ok_rowmajor4 = 1, ok_alt = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const double d = hD[l * 4 + v];
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i * 16 + j] - d) < 1e-9 * fabs(d)) { fi = i; fj = j; }
        (void)fi; (void)fj;
        if (!(fabs(ref[(4 * (l / 16) + v) * 16 + l % 16] - d) < 1e-12 * fabs(d))) ok_rowmajor4 = 0;
        if (!(fabs(ref[((l / 16) + 4 * v) * 16 + l % 16] - d) < 1e-12 * fabs(d))) ok_alt = 0;
        if (l < 2 || l == 17 || l == 63) printf("lane %2d v %d -> D[%2d][%2d]\n", l, v, fi, fj);
    }
    printf("D[i = 4 (lane / 16) + v][j = lane %% 16]: %s\n", ok_rowmajor4 ? "yes" : "no");
    printf("D[i = lane / 16 + 4 v][j = lane %% 16]: %s\n", ok_alt ? "yes" : "no");
    return 0;
}
