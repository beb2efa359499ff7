// Probe: where the prologue of the fused iteration tail (ba_tail_kernel: reduced solve in every workgroup) spends its time.
// One workgroup of 256 threads runs the prologue's phases with 100 MHz wall-clock stamps taken by lane 0 of wave 0:
//   load (reduced system -> LDS, camera blocks staged) | pose-prior terms | factor + solve | publish (dpose, retraction)
// with and without a pose prior, C = 4, each launch right behind a kernel that loads every SIMD (the clock of a busy chip,
// as inside an iteration; an idle chip runs the same code ~2x slower).  Includes ba.hip to reach its internals.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/tail_phases tools/probes/tail_phases.hip \
//        multiple-quadrotor-slam_amd/csrc/api.hip multiple-quadrotor-slam_amd/csrc/comm.hip -ldl ; run on the GPU box.
#include "../../multiple-quadrotor-slam_amd/csrc/ba.hip"
#include <vector>
#include <random>

namespace {
template <int C>
__global__ __launch_bounds__(kBlock, 4) void phases_kernel(const double *lin, const double *poses, const double *calib, const double *sigma,
                                                           const double *prior_poses, const double *prior_sigmas, const uint8_t *prior_mask,
                                                           double *dpose, double *poses_out, double *info, long long *stamps)
{
    constexpr int n = 6 * C, nlin = n * n + n + 2;
    __shared__ double sCam[C * kCamStride];
    __shared__ SolveLds<C> sm;
    const int tid = threadIdx.x;
    long long t0 = wall_clock64();
    for (int k = tid; k < nlin; k += kBlock) sm.lin[k] = lin[k];
    stage_cams<C>(poses, calib, sigma, sCam, tid);
    long long t1 = wall_clock64(), t2 = 0, t3 = 0, t4 = 0;
    if (tid < 64) pose_prior_terms<C>(poses, prior_poses, prior_sigmas, prior_mask, tid, sm.e, sm.w, sm.info);
    __syncthreads();
    build_solve_matrix<C>(sm, 0.0, tid, kBlock);
    __syncthreads();
    if (tid < 64) {
        t2 = wall_clock64();
        bool bad;
        reduced_solve_wave<C>(sm.m, sm.col, sm.y, sm.x, tid, bad);
        t3 = wall_clock64();
        publish_solution<C>(sm, poses, bad, tid, dpose, poses_out, info);
        t4 = wall_clock64();
    }
    __syncthreads();
    if (tid == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t4 - t3; stamps[4] = wall_clock64() - t0; }
}
// keeps every SIMD busy for ~1 ms so that the phases are timed at the clock the part holds under load, as inside an iteration
__global__ __launch_bounds__(256) void busy_kernel(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = fma(a, b, c);
    }
    if (a == 12345.678) out[0] = a;
}
// latency of the pieces of the pivot chain, one wavefront alone on its SIMD: ns per repetition of
//   0: v_rsq_f64 + two Newton steps, each repetition fed by the previous one
//   1: v_readlane pair -> v_fma_f64 with the scalar -> v_readlane pair of the result, chained
//   2: dependent v_fma_f64 chain (one per repetition)
//   3: ds_write_b64 + wait + ds_read_b64 of another lane's value, chained
template <int KIND>
__global__ __launch_bounds__(64) void chain_kernel(double *out, long long *ticks, int reps, double seed)
{
    __shared__ double sbuf[64];
    double v = seed + threadIdx.x * 1e-3, w = 1.0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < reps; ++i) {
        if (KIND == 0) { v = mqs::rsqrt_d(v) + 1.5; }
        else if (KIND == 1) { const double s = read_lane(v, 3); w = fma(-v, s * 1e-9, w); v = read_lane(w, 5) + threadIdx.x * 1e-3; }
        else if (KIND == 2) { v = fma(v, 0.999999, 1e-7); }
        else { sbuf[threadIdx.x] = v; mqs_wave_lds_sync(); v = sbuf[(threadIdx.x + 7) & 63] + 1e-7; mqs_wave_lds_sync(); }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = v + w;
}
}  // namespace

int main()
{
    constexpr int C = 4, n = 24;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> g(0.0, 1.0);
    std::vector<double> B(n * n), lin(n * n + n + 2, 0.0), poses(C * 12, 0.0), calib(C * 9, 0.0), sigma(C, 1.0), psig(C * 6, 0.1);
    for (auto &v : B) v = g(rng);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = (i == j) ? n : 0.0;
            for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
            lin[i * n + j] = s;
        }
    for (int i = 0; i < n; ++i) lin[n * n + i] = g(rng);
    for (int c = 0; c < C; ++c) { poses[12 * c] = poses[12 * c + 4] = poses[12 * c + 8] = 1.0; calib[9 * c] = calib[9 * c + 1] = 480.0; }
    std::vector<double> pp = poses;
    pp[9] = 0.01; pp[1] = 1e-3; pp[3] = -1e-3;
    uint8_t mask[C] = {1, 0, 0, 0};
    double *d_lin, *d_poses, *d_calib, *d_sigma, *d_pp, *d_ps, *d_dpose, *d_out, *d_info;
    uint8_t *d_mask;
    long long *d_st;
    hipMalloc(&d_lin, lin.size() * 8); hipMalloc(&d_poses, 96 * C); hipMalloc(&d_calib, 72 * C); hipMalloc(&d_sigma, 8 * C);
    hipMalloc(&d_pp, 96 * C); hipMalloc(&d_ps, 48 * C); hipMalloc(&d_dpose, 8 * n); hipMalloc(&d_out, 96 * C); hipMalloc(&d_info, 16);
    hipMalloc(&d_mask, C); hipMalloc(&d_st, 64);
    hipMemcpy(d_lin, lin.data(), lin.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_poses, poses.data(), 96 * C, hipMemcpyHostToDevice);
    hipMemcpy(d_calib, calib.data(), 72 * C, hipMemcpyHostToDevice); hipMemcpy(d_sigma, sigma.data(), 8 * C, hipMemcpyHostToDevice);
    hipMemcpy(d_pp, pp.data(), 96 * C, hipMemcpyHostToDevice); hipMemcpy(d_ps, psig.data(), 48 * C, hipMemcpyHostToDevice);
    hipMemcpy(d_mask, mask, C, hipMemcpyHostToDevice);
    for (int with_prior = 0; with_prior < 2; ++with_prior) {
        long long acc[5] = {0, 0, 0, 0, 0};
        const int reps = 50;
        for (int r = 0; r < reps + 5; ++r) {
            hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, 0, d_dpose, r < 5 ? 40000 : 4000);
            hipLaunchKernelGGL((phases_kernel<C>), dim3(1), dim3(kBlock), 0, 0, d_lin, d_poses, d_calib, d_sigma, d_pp, d_ps,
                               with_prior ? d_mask : nullptr, d_dpose, d_out, d_info, d_st);
            long long st[5];
            hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost);
            if (r >= 5) for (int k = 0; k < 5; ++k) acc[k] += st[k];
        }
        printf("{\"pose_prior\": %d, \"ns\": {\"load_and_stage\": %.0f, \"prior_terms_and_matrix\": %.0f, \"factor_and_solve\": %.0f, \"publish\": %.0f, \"total\": %.0f}}\n",
               with_prior, 10.0 * acc[0] / reps, 10.0 * acc[1] / reps, 10.0 * acc[2] / reps, 10.0 * acc[3] / reps, 10.0 * acc[4] / reps);
    }
    {
        const int reps = 2000;
        const char *names[4] = {"rsq_plus_two_newton", "readlane_fma_readlane", "dependent_fma", "lds_write_wait_read"};
        double res[4];
        for (int kind = 0; kind < 4; ++kind) {
            for (int r = 0; r < 3; ++r) {
                hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, 0, d_dpose, 4000);
                if (kind == 0) hipLaunchKernelGGL((chain_kernel<0>), dim3(1), dim3(64), 0, 0, d_lin, d_st, reps, 2.0);
                if (kind == 1) hipLaunchKernelGGL((chain_kernel<1>), dim3(1), dim3(64), 0, 0, d_lin, d_st, reps, 2.0);
                if (kind == 2) hipLaunchKernelGGL((chain_kernel<2>), dim3(1), dim3(64), 0, 0, d_lin, d_st, reps, 2.0);
                if (kind == 3) hipLaunchKernelGGL((chain_kernel<3>), dim3(1), dim3(64), 0, 0, d_lin, d_st, reps, 2.0);
                long long t;
                hipMemcpy(&t, d_st, 8, hipMemcpyDeviceToHost);
                res[kind] = 10.0 * t / reps;
            }
        }
        printf("{\"chain_ns_per_repetition\": {\"%s\": %.1f, \"%s\": %.1f, \"%s\": %.1f, \"%s\": %.1f}}\n", names[0], res[0], names[1], res[1],
               names[2], res[2], names[3], res[3]);
    }
    return 0;
}
