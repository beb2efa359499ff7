// General (sparse-visibility) bundle adjustment for gfx950: any number of poses, each landmark
// observed by an arbitrary subset (CSR by landmark).  This is the graph the reference's tool builds
// from its file set (Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:245-298: 186..881 poses,
// 1k..13k landmarks, 7k..231k observations on the committed data sets), where the dense C <= 8
// kernels of ba.hip do not apply.  Same factor arithmetic (ba_math.h).
//
// Launch structure per linearisation (all asynchronous on one stream):
//   stage_pose_cams   one thread per pose: the 24-double camera block (pose + its camera's calibration)
//   sparse_landmark   one thread per landmark: H_ll, g_l, Cholesky, then one 14-double record per
//                     observation {x, y, Z, Uh (2x3), k (2x2 sym), rh (2)} -- everything a pose-pair
//                     block needs from that observation
//   sparse_pairs      one thread per (landmark, observation pair a <= b) from a precomputed pair list:
//                     the 6x6 block  +/- Jg_a^T k Jg_b  added into the dense (6P)^2 matrix with fp64
//                     global atomics (upper block triangle; pose(a) <= pose(b) by construction)
//   sparse_finish     mirrors the upper triangle, adds pose priors and LM damping
//   sparse_pair_groups  the same blocks WITHOUT atomics when the caller hands over the pair list grouped by pose pair
//                     (mqs_sba_linearize_grouped_dev): one wavefront per group, one writer per block, reproducible
// The reduced system is then factored by the blocked Cholesky below (own kernels, fp64).
// The atomic form is not bitwise reproducible (summation order); the grouped form is.
#include "mqs_common.h"
#include "so3_math.h"
#include <map>
#include <mutex>
#include "ba_math.h"
#include "chol_block.h"
#include "wave_reduce.h"

namespace {

using namespace mqs::ba;
constexpr int kBlock = 256;
constexpr int kRec = 14;      // doubles per observation record

__global__ void stage_pose_cams_kernel(const double *__restrict__ poses, const int32_t *__restrict__ pose_cam,
                                       const double *__restrict__ calib, const double *__restrict__ sigma, int P,
                                       double *__restrict__ cams)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    const int c = pose_cam[j];
    stage_camera(cams + (int64_t)j * kCamStride, poses + (int64_t)j * 12, calib + 9 * c, sigma[c]);
}

// kLanesPerLandmark lanes share a landmark: each takes every kLanesPerLandmark-th observation in both passes, the 3 x 3
// block, its right-hand side, the cost and the count are combined by an xor butterfly inside the lane group (fixed order:
// reproducible), every lane factors the block.  One thread per landmark (the first version) walked a landmark's 17-30
// observations as one chain of dependent loads on 13 k threads -- 0.2 waves per SIMD: 72 us at ICL size, 119 us for the
// 531 landmarks of the image loop.
constexpr int kLanesPerLandmark = 8;

__global__ __launch_bounds__(kBlock) void sparse_landmark_kernel(
    const double *__restrict__ cams, const double *__restrict__ points, const int64_t *__restrict__ obs_ptr,
    const int32_t *__restrict__ obs_pose, const double *__restrict__ obs_uv, const double *__restrict__ prior_w,
    const double *__restrict__ prior_xyz, int64_t N, double lambda, double *__restrict__ rec,
    double *__restrict__ cost_partials)
{
    __shared__ double sRed[2 * (kBlock / 64)];
    const int l = threadIdx.x % kLanesPerLandmark;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / kLanesPerLandmark) + threadIdx.x / kLanesPerLandmark;
    double cost = 0.0, count = 0.0;
    const bool live = i < N;
    {
        const int64_t ii = live ? i : 0;
        const double px = points[3 * ii], py = points[3 * ii + 1], pz = points[3 * ii + 2];
        double pw = 0.0, dx = 0.0, dy = 0.0, dz = 0.0;
        if (live && prior_w && prior_w[ii] > 0.0) {
            pw = prior_w[ii];
            dx = px - prior_xyz[3 * ii]; dy = py - prior_xyz[3 * ii + 1]; dz = pz - prior_xyz[3 * ii + 2];
        }
        PointSystem ps;
        ps.H = mqs::Sym3{0, 0, 0, 0, 0, 0};
        ps.g = mqs::Vec3{0, 0, 0};
        const int64_t k0 = live ? obs_ptr[ii] : 0, k1 = live ? obs_ptr[ii + 1] : 0;
        for (int64_t k = k0 + l; k < k1; k += kLanesPerLandmark) {
            const double *cam = cams + (int64_t)obs_pose[k] * kCamStride;
            const Factor fc = make_factor(cam, px, py, pz, obs_uv[2 * k], obs_uv[2 * k + 1], true);
            double PR[2][3];
            make_PR(cam, fc.x, fc.y, PR);
            point_add_factor(ps, fc, PR);
            cost += fc.half_e2;
            count += fc.valid ? 1.0 : 0.0;
        }
#pragma unroll
        for (int m = kLanesPerLandmark / 2; m >= 1; m >>= 1) {
            ps.H.xx += __shfl_xor(ps.H.xx, m); ps.H.xy += __shfl_xor(ps.H.xy, m); ps.H.xz += __shfl_xor(ps.H.xz, m);
            ps.H.yy += __shfl_xor(ps.H.yy, m); ps.H.yz += __shfl_xor(ps.H.yz, m); ps.H.zz += __shfl_xor(ps.H.zz, m);
            ps.g.x += __shfl_xor(ps.g.x, m); ps.g.y += __shfl_xor(ps.g.y, m); ps.g.z += __shfl_xor(ps.g.z, m);
            cost += __shfl_xor(cost, m); count += __shfl_xor(count, m);
        }
        cost += 0.5 * pw * (dx * dx + dy * dy + dz * dz);
        if (l != 0 || !live) { cost = 0.0; count = 0.0; }          // one lane per landmark reports to the block sum
        point_finish(ps, pw, dx, dy, dz, lambda);
        const double m = ps.ok ? 1.0 : 0.0;
        double w0 = ps.g.x * ps.i00;
        double w1 = fma(-ps.l10, w0, ps.g.y) * ps.i11;
        double w2 = fma(-ps.l21, w1, fma(-ps.l20, w0, ps.g.z)) * ps.i22;
        w0 *= m; w1 *= m; w2 *= m;
        for (int64_t k = k0 + l; k < k1; k += kLanesPerLandmark) {
            const double *cam = cams + (int64_t)obs_pose[k] * kCamStride;
            const Factor fc = make_factor(cam, px, py, pz, obs_uv[2 * k], obs_uv[2 * k + 1], true);
            double PR[2][3];
            make_PR(cam, fc.x, fc.y, PR);
            double U[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double Fa = r ? fc.F01 : fc.F00, Fb = r ? fc.F11 : fc.F01;
                double u0 = fma(Fa, PR[0][0], Fb * PR[1][0]);
                double u1 = fma(Fa, PR[0][1], Fb * PR[1][1]);
                double u2 = fma(Fa, PR[0][2], Fb * PR[1][2]);
                apply_LinvT(ps, u0, u1, u2);
                U[r][0] = m * u0; U[r][1] = m * u1; U[r][2] = m * u2;
            }
            double *o = rec + k * kRec;
            o[0] = fc.x; o[1] = fc.y; o[2] = fc.Z;
            o[3] = U[0][0]; o[4] = U[0][1]; o[5] = U[0][2]; o[6] = U[1][0]; o[7] = U[1][1]; o[8] = U[1][2];
            o[9] = fc.F00 - (U[0][0] * U[0][0] + U[0][1] * U[0][1] + U[0][2] * U[0][2]);
            o[10] = fc.F01 - (U[0][0] * U[1][0] + U[0][1] * U[1][1] + U[0][2] * U[1][2]);
            o[11] = fc.F11 - (U[1][0] * U[1][0] + U[1][1] * U[1][1] + U[1][2] * U[1][2]);
            o[12] = -fc.f0 - (U[0][0] * w0 + U[0][1] * w1 + U[0][2] * w2);
            o[13] = -fc.f1 - (U[1][0] * w0 + U[1][1] * w1 + U[1][2] * w2);
        }
    }
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) { cost += __shfl_xor(cost, h); count += __shfl_xor(count, h); }
    if ((threadIdx.x & 63) == 0) { sRed[2 * (threadIdx.x >> 6)] = cost; sRed[2 * (threadIdx.x >> 6) + 1] = count; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, n = 0;
        for (int w = 0; w < kBlock / 64; ++w) { c += sRed[2 * w]; n += sRed[2 * w + 1]; }
        cost_partials[2 * blockIdx.x] = c;
        cost_partials[2 * blockIdx.x + 1] = n;
    }
}

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// pair (a, b): global observation indices of one landmark, pose(a) <= pose(b); a == b: diagonal block
__global__ __launch_bounds__(kBlock) void sparse_pairs_kernel(const double *__restrict__ rec,
                                                              const int32_t *__restrict__ obs_pose,
                                                              const int64_t *__restrict__ pair_a,
                                                              const int64_t *__restrict__ pair_b, int64_t Q, int n6,
                                                              double *__restrict__ S, double *__restrict__ g)
{
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const int64_t a = pair_a[q], b = pair_b[q];
    const double *ra = rec + a * kRec, *rb = rec + b * kRec;
    const int ja = obs_pose[a], jb = obs_pose[b];
    const JgA A = make_JgA(ra[0], ra[1], ra[2]);
    double T[2][6];
    if (a == b) {
        k_times_Jg(ra[9], ra[10], ra[10], ra[11], A, T);
        double *Sd = S + (int64_t)(6 * ja) * n6 + 6 * ja;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) atomic_add_f64(Sd + (int64_t)i * n6 + j, JgT_T(A, T, i, j));
#pragma unroll
        for (int i = 0; i < 6; ++i) atomic_add_f64(g + 6 * ja + i, JgT_r(A, ra[12], ra[13], i));
    } else {
        const JgA B = make_JgA(rb[0], rb[1], rb[2]);
        const double k00 = -(ra[3] * rb[3] + ra[4] * rb[4] + ra[5] * rb[5]);
        const double k01 = -(ra[3] * rb[6] + ra[4] * rb[7] + ra[5] * rb[8]);
        const double k10 = -(ra[6] * rb[3] + ra[7] * rb[4] + ra[8] * rb[5]);
        const double k11 = -(ra[6] * rb[6] + ra[7] * rb[7] + ra[8] * rb[8]);
        k_times_Jg(k00, k01, k10, k11, B, T);
        if (ja != jb) {
            double *Sd = S + (int64_t)(6 * ja) * n6 + 6 * jb;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) atomic_add_f64(Sd + (int64_t)i * n6 + j, JgT_T(A, T, i, j));
        } else {
            // the same pose observes the landmark twice: block and its transpose land on one diagonal
            // block whose upper triangle is what is kept
            double *Sd = S + (int64_t)(6 * ja) * n6 + 6 * ja;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double v = JgT_T(A, T, i, j);
                    if (i <= j) atomic_add_f64(Sd + (int64_t)i * n6 + j, v);
                    if (j <= i) atomic_add_f64(Sd + (int64_t)j * n6 + i, v);
                }
        }
    }
}

// The same blocks without atomics: the pair list sorted by (pose a, pose b) and cut into groups of equal key
// (group_ptr [G + 1]); one wavefront per group accumulates its pairs' 6 x 6 contributions in registers (lane-strided),
// reduces them over the wave in a fixed butterfly and WRITES the block -- every block of the reduced system has exactly
// one writer, so the result is bitwise reproducible, and the 73 M fp64 atomics of an ICL-sized problem (135 per
// address on average, 2.97 ms) become 2 M record reads (0.2 ms).
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(kBlock) void sparse_pair_groups_kernel(const double *__restrict__ rec,
                                                                    const int32_t *__restrict__ obs_pose,
                                                                    const int64_t *__restrict__ pair_a,
                                                                    const int64_t *__restrict__ pair_b,
                                                                    const int64_t *__restrict__ group_ptr, int64_t G, int n6,
                                                                    double *__restrict__ S, double *__restrict__ g)
{
    const int64_t grp = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (grp >= G) return;
    const int64_t q0 = group_ptr[grp], q1 = group_ptr[grp + 1];
    if (q1 <= q0) return;
    const int ja = obs_pose[pair_a[q0]], jb = obs_pose[pair_b[q0]];
    double acc[36], gacc[6];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) gacc[e] = 0.0;
    for (int64_t q = q0 + lane; q < q1; q += 64) {
        const int64_t a = pair_a[q], b = pair_b[q];
        const double *ra = rec + a * kRec, *rb = rec + b * kRec;
        const JgA A = make_JgA(ra[0], ra[1], ra[2]);
        double T[2][6];
        if (a == b) {
            k_times_Jg(ra[9], ra[10], ra[10], ra[11], A, T);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) acc[i * 6 + j] += JgT_T(A, T, i, j);
#pragma unroll
            for (int i = 0; i < 6; ++i) gacc[i] += JgT_r(A, ra[12], ra[13], i);
        } else {
            const JgA B = make_JgA(rb[0], rb[1], rb[2]);
            const double k00 = -(ra[3] * rb[3] + ra[4] * rb[4] + ra[5] * rb[5]);
            const double k01 = -(ra[3] * rb[6] + ra[4] * rb[7] + ra[5] * rb[8]);
            const double k10 = -(ra[6] * rb[3] + ra[7] * rb[4] + ra[8] * rb[5]);
            const double k11 = -(ra[6] * rb[6] + ra[7] * rb[7] + ra[8] * rb[8]);
            k_times_Jg(k00, k01, k10, k11, B, T);
            if (ja != jb) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i * 6 + j] += JgT_T(A, T, i, j);
            } else {
                // the same pose observes the landmark twice: the block and its transpose land on one diagonal block
                // whose upper triangle is what is kept
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const double v = JgT_T(A, T, i, j);
                        if (i <= j) acc[i * 6 + j] += v;
                        if (j <= i) acc[j * 6 + i] += v;
                    }
            }
        }
    }
    // 36 sums over the wavefront: the first 32 by the transposed reduction of the dense lineariser (wave_reduce.h: 32 exchange-
    // and-add steps, lane l ends with the total of entry l >> 1), the last four by butterflies -- 36 butterflies were 216
    // exchange steps, more vector instructions than the pairs' arithmetic of a typical group (about 140 pairs)
    double first32[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) first32[e] = acc[e];
    const double t32 = mqs::wave::wave_reduce32(first32, lane);
#pragma unroll
    for (int e = 32; e < 36; ++e) acc[e] = wave_sum(acc[e]);
    double *Sd = S + (int64_t)(6 * ja) * n6 + 6 * jb;
    if (ja == jb) {
#pragma unroll
        for (int e = 0; e < 6; ++e) gacc[e] = wave_sum(gacc[e]);
        if (lane < 6) {
            double gv = gacc[0];
#pragma unroll
            for (int e = 1; e < 6; ++e) gv = (lane == e) ? gacc[e] : gv;
            g[6 * ja + lane] = gv;
        }
    }
    // entry e < 32 is on lane 2 e; entries 32..35 (every lane has them) are written by the odd lanes 1, 3, 5, 7
    const int e_out = (lane & 1) ? 32 + (lane >> 1) : (lane >> 1);
    if (!(lane & 1) || lane < 8) {
        double v = t32;
        if (lane & 1) {
            v = acc[32];
#pragma unroll
            for (int e = 33; e < 36; ++e) v = (e_out == e) ? acc[e] : v;
        }
        const int i = e_out / 6, j = e_out % 6;
        // the block and its mirror image: pairs are ordered (ja <= jb: observations sorted by pose inside a landmark), so the
        // transposed block is nobody else's -- a full-matrix mirror pass afterwards (n^2 reads and writes: 137 us of the
        // 340 us linearisation at n = 5286) is not needed on this path
        if (ja != jb || i <= j) {
            Sd[(int64_t)i * n6 + j] = v;
            S[(int64_t)(6 * jb + j) * n6 + 6 * ja + i] = v;
        }
    }
}

// The grouped pair stage has ONE writer per 6 x 6 block and its mirror image.  That holds only if the caller's grouping is
// canonical: pose(pair_a) <= pose(pair_b) for the group's pairs and no two groups with the same pose pair (the key
// pose_a * P + pose_b strictly increases from group to group).  Checked here per group, without a host round trip: the
// number of violating groups is reported in info[3] (0 for a valid grouping; S is not to be trusted otherwise).
__global__ __launch_bounds__(kBlock) void sparse_check_groups_kernel(const int32_t *__restrict__ obs_pose,
                                                                     const int64_t *__restrict__ pair_a,
                                                                     const int64_t *__restrict__ pair_b,
                                                                     const int64_t *__restrict__ group_ptr, int64_t G, int64_t P,
                                                                     double *__restrict__ violations)
{
    const int64_t grp = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (grp >= G) return;
    const int64_t q0 = group_ptr[grp], q1 = group_ptr[grp + 1];
    if (q1 <= q0) return;
    const int64_t ja = obs_pose[pair_a[q0]], jb = obs_pose[pair_b[q0]];
    bool bad = ja > jb || obs_pose[pair_a[q1 - 1]] != ja || obs_pose[pair_b[q1 - 1]] != jb;
    if (grp > 0 && group_ptr[grp] > group_ptr[grp - 1]) {
        const int64_t p0 = group_ptr[grp - 1];
        const int64_t ka = obs_pose[pair_a[p0]], kb = obs_pose[pair_b[p0]];
        bad = bad || !(ka * P + kb < ja * P + jb);
    }
    if (bad) atomic_add_f64(violations, 1.0);
}

__device__ void so3_log_s(const double *R, double w[3])
{
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    const double k = mqs::so3_log_factor(0.5 * (R[0] + R[4] + R[8] - 1.0), 0.25 * fma(vx, vx, fma(vy, vy, vz * vz)));
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
}

// mirror upper -> lower, then pose priors (thread per prior) and damping on the diagonal
__global__ __launch_bounds__(kBlock) void sparse_mirror_kernel(double *__restrict__ S, int n6)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)n6 * n6) return;
    const int i = (int)(t / n6), j = (int)(t % n6);
    if (j < i) S[t] = S[(int64_t)j * n6 + i];
}

__global__ void sparse_priors_kernel(double *__restrict__ S, double *__restrict__ g, int n6,
                                     const double *__restrict__ poses, const int32_t *__restrict__ prior_idx,
                                     const double *__restrict__ prior_poses, const double *__restrict__ prior_sigmas,
                                     int nprior, double lambda, double *__restrict__ prior_cost)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nprior) {
        const int j = prior_idx[t];
        const double *T0 = prior_poses + 12 * t, *T = poses + 12 * (int64_t)j, *sg = prior_sigmas + 6 * t;
        double Rr[9], w[3], e[6];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rr[3 * a + b] = T0[a] * T[b] + T0[3 + a] * T[3 + b] + T0[6 + a] * T[6 + b];
        so3_log_s(Rr, w);
        const double dt[3] = {T[9] - T0[9], T[10] - T0[10], T[11] - T0[11]};
        for (int a = 0; a < 3; ++a) {
            e[a] = w[a];
            e[3 + a] = T0[a] * dt[0] + T0[3 + a] * dt[1] + T0[6 + a] * dt[2];
        }
        double c = 0.0;
        for (int a = 0; a < 6; ++a) {
            const double wi = 1.0 / (sg[a] * sg[a]);
            if (S) {                                                   // S == nullptr: the factors' cost only (LM trial evaluation)
                atomic_add_f64(S + (int64_t)(6 * j + a) * n6 + 6 * j + a, wi);
                atomic_add_f64(g + 6 * j + a, -wi * e[a]);
            }
            c += 0.5 * wi * e[a] * e[a];
        }
        atomic_add_f64(prior_cost, c);
    }
}

// Odometry: BetweenFactor<Pose3>(from, to, measured, sigmas) of bundle_adjust.cpp:301-309 with GTSAM 3.2.1's
// conventions: h = T_from^-1 T_to, error = measured.localCoordinates(h) = (Log(Rm^T Rh), Rm^T (th - tm)) in the
// first-order chart, Jacobians those of `between` only (H_from = -Ad(h^-1), H_to = I), whitened by the six sigmas.
// One thread per factor; adds J^T J / -J^T r into the MIRRORED system (both triangles) and 0.5 |r|^2 into cost.
// S == nullptr: cost only.
__global__ void sparse_between_kernel(double *__restrict__ S, double *__restrict__ g, int n6,
                                      const double *__restrict__ poses, const int32_t *__restrict__ from,
                                      const int32_t *__restrict__ to, const double *__restrict__ meas,
                                      const double *__restrict__ sigmas, int n, double *__restrict__ cost)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int a = from[t], b = to[t];
    const double *T1 = poses + 12 * (int64_t)a, *T2 = poses + 12 * (int64_t)b, *Tm = meas + 12 * (int64_t)t;
    double Rh[9], th[3], Re[9], w[3], e[6];
    const double dt[3] = {T2[9] - T1[9], T2[10] - T1[10], T2[11] - T1[11]};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rh[3 * i + j] = T1[i] * T2[j] + T1[3 + i] * T2[3 + j] + T1[6 + i] * T2[6 + j];
        th[i] = T1[i] * dt[0] + T1[3 + i] * dt[1] + T1[6 + i] * dt[2];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Re[3 * i + j] = Tm[i] * Rh[j] + Tm[3 + i] * Rh[3 + j] + Tm[6 + i] * Rh[6 + j];
    so3_log_s(Re, w);
    for (int i = 0; i < 3; ++i) {
        e[i] = w[i];
        e[3 + i] = Tm[i] * (th[0] - Tm[9]) + Tm[3 + i] * (th[1] - Tm[10]) + Tm[6 + i] * (th[2] - Tm[11]);
    }
    double W[6], c = 0.0;
    for (int i = 0; i < 6; ++i) {
        W[i] = 1.0 / (sigmas[6 * t + i] * sigmas[6 * t + i]);
        c += 0.5 * W[i] * e[i] * e[i];
    }
    atomic_add_f64(cost, c);
    if (!S) return;
    // H1 = -Ad(h^-1),  h^-1 = (Rh^T, tp = -Rh^T th):  Ad = [[Rh^T, 0], [[tp]x Rh^T, Rh^T]]   (order [omega, v])
    double Rt[9], tp[3], H1[36];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = Rh[3 * j + i];
        tp[i] = -(Rh[i] * th[0] + Rh[3 + i] * th[1] + Rh[6 + i] * th[2]);
    }
    const double K[9] = {0.0, -tp[2], tp[1], tp[2], 0.0, -tp[0], -tp[1], tp[0], 0.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            H1[6 * i + j] = -Rt[3 * i + j];
            H1[6 * i + 3 + j] = 0.0;
            H1[6 * (3 + i) + j] = -(K[3 * i] * Rt[j] + K[3 * i + 1] * Rt[3 + j] + K[3 * i + 2] * Rt[6 + j]);
            H1[6 * (3 + i) + 3 + j] = -Rt[3 * i + j];
        }
    for (int i = 0; i < 6; ++i) {
        double gi = 0.0;
        for (int k = 0; k < 6; ++k) gi += H1[6 * k + i] * W[k] * e[k];
        atomic_add_f64(g + 6 * a + i, -gi);
        atomic_add_f64(g + 6 * b + i, -W[i] * e[i]);
        atomic_add_f64(S + (int64_t)(6 * b + i) * n6 + 6 * b + i, W[i]);
        for (int j = 0; j < 6; ++j) {
            double sij = 0.0;
            for (int k = 0; k < 6; ++k) sij += H1[6 * k + i] * W[k] * H1[6 * k + j];
            atomic_add_f64(S + (int64_t)(6 * a + i) * n6 + 6 * a + j, sij);
            const double cross = H1[6 * j + i] * W[j];                       // (H1^T W)[i][j]
            atomic_add_f64(S + (int64_t)(6 * a + i) * n6 + 6 * b + j, cross);
            atomic_add_f64(S + (int64_t)(6 * b + j) * n6 + 6 * a + i, cross);
        }
    }
}

__global__ void sparse_damp_kernel(double *__restrict__ S, int n6, double lambda)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n6) {
        double &d = S[(int64_t)i * n6 + i];
        d = (lambda >= 0.0) ? d * (1.0 + lambda) : d - lambda;       // lambda < 0: |lambda| * I (GTSAM 3.2.1's default damping)
    }
}

// The solve's input contract is the usual one of a Cholesky routine: the LOWER triangle (with the diagonal).  The factor
// kernels read their panel input from the mirror image in the upper triangle (chol_step_kernel), so the natural-order
// solve first copies lower -> upper inside the band: 32 x 32 tiles through LDS, one workgroup per tile, grid = block rows x
// (band tiles + 1).  A caller that hands over both triangles (the linearisers here do) gets the same bits as before.
// (The chunked solve has its own pass over the tiles of its plan: nd_mirror_kernel, chol_nd.hip.)
__global__ __launch_bounds__(256) void sparse_mirror_lower_kernel(double *__restrict__ S, int n6)
{
    __shared__ double sT[32][33];
    const int br = blockIdx.x, off = blockIdx.y, bc = br - off;
    if (bc < 0) return;
    const int r0 = 32 * br, c0 = 32 * bc;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        sT[ty + 8 * k][tx] = (r < n6 && c < n6) ? S[(int64_t)r * n6 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // element (c0 + ty', r0 + tx) of the upper triangle = element (r0 + tx, c0 + ty') of the lower one
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < n6 && c < n6 && r > c) S[(int64_t)c * n6 + r] = sT[tx][ty + 8 * k];     // whole tiles: the factor kernels work on 32 x 32 blocks
    }
}

__global__ __launch_bounds__(kBlock) void sparse_backsub_kernel(
    const double *__restrict__ cams, const double *__restrict__ points, const int64_t *__restrict__ obs_ptr,
    const int32_t *__restrict__ obs_pose, const double *__restrict__ obs_uv, const double *__restrict__ prior_w,
    const double *__restrict__ prior_xyz, int64_t N, double lambda, const double *__restrict__ dpose,
    double *__restrict__ points_out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const double px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
    double pw = 0.0, dx = 0.0, dy = 0.0, dz = 0.0;
    if (prior_w && prior_w[i] > 0.0) {
        pw = prior_w[i];
        dx = px - prior_xyz[3 * i]; dy = py - prior_xyz[3 * i + 1]; dz = pz - prior_xyz[3 * i + 2];
    }
    PointSystem ps;
    ps.H = mqs::Sym3{0, 0, 0, 0, 0, 0};
    ps.g = mqs::Vec3{0, 0, 0};
    double rx = 0, ry = 0, rz = 0;
    for (int64_t k = obs_ptr[i]; k < obs_ptr[i + 1]; ++k) {
        const int j = obs_pose[k];
        const double *cam = cams + (int64_t)j * kCamStride;
        const Factor fc = make_factor(cam, px, py, pz, obs_uv[2 * k], obs_uv[2 * k + 1], true);
        double PR[2][3];
        make_PR(cam, fc.x, fc.y, PR);
        point_add_factor(ps, fc, PR);
        double Jg[2][6];
        make_Jg(fc.x, fc.y, fc.Z, Jg);
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { s0 = fma(Jg[0][c], dpose[6 * j + c], s0); s1 = fma(Jg[1][c], dpose[6 * j + c], s1); }
        const double t0 = fma(fc.F00, s0, fc.F01 * s1), t1 = fma(fc.F01, s0, fc.F11 * s1);
        rx = fma(PR[0][0], t0, fma(PR[1][0], t1, rx));
        ry = fma(PR[0][1], t0, fma(PR[1][1], t1, ry));
        rz = fma(PR[0][2], t0, fma(PR[1][2], t1, rz));
    }
    point_finish(ps, pw, dx, dy, dz, lambda);
    double v0 = ps.g.x - rx, v1 = ps.g.y - ry, v2 = ps.g.z - rz;
    v0 = v0 * ps.i00;
    v1 = fma(-ps.l10, v0, v1) * ps.i11;
    v2 = fma(-ps.l21, v1, fma(-ps.l20, v0, v2)) * ps.i22;
    apply_Lt_inv(ps, v0, v1, v2);
    const double m = ps.ok ? 1.0 : 0.0;
    points_out[3 * i] = px + m * v0;
    points_out[3 * i + 1] = py + m * v1;
    points_out[3 * i + 2] = pz + m * v2;
}

__global__ __launch_bounds__(kBlock) void sparse_cost_kernel(
    const double *__restrict__ cams, const double *__restrict__ points, const int64_t *__restrict__ obs_ptr,
    const int32_t *__restrict__ obs_pose, const double *__restrict__ obs_uv, const double *__restrict__ prior_w,
    const double *__restrict__ prior_xyz, int64_t N, double *__restrict__ cost_partials)
{
    __shared__ double sRed[2 * (kBlock / 64)];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double cost = 0.0, count = 0.0;
    if (i < N) {
        const double px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        if (prior_w && prior_w[i] > 0.0) {
            const double dx = px - prior_xyz[3 * i], dy = py - prior_xyz[3 * i + 1], dz = pz - prior_xyz[3 * i + 2];
            cost = 0.5 * prior_w[i] * (dx * dx + dy * dy + dz * dz);
        }
        for (int64_t k = obs_ptr[i]; k < obs_ptr[i + 1]; ++k) {
            const Factor fc = make_factor(cams + (int64_t)obs_pose[k] * kCamStride, px, py, pz, obs_uv[2 * k],
                                          obs_uv[2 * k + 1], true);
            cost += fc.half_e2;
            count += fc.valid ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) { cost += __shfl_xor(cost, h); count += __shfl_xor(count, h); }
    if ((threadIdx.x & 63) == 0) { sRed[2 * (threadIdx.x >> 6)] = cost; sRed[2 * (threadIdx.x >> 6) + 1] = count; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, n = 0;
        for (int w = 0; w < kBlock / 64; ++w) { c += sRed[2 * w]; n += sRed[2 * w + 1]; }
        cost_partials[2 * blockIdx.x] = c;
        cost_partials[2 * blockIdx.x + 1] = n;
    }
}

// The screen of an adjustment (slam_device.py: a landmark whose worst observation misses the estimate by more than a bound
// sits out): per landmark the largest PIXEL residual of its observations; +inf when one of them lies behind its camera.
__global__ __launch_bounds__(kBlock) void sparse_worst_residual_kernel(
    const double *__restrict__ cams, const double *__restrict__ points, const int64_t *__restrict__ obs_ptr,
    const int32_t *__restrict__ obs_pose, const double *__restrict__ obs_uv, const int32_t *__restrict__ pose_cam,
    const double *__restrict__ sigma, int64_t N, double *__restrict__ worst, double *__restrict__ min_depth)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const double px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
    double w = 0.0, zmin = HUGE_VAL;
    for (int64_t k = obs_ptr[i]; k < obs_ptr[i + 1]; ++k) {
        const int j = obs_pose[k];
        const double *cam = cams + (int64_t)j * kCamStride;
        const Factor fc = make_factor(cam, px, py, pz, obs_uv[2 * k], obs_uv[2 * k + 1], true);
        const double r = fc.valid ? sqrt(2.0 * fc.half_e2) * sigma[pose_cam[j]] : HUGE_VAL;       // the factor is whitened by sigma
        w = fmax(w, r);
        // depth along the optical axis (make_factor's Z; a factor behind its camera reports Z = 1, hence again from the block)
        zmin = fmin(zmin, fma(cam[2], px - cam[9], fma(cam[5], py - cam[10], cam[8] * (pz - cam[11]))));
    }
    worst[i] = w;
    if (min_depth) min_depth[i] = zmin;
}

__global__ __launch_bounds__(kBlock) void sum_cost_partials_kernel(const double *__restrict__ partials, int n,
                                                                  double *__restrict__ out)
{
    __shared__ double sC[kBlock], sN[kBlock];
    double c = 0, m = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) { c += partials[2 * i]; m += partials[2 * i + 1]; }
    sC[threadIdx.x] = c; sN[threadIdx.x] = m;
    __syncthreads();
    for (int h = kBlock / 2; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) { sC[threadIdx.x] += sC[threadIdx.x + h]; sN[threadIdx.x] += sN[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sC[0]; out[1] = sN[0]; }
}

__global__ void sparse_retract_kernel(const double *__restrict__ poses, const double *__restrict__ dpose, int P,
                                      double *__restrict__ poses_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    const double *T = poses + 12 * (int64_t)j, *d = dpose + 6 * (int64_t)j;
    const double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double a, b;
    mqs::so3_exp_factors(th2, a, b);
    const double K[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
    double E[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const double k2 = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
            E[3 * r + c] = ((r == c) ? 1.0 : 0.0) + a * K[3 * r + c] + b * k2;
        }
    double *O = poses_out + 12 * (int64_t)j;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) O[3 * r + c] = T[3 * r] * E[c] + T[3 * r + 1] * E[3 + c] + T[3 * r + 2] * E[6 + c];
    for (int r = 0; r < 3; ++r) O[9 + r] = T[9 + r] + T[3 * r] * d[3] + T[3 * r + 1] * d[4] + T[3 * r + 2] * d[5];
}

// ---------------------------------------------------------------------------------------------
// Cholesky solve of the reduced camera system, n = 6P: right-looking, block size 32, lower triangle, in place, row-major.
// One launch per block column (chol_step_kernel: panel + trailing update on the fp64 matrix pipe + the next diagonal block),
// then forward / backward substitution.  The same kernels at every size and band width -- no library: rocSOLVER's POTRF /
// POTRS, which round 1 used above 1 536 dense unknowns, took 4.7 / 9.9 / 20.3 ms at n = 1 560 / 3 000 / 5 286 (plus 160 ms of
// start-up on the first call) where these take 1.8 / 4.5 / 13.9 ms.  Long narrow bands are cut into chunks (chol_nd.hip).
// ---------------------------------------------------------------------------------------------
constexpr int NB = mqs::chol::NB;

// The first diagonal block (the later ones are factored inside chol_step_kernel): one wavefront, the block staged through LDS
// for mqs::chol::factor_diag_block_from_lds (chol_block.h: L below the diagonal, inv(L)^T above it).
__global__ __launch_bounds__(64) void chol_diag_kernel(double *__restrict__ A, int n, int k0, int *__restrict__ bad)
{
    __shared__ double sT[NB * mqs::chol::kLd];
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    for (int e = threadIdx.x; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        if (r < nb && c <= r) sT[r * mqs::chol::kLd + c] = A[(int64_t)(k0 + r) * n + k0 + c];
    }
    mqs_wave_lds_sync();
    mqs::chol::factor_diag_block_from_lds(sT, A, n, k0, bad, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch per block column instead of three: panel, trailing update AND the next diagonal block's factorisation.
//
// The three-launch form costs 8.8 + 6.3 + 7.2 us per block column, each launch starting cold.  Fused, the dependency
// between the panel (X = A inv(L)^T) and the update that consumes it crosses workgroups -- so every update tile forms the
// two panel blocks it needs itself (2 x 32 x 32 x 32 FMAs: nothing next to a launch), and the tile that IS the next
// diagonal block factors it on the spot (wave 0, the chol_diag_kernel arithmetic on the tile it has just updated in LDS).
// What makes this race-free without a grid-wide barrier is where a step READS its panel input: from the mirror image in
// the upper triangle (every update writes its tile to both triangles; the symmetric input has both), while the finished
// panel X -- the final L entries -- is WRITTEN to the lower triangle by the tiles of the first tile column.  Nobody reads
// what another workgroup of the same launch writes.  Diagonal blocks keep the established layout (L below, inv(L)^T above).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void chol_step_kernel(double *__restrict__ A, int n, int k0, int hb, int *__restrict__ bad)
{
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int t0 = k0 + nb;
    int rem = n - t0;
    if (rem > hb) rem = hb;
    const int i0 = t0 + bi * NB, j0 = t0 + bj * NB;             // first rows of the two panel blocks / the tile's origin
    const int lim = t0 + rem;                                    // rows [t0, lim) take part
    __shared__ double sLi[NB][NB + 1], sAi[NB][NB + 1], sAj[NB][NB + 1], sXi[NB][NB + 1], sXj[NB][NB + 1];
    __shared__ double sM[128];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += kBlock) {
        const int a = e / NB, b = e % NB;
        // inv(L)[a][b], lower: strictly lower part stored transposed in the diagonal block's upper triangle
        double v = 0.0;
        if (a < nb && b < nb) v = (b < a) ? A[(int64_t)(k0 + b) * n + k0 + a] : ((b == a) ? 1.0 / A[(int64_t)(k0 + a) * n + k0 + a] : 0.0);
        sLi[a][b] = v;
        // panel input rows from the MIRROR: A[i0 + r][k0 + c] = S[k0 + c][i0 + r]; here a = c (row of S), b = r (column)
        sAi[b][a] = (a < nb && i0 + b < lim) ? A[(int64_t)(k0 + a) * n + i0 + b] : 0.0;
        sAj[b][a] = (a < nb && j0 + b < lim) ? A[(int64_t)(k0 + a) * n + j0 + b] : 0.0;
    }
    __syncthreads();
    const int lane = tid & 63, wr = (tid >> 6) >> 1, wc = (tid >> 6) & 1;
    {
        // X = A inv(L)^T for both blocks, then the tile update, on the fp64 matrix pipe (chol_block.h: 8 MFMAs per wavefront and
        // product; the vector version -- 1 x 4 strips for X, 2 x 2 tiles for the update -- was bound by its LDS reads)
        const mqs::chol::double4v xi = mqs::chol::tile_quadrant_mfma(&sAi[0][0], &sLi[0][0], wr, wc, lane);
        const mqs::chol::double4v xj = (bi == bj) ? xi : mqs::chol::tile_quadrant_mfma(&sAj[0][0], &sLi[0][0], wr, wc, lane);
        const int c = mqs::chol::quadrant_col(wc, lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = mqs::chol::quadrant_row(wr, lane, v);
            sXi[r][c] = xi[v];
            sXj[r][c] = xj[v];
            if (bj == 0 && i0 + r < lim && c < nb) A[(int64_t)(i0 + r) * n + k0 + c] = xi[v];      // the panel: final entries of L
        }
    }
    __syncthreads();
    const mqs::chol::double4v acc = mqs::chol::tile_quadrant_mfma(&sXi[0][0], &sXj[0][0], wr, wc, lane);
    const bool next_diag = bi == 0 && bj == 0;                  // this tile is the next diagonal block
    double *sT = &sAi[0][0];                                    // reused: the updated tile for the factorisation
    __syncthreads();                                             // sAi is free
    {
        const int tc = mqs::chol::quadrant_col(wc, lane), c = j0 + tc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int tr = mqs::chol::quadrant_row(wr, lane, v), r = i0 + tr;
            if (next_diag) {
                // the whole next diagonal block goes to LDS for the factorisation below: updated inside the band's reach
                // (rows < lim), as it stands beyond it (a band narrower than a block: no fill there)
                if (r < n && c <= r) sT[tr * (NB + 1) + tc] = A[(int64_t)r * n + c] - ((r < lim && c < lim) ? acc[v] : 0.0);
            } else if (r < lim && c < lim && c <= r) {
                const double val = A[(int64_t)r * n + c] - acc[v];
                A[(int64_t)r * n + c] = val;
                if (bi != bj) A[(int64_t)c * n + r] = val;
            }
        }
    }
    if (!next_diag) return;
    __syncthreads();
    mqs::chol::factor_diag_block_from_lds_4w(sT, A, n, t0, bad, tid, sM);      // the next diagonal block (origin t0), four waves
}

// L y = b then L^T x = y; one workgroup, blocked by 256 rows with a block-level dot product per row
// Forward / backward substitution for a BANDED factor by one workgroup.  The unknowns live in LDS; the band is
// streamed through LDS in slabs of kSlab columns (forward) / rows (backward), so global-memory latency is paid once per
// slab.  Inside a slab wave 0 works column-oriented -- x_i = b_i / L_ii, then b_j -= L_ji x_i for the <= hb dependants,
// one per lane -- so no cross-lane reduction sits on the serial chain.  LDS: n + kSlab * (hb + kSlab) doubles.
constexpr int kSlab = 32;

__global__ __launch_bounds__(kBlock) void chol_solve_banded_kernel(const double *__restrict__ L, int n, int hb, double *__restrict__ x)
{
    extern __shared__ double sMem[];
    double *sXv = sMem;                    // [n]
    double *sB = sMem + n;                 // slab: kSlab x (hb + kSlab)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wslab = hb + kSlab;
    for (int i = tid; i < n; i += kBlock) sXv[i] = x[i];
    __syncthreads();
    // forward, columns i0 .. i0 + kSlab - 1: slab entry (r, c) = L[i0 + c][i0 + r], c = r .. r + hb
    for (int i0 = 0; i0 < n; i0 += kSlab) {
        for (int e = tid; e < kSlab * wslab; e += kBlock) {
            const int c = e / kSlab, r = e % kSlab;                 // consecutive threads: consecutive columns of one row of L
            const int j = i0 + c, i = i0 + r;
            double v = (i < n && j < n && j >= i) ? L[(int64_t)j * n + i] : 0.0;
            if (c == r && v != 0.0) v = 1.0 / v;                    // the diagonal is kept as its reciprocal
            sB[r * wslab + c] = v;
        }
        __syncthreads();
        if (wave == 0) {
            for (int r = 0; r < kSlab && i0 + r < n; ++r) {
                const int i = i0 + r;
                const double xi = sXv[i] * sB[r * wslab + r];
                for (int c = r + 1 + lane; c <= r + hb && i0 + c < n; c += 64) sXv[i0 + c] = fma(-sB[r * wslab + c], xi, sXv[i0 + c]);
                if (lane == 0) sXv[i] = xi;
                mqs_wave_lds_sync();
            }
        }
        __syncthreads();
    }
    // backward, rows i0 + kSlab - 1 .. i0: slab entry (r, c) = L[i0 + r][i0 - hb + c], c = r .. r + hb (the diagonal at c = r + hb)
    for (int i0 = ((n - 1) / kSlab) * kSlab; i0 >= 0; i0 -= kSlab) {
        const int c0 = i0 - hb;
        for (int e = tid; e < kSlab * wslab; e += kBlock) {
            const int r = e / wslab, c = e % wslab;
            const int i = i0 + r, j = c0 + c;
            double v = (i < n && j >= 0 && j <= i) ? L[(int64_t)i * n + j] : 0.0;
            if (j == i && v != 0.0) v = 1.0 / v;
            sB[e] = v;
        }
        __syncthreads();
        if (wave == 0) {
            for (int r = kSlab - 1; r >= 0; --r) {
                const int i = i0 + r;
                if (i >= n) continue;
                const double xi = sXv[i] * sB[r * wslab + hb + r];
                for (int c = r + lane; c < r + hb; c += 64) {          // columns i - hb .. i - 1
                    const int j = c0 + c;
                    if (j >= 0) sXv[j] = fma(-sB[r * wslab + c], xi, sXv[j]);
                }
                if (lane == 0) sXv[i] = xi;
                mqs_wave_lds_sync();
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kBlock) x[i] = sXv[i];
}

__global__ __launch_bounds__(kBlock) void chol_solve_kernel(const double *__restrict__ L, int n, double *__restrict__ x)
{
    __shared__ double sRed[kBlock];
    const int tid = threadIdx.x;
    // forward: x[i] = (b[i] - sum_{j<i} L[i][j] x[j]) / L[i][i]
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = tid; j < i; j += kBlock) s += L[(int64_t)i * n + j] * x[j];
        sRed[tid] = s;
        __syncthreads();
        for (int h = kBlock / 2; h >= 1; h >>= 1) {
            if (tid < h) sRed[tid] += sRed[tid + h];
            __syncthreads();
        }
        if (tid == 0) x[i] = (x[i] - sRed[0]) / L[(int64_t)i * n + i];
        __syncthreads();
    }
    // backward: x[i] = (y[i] - sum_{j>i} L[j][i] x[j]) / L[i][i]
    for (int i = n - 1; i >= 0; --i) {
        double s = 0.0;
        for (int j = i + 1 + tid; j < n; j += kBlock) s += L[(int64_t)j * n + i] * x[j];
        sRed[tid] = s;
        __syncthreads();
        for (int h = kBlock / 2; h >= 1; h >>= 1) {
            if (tid < h) sRed[tid] += sRed[tid + h];
            __syncthreads();
        }
        if (tid == 0) x[i] = (x[i] - sRed[0]) / L[(int64_t)i * n + i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Forward / backward substitution for the banded factor in PRODUCT FORM, one workgroup of 1024 threads: chol_diag_kernel
// left inv(L_kk) of every 32 x 32 diagonal block in that block's upper triangle, so a block of unknowns is a 32 x 32
// mat-vec (x_blk = inv(L_kk) b_blk) and its <= hb dependants a panel mat-vec -- no per-unknown serial chain at all (the
// column-oriented kernel above walks 2 n dependent steps; at n = 5286, hb = 101: 5 ms -> 1.3 ms).
// LDS: inv(L_kk) 32 x 33, the panel hb x 33, 32 x 33 partial sums, the unknowns n.
// (A whole-factorisation variant of this -- diagonal factor, panel and trailing update of every block column by the same
// persistent workgroup -- was built and measured: 11.3 ms against 10.4 ms for the launches it replaced.  One wavefront
// retires an instruction every 4-5 cycles at best, and with the 128-VGPR budget of a 1024-thread workgroup the 32 x 32
// factor has to live in LDS (~300 instructions per pivot step, 23 us per block) where the stand-alone 64-thread kernel
// keeps it in registers.)
// ---------------------------------------------------------------------------------------------
constexpr int kPT = 1024;
constexpr int kLdT = NB + 1;

// panel rows held in LDS at a time: the whole band when it fits beside the unknowns, else as many as do (the dependants are
// then processed in chunks)
static inline int banded_blocked_cap(int n, int hb)
{
    const int64_t budget = (int64_t)(150 * 1024) / (int64_t)sizeof(double) - ((int64_t)2 * NB * kLdT + 64 + n + 8);
    int64_t cap = budget / kLdT;
    if (cap > hb) cap = hb;
    return (int)(cap < 0 ? 0 : cap);
}

static inline size_t banded_blocked_lds_bytes(int n, int cap)
{
    return ((size_t)2 * NB * kLdT + 64 + (size_t)cap * kLdT + (size_t)n + 8) * sizeof(double);
}

__device__ __forceinline__ void load_inv_diag(const double *__restrict__ A, int n, int k0, int nb, double *sLi, int tid)
{
    // inv(L_kk) as chol_diag_kernel stores it: strictly lower part transposed in the block's upper triangle, diagonal = 1 / L_jj
    for (int e = tid; e < NB * NB; e += kPT) {
        const int j = e >> 5, k = e & 31;
        double v = 0.0;
        if (j < nb && k < nb) {
            if (k < j) v = A[(int64_t)(k0 + k) * n + k0 + j];
            else if (k == j) v = 1.0 / A[(int64_t)(k0 + j) * n + k0 + j];
        }
        sLi[j * kLdT + k] = v;
    }
}

// The whole band's panel fits in LDS beside the unknowns (the usual banded case): one load phase per block.
__global__ __launch_bounds__(kPT) void chol_solve_banded_blocked1_kernel(const double *__restrict__ A, int n, int hb,
                                                                         double *__restrict__ x)
{
    extern __shared__ double sMem[];
    double *sLi = sMem;                      // [NB][kLdT] inv(L_kk), lower
    double *sQ = sLi + NB * kLdT;            // [NB][kLdT] partial sums of the backward substitution
    double *sCol = sQ + NB * kLdT;           // [64]
    double *sP = sCol + 64;                  // [hb][kLdT] panel below the diagonal block
    double *sXv = sP + (size_t)hb * kLdT;    // [n] right-hand side -> solution
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += kPT) sXv[i] = x[i];

    // The factor entries of block k + 1 do not depend on the unknowns: they are fetched into registers (one inverse-block
    // entry and up to kPre panel entries per thread) while block k's three phases run, and stored to LDS at the top of the
    // next round -- the global-memory latency of a block step (about half of its 3.8 us) leaves the serial chain.
    constexpr int kPre = 4;                                  // panel entries per thread held in flight: hb <= 128
    const bool prefetch = hb * NB <= kPre * kPT;
    double pLi = 0.0, pP[kPre] = {0.0, 0.0, 0.0, 0.0};
    auto fetch = [&](int k0) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        const int j = tid >> 5, k = tid & 31;
        pLi = 0.0;
        if (j < nb && k <= j) pLi = A[(int64_t)(k0 + k) * n + k0 + j];     // k == j: the diagonal itself, inverted at commit
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = tid + u * kPT, r = e >> 5, c = e & 31;
            pP[u] = (e < m * NB && c < nb) ? A[(int64_t)(t0 + r) * n + k0 + c] : 0.0;
        }
    };
    auto commit = [&](int k0) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        int m = n - (k0 + nb);
        if (m > hb) m = hb;
        const int j = tid >> 5, k = tid & 31;
        sLi[j * kLdT + k] = (j < nb && k == j) ? 1.0 / pLi : pLi;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int e = tid + u * kPT;
            if (e < m * NB) sP[(e >> 5) * kLdT + (e & 31)] = pP[u];
        }
    };
    auto load_block = [&](int k0) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        load_inv_diag(A, n, k0, nb, sLi, tid);
        for (int e = tid; e < m * NB; e += kPT) {
            const int r = e >> 5, c = e & 31;
            sP[r * kLdT + c] = (c < nb) ? A[(int64_t)(t0 + r) * n + k0 + c] : 0.0;
        }
    };

    // ---- forward substitution L y = b, block by block ----
    if (prefetch) fetch(0);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        if (prefetch) commit(k0); else load_block(k0);
        __syncthreads();
        if (prefetch && k0 + NB < n) fetch(k0 + NB);
        if (wave == 0) {
            const int j = lane & 31;
            double s = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) s = fma(sLi[j * kLdT + k], (k < nb) ? sXv[k0 + k] : 0.0, s);
            mqs_wave_lds_sync();
            if (lane < nb) sXv[k0 + lane] = s;
        }
        __syncthreads();
        if (tid < m) {
            double s = 0.0;
#pragma unroll 8
            for (int c = 0; c < NB; ++c) s = fma(sP[tid * kLdT + c], (c < nb) ? sXv[k0 + c] : 0.0, s);
            sXv[t0 + tid] -= s;
        }
        __syncthreads();
    }
    // ---- backward substitution L^T x = y ----
    const int klast = ((n - 1) / NB) * NB;
    if (prefetch) fetch(klast);
    for (int k0 = klast; k0 >= 0; k0 -= NB) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        if (prefetch) commit(k0); else load_block(k0);
        __syncthreads();
        if (prefetch && k0 >= NB) fetch(k0 - NB);
        {
            // t_c = sum_r L[t0 + r][k0 + c] x[t0 + r]: 32 partial sums per column, then wave 0 combines them
            const int c = tid & 31, part = tid >> 5;
            double s = 0.0;
            for (int r = part; r < m; r += kPT / NB) s = fma(sP[r * kLdT + c], sXv[t0 + r], s);
            sQ[part * kLdT + c] = s;
        }
        __syncthreads();
        if (wave == 0) {
            const int c = lane & 31;
            double t = 0.0;
#pragma unroll 8
            for (int part = 0; part < kPT / NB; ++part) t += sQ[part * kLdT + c];
            const double yc = (c < nb) ? sXv[k0 + c] - t : 0.0;
            if (lane < NB) sCol[lane] = yc;
            mqs_wave_lds_sync();
            double s = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) s = fma(sLi[k * kLdT + c], sCol[k], s);     // inv(L)^T: zero for k < c
            if (lane < nb) sXv[k0 + lane] = s;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kPT) x[i] = sXv[i];
}

// The same with the panel taken `cap` rows at a time (dense factors, very wide bands).
__global__ __launch_bounds__(kPT) void chol_solve_banded_blocked_kernel(const double *__restrict__ A, int n, int hb, int cap,
                                                                         double *__restrict__ x)
{
    extern __shared__ double sMem[];
    double *sLi = sMem;                      // [NB][kLdT] inv(L_kk), lower
    double *sQ = sLi + NB * kLdT;            // [NB][kLdT] partial sums of the backward substitution
    double *sCol = sQ + NB * kLdT;           // [64]
    double *sP = sCol + 64;                  // [cap][kLdT] panel below the diagonal block (cap rows at a time)
    double *sXv = sP + (size_t)cap * kLdT;   // [n] right-hand side -> solution
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += kPT) sXv[i] = x[i];

    // ---- forward substitution L y = b, block by block ----
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        load_inv_diag(A, n, k0, nb, sLi, tid);
        for (int c0 = 0; c0 < m || c0 == 0; c0 += cap) {            // dependants, `cap` rows of the panel at a time
            const int mc = (m - c0) < cap ? (m - c0) : cap;
            for (int e = tid; e < mc * NB; e += kPT) {
                const int r = e >> 5, c = e & 31;
                sP[r * kLdT + c] = (c < nb) ? A[(int64_t)(t0 + c0 + r) * n + k0 + c] : 0.0;
            }
            __syncthreads();
            if (c0 == 0) {                                            // the block itself, once, beside the first chunk's load
                if (wave == 0) {
                    const int j = lane & 31;
                    double s = 0.0;
#pragma unroll 8
                    for (int k = 0; k < NB; ++k) s = fma(sLi[j * kLdT + k], (k < nb) ? sXv[k0 + k] : 0.0, s);
                    mqs_wave_lds_sync();
                    if (lane < nb) sXv[k0 + lane] = s;
                }
                __syncthreads();
            }
            for (int r = tid; r < mc; r += kPT) {
                double s = 0.0;
#pragma unroll 8
                for (int c = 0; c < NB; ++c) s = fma(sP[r * kLdT + c], (c < nb) ? sXv[k0 + c] : 0.0, s);
                sXv[t0 + c0 + r] -= s;
            }
            __syncthreads();
        }
    }
    // ---- backward substitution L^T x = y ----
    for (int k0 = ((n - 1) / NB) * NB; k0 >= 0; k0 -= NB) {
        const int nb = (n - k0) < NB ? (n - k0) : NB;
        const int t0 = k0 + nb;
        int m = n - t0;
        if (m > hb) m = hb;
        load_inv_diag(A, n, k0, nb, sLi, tid);
        // t_c = sum_r L[t0 + r][k0 + c] x[t0 + r]: 32 partial sums per column (over all chunks), then wave 0 combines them
        const int pc = tid & 31, part = tid >> 5;
        double ps = 0.0;
        for (int c0 = 0; c0 < m; c0 += cap) {
            const int mc = (m - c0) < cap ? (m - c0) : cap;
            if (c0 > 0) __syncthreads();                              // the previous chunk has been consumed
            for (int e = tid; e < mc * NB; e += kPT) {
                const int r = e >> 5, c = e & 31;
                sP[r * kLdT + c] = (c < nb) ? A[(int64_t)(t0 + c0 + r) * n + k0 + c] : 0.0;
            }
            __syncthreads();
            for (int r = part; r < mc; r += kPT / NB) ps = fma(sP[r * kLdT + pc], sXv[t0 + c0 + r], ps);
        }
        sQ[part * kLdT + pc] = ps;
        __syncthreads();
        if (wave == 0) {
            const int c = lane & 31;
            double t = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < kPT / NB; ++p2) t += sQ[p2 * kLdT + c];
            const double yc = (c < nb) ? sXv[k0 + c] - t : 0.0;
            if (lane < NB) sCol[lane] = yc;
            mqs_wave_lds_sync();
            double s = 0.0;
#pragma unroll 8
            for (int k = 0; k < NB; ++k) s = fma(sLi[k * kLdT + c], sCol[k], s);     // inv(L)^T: zero for k < c
            if (lane < nb) sXv[k0 + lane] = s;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kPT) x[i] = sXv[i];
}

// ---------------------------------------------------------------------------------------------
// Substitutions for a DENSE factor (pose graphs with loop closures: no band to exploit), n in the thousands.  One workgroup
// streaming the whole factor (112 MB at n = 5286) took 12 ms; here every block column is one launch over all the rows
// (forward) / columns (backward) it touches, right-looking in both directions:
//   forward   y_k = inv(L_kk) b_k, then b_r -= L[r][k-block] y_k for every row r below the block (one row per thread)
//   backward  x_k = inv(L_kk)^T y_k, then y_c -= L[k-block][c]^T x_k for every column c left of the block (one column per
//             thread: consecutive threads, consecutive addresses)
// Every workgroup forms the 32 unknowns of the block itself (a 32 x 32 mat-vec); workgroup 0 stores them -- into the OTHER
// vector (ytmp forward, x backward), because the block's right-hand side is still being read by the other workgroups of
// the launch.  One writer per entry per launch: the result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void dense_fwd_step_kernel(const double *__restrict__ A, int n, int k0, double *__restrict__ b,
                                                               double *__restrict__ ytmp)
{
    __shared__ double sLi[NB][NB + 1], sB[NB], sY[NB];
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int tid = threadIdx.x;
    mqs::chol::load_inv_diag_block(A, n, k0, nb, &sLi[0][0], tid, kBlock);
    if (tid < NB) sB[tid] = tid < nb ? b[k0 + tid] : 0.0;
    __syncthreads();
    if (tid < NB) {
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < NB; ++k) s = fma(sLi[tid][k], sB[k], s);
        sY[tid] = s;
        if (blockIdx.x == 0 && tid < nb) ytmp[k0 + tid] = s;
    }
    __syncthreads();
    const int r = k0 + nb + blockIdx.x * kBlock + tid;
    if (r < n) {
        const double *row = A + (int64_t)r * n + k0;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) s = fma((c < nb) ? row[c] : 0.0, sY[c], s);
        b[r] -= s;
    }
}

__global__ __launch_bounds__(kBlock) void dense_bwd_step_kernel(const double *__restrict__ A, int n, int k0, double *__restrict__ ytmp,
                                                               double *__restrict__ x)
{
    __shared__ double sLi[NB][NB + 1], sY[NB], sXk[NB];
    const int nb = (n - k0) < NB ? (n - k0) : NB;
    const int tid = threadIdx.x;
    mqs::chol::load_inv_diag_block(A, n, k0, nb, &sLi[0][0], tid, kBlock);
    if (tid < NB) sY[tid] = tid < nb ? ytmp[k0 + tid] : 0.0;
    __syncthreads();
    if (tid < NB) {
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < NB; ++k) s = fma(sLi[k][tid], sY[k], s);          // inv(L)^T: zero for k < column
        sXk[tid] = s;
        if (blockIdx.x == 0 && tid < nb) x[k0 + tid] = s;
    }
    __syncthreads();
    const int c = blockIdx.x * kBlock + tid;
    if (c < k0) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < NB; ++r)
            if (r < nb) s = fma(A[(int64_t)(k0 + r) * n + c], sXk[r], s);
        ytmp[c] -= s;
    }
}

// an n-vector of scratch per (device, stream) for the launches above, grown on demand and kept
double *dense_solve_scratch(size_t n, hipStream_t stream)
{
    struct Entry { double *p = nullptr; size_t n = 0; };
    static std::mutex mutex;
    static std::map<std::pair<int, void *>, Entry> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mutex);
    Entry &e = cache[{dev, (void *)stream}];
    if (e.n < n) {
        if (e.p) {
            if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;      // an earlier solve may still be using it
            (void)hipFree(e.p);
        }
        e.p = nullptr; e.n = 0;
        if (hipMalloc((void **)&e.p, n * sizeof(double)) != hipSuccess) return nullptr;
        e.n = n;
    }
    return e.p;
}

#ifndef MQS_SBA_BLOCKED_SUBST
#define MQS_SBA_BLOCKED_SUBST 1       // 0: the column-oriented substitution kernel (A/B builds)
#endif
#ifndef MQS_SBA_DENSE_MIN_N
#define MQS_SBA_DENSE_MIN_N 1536      // dense factors from this size on: substitutions as one launch per block column
#endif
constexpr int kDenseSubstMinN = MQS_SBA_DENSE_MIN_N;

}  // namespace

extern "C" {

int64_t mqs_sba_workspace_bytes(int64_t P, int64_t N, int64_t M)
{
    if (P < 0 || N < 0 || M < 0) return 0;
    const int64_t blocks = (N + kBlock / kLanesPerLandmark - 1) / (kBlock / kLanesPerLandmark) + 1;
    return (P * kCamStride + M * kRec + 2 * blocks + 16) * (int64_t)sizeof(double);
}

// S [(6P)^2], g [6P], info[4] = {cost, valid count, pose-prior cost, groups that violate the canonical ordering}.  S and g are
// overwritten.
int mqs_sba_linearize_grouped_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                                  const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                                  const int32_t *obs_pose, const double *obs_uv, int64_t M, const int64_t *pair_a,
                                  const int64_t *pair_b, int64_t Q, const int64_t *group_ptr, int64_t G,
                                  const double *prior_w, const double *prior_xyz, const int32_t *pose_prior_idx,
                                  const double *pose_prior_poses, const double *pose_prior_sigmas, int n_pose_prior,
                                  double lambda, double *S, double *g, double *info, void *workspace,
                                  int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(G >= 0 && (G == 0 || group_ptr), "group_ptr must not be null when G > 0");
    MQS_ARG_CHECK(P >= 1 && N >= 0 && M >= 0 && Q >= 0, "sizes must be non-negative, P >= 1");
    MQS_ARG_CHECK(6 * P <= 46000, "6P too large for the dense reduced system");
    MQS_ARG_CHECK(poses && pose_cam && calib && sigma && S && g && info && workspace, "pointers must not be null");
    MQS_ARG_CHECK(N == 0 || (points && obs_ptr), "points, obs_ptr must not be null");
    MQS_ARG_CHECK(M == 0 || (obs_pose && obs_uv), "obs arrays must not be null");
    MQS_ARG_CHECK(Q == 0 || (pair_a && pair_b), "pair list must not be null");
    MQS_ARG_CHECK(!prior_w || prior_xyz, "prior_xyz required with prior_w");
    MQS_ARG_CHECK(n_pose_prior == 0 || (pose_prior_idx && pose_prior_poses && pose_prior_sigmas), "pose prior arrays");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_workspace_bytes(P, N, M), "workspace too small (mqs_sba_workspace_bytes)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int n6 = (int)(6 * P);
    double *cams = static_cast<double *>(workspace);
    double *rec = cams + P * kCamStride;
    double *partials = rec + M * kRec;
    const int lm_blocks = (int)((N + kBlock / kLanesPerLandmark - 1) / (kBlock / kLanesPerLandmark));
    MQS_HIP_CHECK(hipMemsetAsync(S, 0, (size_t)n6 * n6 * sizeof(double), stream));
    MQS_HIP_CHECK(hipMemsetAsync(g, 0, (size_t)n6 * sizeof(double), stream));
    MQS_HIP_CHECK(hipMemsetAsync(info, 0, 4 * sizeof(double), stream));
    hipLaunchKernelGGL(stage_pose_cams_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, poses, pose_cam,
                       calib, sigma, (int)P, cams);
    if (N > 0) {
        hipLaunchKernelGGL(sparse_landmark_kernel, dim3(lm_blocks), dim3(kBlock), 0, stream, cams, points, obs_ptr,
                           obs_pose, obs_uv, prior_w, prior_xyz, N, lambda, rec, partials);
        hipLaunchKernelGGL(sum_cost_partials_kernel, dim3(1), dim3(kBlock), 0, stream, partials, lm_blocks, info);
    }
    if (Q > 0 && G > 0) {
        hipLaunchKernelGGL(sparse_check_groups_kernel, dim3((unsigned)((G + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                           obs_pose, pair_a, pair_b, group_ptr, G, P, info + 3);
        hipLaunchKernelGGL(sparse_pair_groups_kernel, dim3((unsigned)((G + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                           stream, rec, obs_pose, pair_a, pair_b, group_ptr, G, n6, S, g);
    }
    else if (Q > 0)
        hipLaunchKernelGGL(sparse_pairs_kernel, dim3((unsigned)((Q + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rec,
                           obs_pose, pair_a, pair_b, Q, n6, S, g);
    if (n_pose_prior > 0)
        hipLaunchKernelGGL(sparse_priors_kernel, dim3((n_pose_prior + 63) / 64), dim3(64), 0, stream, S, g, n6, poses,
                           pose_prior_idx, pose_prior_poses, pose_prior_sigmas, n_pose_prior, lambda, info + 2);
    if (!(Q > 0 && G > 0))     // the grouped pair stage writes both triangles itself
        hipLaunchKernelGGL(sparse_mirror_kernel, dim3((unsigned)(((int64_t)n6 * n6 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           stream, S, n6);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_sba_linearize_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                          const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                          const int32_t *obs_pose, const double *obs_uv, int64_t M, const int64_t *pair_a,
                          const int64_t *pair_b, int64_t Q, const double *prior_w, const double *prior_xyz,
                          const int32_t *pose_prior_idx, const double *pose_prior_poses,
                          const double *pose_prior_sigmas, int n_pose_prior, double lambda, double *S, double *g,
                          double *info, void *workspace, int64_t workspace_bytes, void *stream_)
{
    return mqs_sba_linearize_grouped_dev(poses, pose_cam, P, calib, sigma, points, N, obs_ptr, obs_pose, obs_uv, M, pair_a,
                                         pair_b, Q, nullptr, 0, prior_w, prior_xyz, pose_prior_idx, pose_prior_poses,
                                         pose_prior_sigmas, n_pose_prior, lambda, S, g, info, workspace, workspace_bytes,
                                         stream_);
}

// In place: S is replaced by its Cholesky factor (after lambda*diag(S) damping), x (= g on entry) by the
// solution; poses_out (may be NULL) = retract(poses, x).  bad[0] (int, device) is set when S is not
// positive definite.
int mqs_sba_solve_banded_dev(double *S, double *x, int64_t P, int64_t half_bandwidth, double lambda, const double *poses,
                             double *poses_out, int *bad, void *stream_)
{
    MQS_ARG_CHECK(P >= 1 && S && x && bad, "arguments");
    MQS_ARG_CHECK(half_bandwidth >= 0, "half_bandwidth >= 0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int n = (int)(6 * P);
    const int hb = half_bandwidth < n ? (int)half_bandwidth : n;
    MQS_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), stream));
    if (lambda != 0.0) hipLaunchKernelGGL(sparse_damp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, n, lambda);
    const bool banded = 3 * (int64_t)hb < n;            // the band is worth exploiting
    bool nd_done = false;
    int rc_nd = MQS_OK;
    if (banded && (rc_nd = mqs_chol_nd_solve(S, x, n, hb, bad, stream, &nd_done), rc_nd != MQS_OK || nd_done)) {
        // the band cut into independent chunks (chol_nd.hip): ~40 dependent launches instead of one per block column
        if (rc_nd != MQS_OK) return rc_nd;
    } else {
        if (n > 1 && hb > 0) {                      // input contract: the lower triangle (see sparse_mirror_lower_kernel)
            const int brows = (n + 31) / 32, boffs = (hb + 31) / 32 + 1;
            hipLaunchKernelGGL(sparse_mirror_lower_kernel, dim3(brows, boffs < brows ? boffs : brows), dim3(256), 0, stream, S, n);
        }
        // right-looking blocked Cholesky; rows further than hb below a block column are zero there and stay zero
        // (no fill outside the band), so the panel and the trailing update stop hb rows below it
        // one launch per block column: the first diagonal block, then panel + update + next diagonal block fused (see
        // chol_step_kernel).  Both triangles are valid here: sparse_mirror_lower_kernel above.
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, stream, S, n, 0, bad);
        for (int k0 = 0; k0 < n; k0 += NB) {
            const int nb = (n - k0) < NB ? (n - k0) : NB;
            int rem = n - k0 - nb;
            if (rem > hb) rem = hb;
            if (rem > 0) {
                const int tiles = (rem + NB - 1) / NB;
                hipLaunchKernelGGL(chol_step_kernel, dim3(tiles, tiles), dim3(kBlock), 0, stream, S, n, k0, hb, bad);
            }
        }
        const size_t lds = ((size_t)n + (size_t)kSlab * (hb + kSlab)) * 8;
        const int hbs = banded ? hb : n;                     // the dense factor is a band of full width
        const int cap = banded_blocked_cap(n, hbs);
        if (!banded && n >= kDenseSubstMinN) {
            double *ytmp = dense_solve_scratch((size_t)n, stream);
            MQS_ARG_CHECK(ytmp != nullptr, "scratch vector for the dense substitutions could not be allocated");
            for (int k0 = 0; k0 < n; k0 += NB) {
                const int nb = (n - k0) < NB ? (n - k0) : NB;
                const int rows = n - k0 - nb;
                hipLaunchKernelGGL(dense_fwd_step_kernel, dim3(rows > 0 ? (rows + kBlock - 1) / kBlock : 1), dim3(kBlock), 0, stream, S, n,
                                   k0, x, ytmp);
            }
            for (int k0 = ((n - 1) / NB) * NB; k0 >= 0; k0 -= NB)
                hipLaunchKernelGGL(dense_bwd_step_kernel, dim3(k0 > 0 ? (k0 + kBlock - 1) / kBlock : 1), dim3(kBlock), 0, stream, S, n, k0,
                                   ytmp, x);
        }
        else if (MQS_SBA_BLOCKED_SUBST && cap >= (hbs < 64 ? hbs : 64)) {
            static mqs_lds_opt_in opt_b, opt_b1;            // per device
            MQS_HIP_CHECK(mqs_lds_opt_in_once(opt_b, reinterpret_cast<const void *>(chol_solve_banded_blocked_kernel), 150 * 1024));
            if (cap >= hbs && hbs > 0) {
                MQS_HIP_CHECK(mqs_lds_opt_in_once(opt_b1, reinterpret_cast<const void *>(chol_solve_banded_blocked1_kernel), 150 * 1024));
                hipLaunchKernelGGL(chol_solve_banded_blocked1_kernel, dim3(1), dim3(kPT), banded_blocked_lds_bytes(n, hbs), stream, S,
                                   n, hbs, x);
            } else {
                hipLaunchKernelGGL(chol_solve_banded_blocked_kernel, dim3(1), dim3(kPT), banded_blocked_lds_bytes(n, cap > 0 ? cap : 1),
                                   stream, S, n, hbs, cap > 0 ? cap : 1, x);
            }
        }
        else if (banded && lds <= 150 * 1024) {
            static mqs_lds_opt_in opt;                       // per device
            MQS_HIP_CHECK(mqs_lds_opt_in_once(opt, reinterpret_cast<const void *>(chol_solve_banded_kernel), 150 * 1024));
            hipLaunchKernelGGL(chol_solve_banded_kernel, dim3(1), dim3(kBlock), lds, stream, S, n, hb, x);
        }
        else
            hipLaunchKernelGGL(chol_solve_kernel, dim3(1), dim3(kBlock), 0, stream, S, n, x);
    }
    if (poses_out)
        hipLaunchKernelGGL(sparse_retract_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, poses, x, (int)P,
                           poses_out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_sba_solve_dev(double *S, double *x, int64_t P, double lambda, const double *poses, double *poses_out, int *bad,
                      void *stream_)
{
    return mqs_sba_solve_banded_dev(S, x, P, 6 * P, lambda, poses, poses_out, bad, stream_);
}

int mqs_sba_backsub_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                        const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                        const int32_t *obs_pose, const double *obs_uv, int64_t M, const double *prior_w,
                        const double *prior_xyz, double lambda, const double *dpose, double *points_out,
                        void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(P >= 1 && N >= 0 && M >= 0, "sizes");
    MQS_ARG_CHECK(poses && pose_cam && calib && sigma && dpose && workspace, "pointers must not be null");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_workspace_bytes(P, N, M), "workspace too small");
    if (N == 0) return MQS_OK;
    MQS_ARG_CHECK(points && points_out && obs_ptr && (M == 0 || (obs_pose && obs_uv)), "landmark arrays");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *cams = static_cast<double *>(workspace);
    hipLaunchKernelGGL(stage_pose_cams_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, poses, pose_cam,
                       calib, sigma, (int)P, cams);
    hipLaunchKernelGGL(sparse_backsub_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, cams,
                       points, obs_ptr, obs_pose, obs_uv, prior_w, prior_xyz, N, lambda, dpose, points_out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_sba_cost_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib, const double *sigma,
                     const double *points, int64_t N, const int64_t *obs_ptr, const int32_t *obs_pose,
                     const double *obs_uv, int64_t M, const double *prior_w, const double *prior_xyz, double *out,
                     void *workspace, int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(P >= 1 && N >= 0 && M >= 0 && out && workspace, "arguments");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_workspace_bytes(P, N, M), "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *cams = static_cast<double *>(workspace);
    double *partials = cams + P * kCamStride + M * kRec;
    MQS_HIP_CHECK(hipMemsetAsync(out, 0, 2 * sizeof(double), stream));
    if (N == 0) return MQS_OK;
    const int lm_blocks = (int)((N + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(stage_pose_cams_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, poses, pose_cam,
                       calib, sigma, (int)P, cams);
    hipLaunchKernelGGL(sparse_cost_kernel, dim3(lm_blocks), dim3(kBlock), 0, stream, cams, points, obs_ptr, obs_pose,
                       obs_uv, prior_w, prior_xyz, N, partials);
    hipLaunchKernelGGL(sum_cost_partials_kernel, dim3(1), dim3(kBlock), 0, stream, partials, lm_blocks, out);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

int mqs_sba_worst_residual_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib, const double *sigma,
                               const double *points, int64_t N, const int64_t *obs_ptr, const int32_t *obs_pose,
                               const double *obs_uv, int64_t M, double *worst, double *min_depth, void *workspace,
                               int64_t workspace_bytes, void *stream_)
{
    MQS_ARG_CHECK(P >= 1 && N >= 0 && M >= 0 && workspace, "arguments");
    MQS_ARG_CHECK(workspace_bytes >= mqs_sba_workspace_bytes(P, N, M), "workspace too small");
    if (N == 0) return MQS_OK;
    MQS_ARG_CHECK(poses && pose_cam && calib && sigma && points && obs_ptr && worst && (M == 0 || (obs_pose && obs_uv)), "pointers must not be null");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *cams = static_cast<double *>(workspace);
    hipLaunchKernelGGL(stage_pose_cams_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, poses, pose_cam,
                       calib, sigma, (int)P, cams);
    hipLaunchKernelGGL(sparse_worst_residual_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, cams,
                       points, obs_ptr, obs_pose, obs_uv, pose_cam, sigma, N, worst, min_depth);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// Adds the odometry BetweenFactors to a linearised (mirrored) system; S == NULL: only cost[0] += their cost.
int mqs_sba_between_dev(const double *poses, int64_t P, const int32_t *odo_from, const int32_t *odo_to,
                        const double *odo_meas, const double *odo_sigmas, int64_t n_odo, double *S, double *g,
                        double *cost, void *stream_)
{
    MQS_ARG_CHECK(P >= 1 && n_odo >= 0 && n_odo <= 0x7fffffff, "P >= 1, n_odo >= 0");
    if (n_odo == 0) return MQS_OK;
    MQS_ARG_CHECK(poses && odo_from && odo_to && odo_meas && odo_sigmas && cost, "pointers must not be null");
    MQS_ARG_CHECK((S == nullptr) == (g == nullptr), "S and g go together");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(sparse_between_kernel, dim3((unsigned)((n_odo + 63) / 64)), dim3(64), 0, stream, S, g, (int)(6 * P),
                       poses, odo_from, odo_to, odo_meas, odo_sigmas, (int)n_odo, cost);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

// ---------------------------------------------------------------------------------------------
// Levenberg-Marquardt over the sparse problem, driven from here (round 4): what `LevenbergMarquardtOptimizer(graph,
// initial).optimize()` is to bundle_adjust.cpp:323-324.  The schedule is GTSAM 3.2.1's default (lambda 1e-5, factor 10, upper
// bound 1e5, a failed trial multiplies lambda and tries again from the same linearisation point, stop on absolute / relative
// cost decrease) -- the loop `sparse_ba.SparseBundleAdjuster.optimize_host_loop` runs from Python over the same entry points;
// here a trial is ONE host synchronisation (the trial's landmark cost, pose-prior cost, odometry cost and the factorisation's
// `bad` word come back together through a pinned buffer) and no interpreter between the launches: at the sizes the SLAM loop
// adjusts behind a keyframe (<= 60 poses, 10-15 k observations) the kernels are microseconds and the Python loop's four
// synchronisations and dozen ctypes calls per trial were the cost.
// ---------------------------------------------------------------------------------------------
namespace {

// the driver's words: the last 16 doubles of the workspace (mqs_sba_workspace_bytes leaves them free), read back in one copy
struct LmSums {
    double lin_info[4];          // of the last linearisation: {cost, valid, pose-prior cost, pair groups out of canonical order}
    double lin_odo_cost;         // (the odometry factors' cost at the linearisation point: not used)
    double spare[3];
    double landmark_cost, valid, prior_cost, odo_cost;          // of the estimate last evaluated (lm_enqueue_cost)
    int bad, pad;                // the factorisation's verdict
    double spare2[3];
};
static_assert(sizeof(LmSums) == 16 * sizeof(double), "the driver's words are the workspace's spare 16 doubles");

// the cost of the estimate (poses, points) into `sums` (device: LmSums::landmark_cost ...): three small launches
int lm_enqueue_cost(const mqs_sba_problem_dev *p, const double *poses, const double *points, double *sums, hipStream_t stream)
{
    int rc = mqs_sba_cost_dev(poses, p->pose_cam, p->P, p->calib, p->sigma, points, p->N, p->obs_ptr, p->obs_pose, p->obs_uv, p->M,
                              p->prior_w, p->prior_xyz, sums, p->workspace, p->workspace_bytes, stream);      // zeroes sums[0..1]
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemsetAsync(sums + 2, 0, 2 * sizeof(double), stream));
    if (p->n_pose_prior > 0)
        hipLaunchKernelGGL(sparse_priors_kernel, dim3((p->n_pose_prior + 63) / 64), dim3(64), 0, stream, nullptr, nullptr, (int)(6 * p->P),
                           poses, p->pose_prior_idx, p->pose_prior_poses, p->pose_prior_sigmas, p->n_pose_prior, 0.0, sums + 2);
    if (p->n_odo > 0)
        hipLaunchKernelGGL(sparse_between_kernel, dim3((unsigned)((p->n_odo + 63) / 64)), dim3(64), 0, stream, nullptr, nullptr,
                           (int)(6 * p->P), poses, p->odo_from, p->odo_to, p->odo_meas, p->odo_sigmas, (int)p->n_odo, sums + 3);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

LmSums *lm_pinned()
{
    static thread_local LmSums *h = nullptr;                // one per calling thread, for the life of the process
    if (!h && hipHostMalloc(reinterpret_cast<void **>(&h), sizeof(LmSums), hipHostMallocPortable) != hipSuccess) h = nullptr;
    return h;
}

}  // namespace

int64_t mqs_sba_lm_workspace_bytes(int64_t P, int64_t N, int64_t M) { return mqs_sba_workspace_bytes(P, N, M); }

int mqs_sba_optimize_lm_dev(const mqs_sba_problem_dev *p, const mqs_sba_lm_params *lm, double *cost_history, int32_t history_cap,
                            int32_t *n_history, void *stream_)
{
    MQS_ARG_CHECK(p && lm && cost_history && n_history && history_cap >= 1, "arguments must not be null");
    MQS_ARG_CHECK(p->P >= 1 && p->N >= 0 && p->M >= 0 && p->Q >= 0 && p->G >= 0 && p->n_odo >= 0 && p->n_pose_prior >= 0, "sizes");
    MQS_ARG_CHECK(p->poses && p->poses_new && p->S && p->g && p->workspace, "poses, poses_new, S, g, workspace must not be null");
    MQS_ARG_CHECK(p->N == 0 || (p->points && p->points_new), "points, points_new must not be null");
    MQS_ARG_CHECK(p->workspace_bytes >= mqs_sba_workspace_bytes(p->P, p->N, p->M), "workspace too small (mqs_sba_lm_workspace_bytes)");
    MQS_ARG_CHECK(lm->lambda_factor > 1.0 && lm->lambda_initial > 0.0 && lm->max_iterations >= 0, "LM parameters");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    LmSums *host = lm_pinned();
    MQS_ARG_CHECK(host != nullptr, "pinned host buffer for the trial results could not be allocated");
    double *blk = static_cast<double *>(p->workspace) + mqs_sba_workspace_bytes(p->P, p->N, p->M) / (int64_t)sizeof(double) - 16;
    double *sums = blk + 8;
    int *bad = reinterpret_cast<int *>(blk + 12);
    const double sgn = lm->damping == MQS_SBA_DAMPING_MARQUARDT ? 1.0 : -1.0;
    auto total = [&](bool with_bad, double &out, bool &ok) -> int {
        if (!with_bad) MQS_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), stream));
        MQS_HIP_CHECK(hipMemcpyAsync(host, blk, sizeof(LmSums), hipMemcpyDeviceToHost, stream));
        MQS_HIP_CHECK(hipStreamSynchronize(stream));
        ok = host->bad == 0;
        out = host->landmark_cost;                          // the order the host loop adds them in
        if (p->n_pose_prior > 0) out += host->prior_cost;
        if (p->n_odo > 0) out += host->odo_cost;
        return MQS_OK;
    };
    int rc = lm_enqueue_cost(p, p->poses, p->points, sums, stream);
    if (rc != MQS_OK) return rc;
    double cur = 0.0;
    bool ok = true;
    if ((rc = total(false, cur, ok)) != MQS_OK) return rc;
    int nh = 0;
    cost_history[nh++] = cur;
    double lam = lm->lambda_initial;
    for (int it = 0; it < lm->max_iterations; ++it) {
        bool improved = false;
        double fresh = 0.0;
        while (lam <= lm->lambda_upper) {
            const double l = sgn * lam;
            rc = mqs_sba_linearize_grouped_dev(p->poses, p->pose_cam, p->P, p->calib, p->sigma, p->points, p->N, p->obs_ptr, p->obs_pose,
                                               p->obs_uv, p->M, p->pair_a, p->pair_b, p->Q, p->group_ptr, p->G, p->prior_w, p->prior_xyz,
                                               p->pose_prior_idx, p->pose_prior_poses, p->pose_prior_sigmas, p->n_pose_prior, l, p->S, p->g,
                                               blk, p->workspace, p->workspace_bytes, stream);
            if (rc != MQS_OK) return rc;
            if (p->n_odo > 0) {
                rc = mqs_sba_between_dev(p->poses, p->P, p->odo_from, p->odo_to, p->odo_meas, p->odo_sigmas, p->n_odo, p->S, p->g, blk + 4, stream);
                if (rc != MQS_OK) return rc;
            }
            rc = mqs_sba_solve_banded_dev(p->S, p->g, p->P, p->half_bandwidth, l, p->poses, p->poses_new, bad, stream);
            if (rc != MQS_OK) return rc;
            rc = mqs_sba_backsub_dev(p->poses, p->pose_cam, p->P, p->calib, p->sigma, p->points, p->N, p->obs_ptr, p->obs_pose, p->obs_uv, p->M,
                                     p->prior_w, p->prior_xyz, l, p->g, p->points_new, p->workspace, p->workspace_bytes, stream);
            if (rc != MQS_OK) return rc;
            rc = lm_enqueue_cost(p, p->poses_new, p->N > 0 ? p->points_new : p->points, sums, stream);
            if (rc != MQS_OK) return rc;
            if ((rc = total(true, fresh, ok)) != MQS_OK) return rc;
            MQS_ARG_CHECK(host->lin_info[3] == 0.0, "pair groups are not in canonical order (mqs_sba_group_pairs_dev builds them)");
            if (!ok) fresh = HUGE_VAL;
            if (fresh <= cur) {
                MQS_HIP_CHECK(hipMemcpyAsync(p->poses, p->poses_new, (size_t)p->P * 12 * sizeof(double), hipMemcpyDeviceToDevice, stream));
                if (p->N > 0)
                    MQS_HIP_CHECK(hipMemcpyAsync(p->points, p->points_new, (size_t)p->N * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
                lam = lam / lm->lambda_factor;
                if (lam < 1e-20) lam = 1e-20;
                improved = true;
                break;
            }
            lam *= lm->lambda_factor;
        }
        if (!improved) break;
        if (nh < history_cap) cost_history[nh++] = fresh;
        const double dec = fabs(cur - fresh);
        const bool done = dec < lm->abs_tol || dec / (cur > 1e-300 ? cur : 1e-300) < lm->rel_tol;
        cur = fresh;
        if (done) break;
    }
    *n_history = nh;
    return MQS_OK;
}

}  // extern "C"
