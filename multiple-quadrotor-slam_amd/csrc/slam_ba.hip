// Bundle adjustment INSIDE the device-resident frame loop (BASELINE configs[4]: detect -> track -> triangulate -> BA per keyframe)
// as ONE persistent launch on the loop's resident state -- the counterpart of what the reference does with its recording
// afterwards: `performBundleAdjustment`, Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:190-330 (iSAM_version 0: the whole
// graph, one batch LevenbergMarquardtOptimizer::optimize(), :323-324), fed by slam2.py's BundleAdjustmentInfoContainer
// (slam2.py:519-522, 634-641, 681-687, 1167-1169).
//
// Round 4 ran this from Python: the log came to the host, numpy built the CSR problem, a fresh SparseBundleAdjuster uploaded it,
// every Levenberg-Marquardt trial was ~40 launches and one synchronisation (1.6 ms per adjustment on a 360-unknown system whose
// kernels are microseconds).  Here the whole adjustment -- problem build, screens, every LM trial of every pass, write-back -- is
// one kernel of G workgroups (one per CU) that meet at grid-wide barriers:
//
//   build     log entry -> (landmark, pose) cell of a dense table T[pose][landmark] = log index (atomicMax: the later of two
//             observations of one cell wins, deterministically); per landmark: observation count, first / last pose; per pose: its
//             observations compacted in landmark order.  No sort, no pair list: the landmark phases walk a column of T, the
//             reduced-system phase walks a pose's list and looks the other pose's cell up.
//   trial     A  per landmark (8 lanes): H_ll, its Cholesky factor, one 14-double record per observation (ba_sparse.hip's)
//             B  per pose pair (one wavefront): the 6 x 6 block of the reduced camera system from the records of the landmarks
//                both poses see, pose prior, odometry factors, damping; the right-hand side rides along as one more matrix row
//             C  blocked Cholesky over 32 x 32 tiles on the fp64 matrix pipe, one grid barrier per block column; the extra row
//                comes out as the forward-substituted right-hand side
//             D  backward substitution (workgroup 0)
//             E  per landmark: back-substitution, then the cost of the trial estimate
//             F  every workgroup adds the cost pieces in the same order and takes the same accept / reject decision
//   screens   worst residual / smallest depth per landmark, the median depth by rank counting, all inside the launch
//
// Round 6: the poses of the problem are a SELECTION of the accepted frames (mqs_slam_bundle_adjust_window: typically every keyframe so
// far + every frame since the third keyframe from the end), read once per workgroup from the pinned report block into LDS; a log entry's
// frame maps to its problem pose by binary search there; frames left out keep their pose relative to the selected pose in front of them.
// The reference's committed runs are 376 and 881 poses long (Work/SLAM/datasets/ICL_NUIM/*/traj_out.cam0-slam2.txt): the 256 cameras the LDS
// holds bound ONE problem now, not the run.  And the residual screen also looks behind every `screen_iters` iterations (legs).
//
// Inter-workgroup visibility: every byte one workgroup writes and another reads inside the launch is stored write-through and
// loaded past the vector L1 (relaxed agent-scope atomics: global_store / global_load ... sc1), every storing wave drains its
// stores (s_waitcnt vmcnt(0)) in front of the workgroup barrier behind which one lane adds to the barrier counter, and the
// counter is polled with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility: the all-sc1 form, one workgroup per CU --
// which the 112 KB of LDS per workgroup enforces).  No fence anywhere.  Every wait is bounded (2 s) and a wait that gives up
// ends the launch with status 1.
#include "mqs_common.h"
#include "so3_math.h"
#include "ba_math.h"
#include "chol_block.h"
#include "wave_reduce.h"
#include "slam_state.h"
#include <new>
#include <mutex>

namespace {

using namespace mqs::ba;
using namespace mqs::slamst;

constexpr int kT = 256;                          // threads per workgroup
constexpr int kMaxPoses = MQS_SLAM_BA_MAX_POSES; // cameras staged in LDS (twice)
constexpr int kRec = 14;                         // doubles per observation record (ba_sparse.hip)
constexpr int kListCap = 2048;                   // observations of one pose
constexpr int kERec = 96;                        // doubles per odometry-edge record
constexpr int kLanes = 32;                       // lanes per landmark in the landmark phases: half a wavefront
constexpr int TB = mqs::chol::NB, TLD = mqs::chol::kLd;
constexpr int kCtr = 4096;                       // one-shot counters (a fresh one per use)
constexpr int kMaxEdges = 4096;
constexpr int kStamps = 2048;
constexpr int kMaxGroups = 256;                    // workgroups of a launch at most (partials)
constexpr long long kSpinTicks = 200000000ll;    // 2 s of the 100 MHz wall clock
constexpr double kAugDiag = 1e200;               // diagonal entry of the right-hand side's row in the augmented matrix

template <class T> __device__ __forceinline__ T ldg(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void stg(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// what the kernel works on (device pointers into the adjuster's arena)
struct BaDev {
    // resident across adjustments
    double *pose0;                     // [12] camera-to-world pose of the first frame as mqs_slam_start estimated it: where its prior sits
    double *objp0;                     // [n0][3] the start-up landmarks as given: where their priors sit
    int32_t *bad;                      // [max_landmarks] 1: retired
    int32_t *odo_from, *odo_to;        // [kMaxEdges]
    double *odo_meas;                  // [kMaxEdges][12] measured relative pose (R row-major, t)
    int32_t *e_in, *e_out;             // [traj_cap] the edge that ends / starts at a trajectory pose, or -1
    // per adjustment
    int32_t *lm_res;                   // [log_cap] landmark of a log entry, -1: not part of the problem
    int32_t *T;                        // [P][N]
    int32_t *per_lm, *use, *pmin, *pmax, *cand;   // [N_cap]
    int32_t *out_cnt;                  // [N_cap] a landmark's observations at accepted frames that are NOT poses of this problem (a selection)
    double *traj_old;                  // [kMaxPoses][12] the trajectory rows of the problem's poses as this launch found them (the carry of the rest)
    const int32_t *sel_host;           // pinned host memory: the selection (trajectory index of problem pose k), read once by every workgroup
    long long *plist;                  // [kMaxPoses][kListCap] a pose's observations in landmark order: log index | landmark << 32
    int32_t *pcount, *pl_lo, *pl_hi;   // [kMaxPoses] x 3
    long long *hits;                   // [hits_cap] per pose pair (ja < jb): the observations of a common landmark, log index in ja | in jb << 32
    int32_t *blk_off, *blk_cnt;        // [kMaxPoses (kMaxPoses + 1) / 2] where a pose pair's hits lie
    double *rec;                       // [kRec][rec_stride]: field q of log entry e at rec[q * rec_stride + e] -- the entries a wavefront gathers
                                       // (one pose's observations in landmark order) are nearly consecutive log indices, so a field's
                                       // 64 loads fall into a few lines; entry-major, every lane touched a line of its own per field
    long long rec_stride;
    double *poses_a, *poses_b, *poses_init;       // [kMaxPoses][12]
    double *pts_a, *pts_b, *pts_init;  // [N_cap][3]
    double *worst, *zmin;              // [N_cap]
    double *S, *Lp;                    // tiles [ntile_cap][32 x 32]: the system (factored in place) / the finished panel blocks
    double *dpose, *ysol;              // [6 kMaxPoses + 64] the step; the right-hand side as the back-substitution's super-blocks leave it
    double *erec, *prec;               // [kMaxEdges][kERec], [16]
    double *partials;                  // [G][4]
    double *med;                       // [4]
    uint32_t *sync;                    // [0] barrier counter, [1] abort word: one of two 64-byte blocks, alternating by launch
    uint32_t *sync_next;               // the other block: zeroed by this launch for the next one (no fill launch in front of a launch)
    double *poses_host;                // pinned host memory the write-back phase leaves the adjusted poses in as well ([R | t], 12 each); cap below
    int poses_host_cap;
    int32_t *ctr;                      // [kCtr]
    double *report;                    // [MQS_SLAM_BA_REPORT], in pinned host memory (NaN-filled by the host in front of a launch)
    int N_cap, P_cap, ntile_cap, n0;
    long long hits_cap;
    long long *stamps;                 // null, or [kStamps][2] (phase id, 100 MHz wall clock) written by workgroup 0 (mqs_debug_slam_ba_stamps)
};

struct BaParams {
    int P, G, key_pose;                              // P: poses of the PROBLEM (<= kMaxPoses)
    // the selection (mqs_slam_bundle_adjust_window): problem pose k = trajectory pose sel[k] (ascending, the last = the last accepted
    // frame); identity: sel[k] = k, P = P_all -- the plain call
    int P_all, sel0, identity;                       // accepted frames; sel[0]
    int anchor2;                                     // trajectory index of a second pose held by a prior at its current value, or -1
    int carry;                                       // the accepted frames behind sel[0] that are not poses of the problem keep their pose RELATIVE to the nearest problem pose in front of them
    double out_prior_w;                              // 1 / sigma^2 of the prior on a landmark that frames outside the problem have seen, at its current value; 0: none
    int max_iterations, min_observations, max_passes, damping;
    int screen_iters;                                // 0: the residual screen only behind a finished adjustment; K: also behind every K iterations of one (legs)
    int add_edge, edge_from, edge_to, n_odo;         // n_odo: edges AFTER this call's has been appended
    double outlier_px, gross_px, border, min_depth_ratio;
    double prior_w, isigma_px, sigma_px;
    double pose_w[6], odo_w[6];                      // 1 / sigma^2
    double lam0, lam_factor, lam_upper, abs_tol, rel_tol;
    int W, H;
};

extern __shared__ double g_sm[];                      // dynamic LDS of slam_ba_kernel

struct Cx {                                          // per-thread view of the launch
    BaDev b; SlamDev d; BaParams p;
    int tid, lane, wave, wg, G;
    int P, N, nlog, n, nt;                           // n = 6 P unknowns; nt tiles per dimension of the augmented (n + 1) matrix
    int n0;                                          // start-up landmarks that are the gauge of THIS problem (0 when frame 0 is not one of its poses)
    int a2;                                          // problem index of the second anchored pose, or -1
    uint32_t epoch;
    int ctr_next, n_stamp;
};

// LDS layout: the cameras at the linearisation point, the trial's cameras (the Cholesky tiles and the back-substitution's vector
// use that space while no trial estimate exists), the step, reduction scratch
__device__ __forceinline__ double *lds_cam() { return g_sm; }
__device__ __forceinline__ double *lds_cam_new() { return g_sm + kMaxPoses * kCamStride; }
__device__ __forceinline__ double *lds_step() { return g_sm + 2 * kMaxPoses * kCamStride; }
__device__ __forceinline__ double *lds_red() { return g_sm + 2 * kMaxPoses * kCamStride + 6 * kMaxPoses + 32; }
__device__ __forceinline__ int *lds_flag() { return reinterpret_cast<int *>(g_sm + 2 * kMaxPoses * kCamStride + 6 * kMaxPoses + 32 + 32); }
__device__ __forceinline__ int *lds_sel() { return reinterpret_cast<int *>(g_sm + 2 * kMaxPoses * kCamStride + 6 * kMaxPoses + 32 + 64); }      // [kMaxPoses] the selection
constexpr size_t kLdsBytes = ((size_t)2 * kMaxPoses * kCamStride + 6 * kMaxPoses + 32 + 64) * sizeof(double) + kMaxPoses * sizeof(int);

// problem pose k -> trajectory pose; trajectory pose j -> problem pose or -1 (binary search of the ascending selection in LDS);
// the last problem pose at or in front of trajectory pose j (j >= sel[0])
__device__ __forceinline__ int pose_of(const Cx &c, int k) { return c.p.identity ? k : lds_sel()[k]; }
__device__ __forceinline__ int sel_lower(const Cx &c, int j)
{
    int lo = 0, hi = c.P;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lds_sel()[mid] < j) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int map_pose(const Cx &c, int j)
{
    if (c.p.identity) return (j >= 0 && j < c.P) ? j : -1;
    if (j < 0) return -1;
    const int lo = sel_lower(c, j);
    return (lo < c.P && lds_sel()[lo] == j) ? lo : -1;
}
__device__ __forceinline__ int anchor_of(const Cx &c, int j)
{
    const int lo = sel_lower(c, j);
    return (lo < c.P && lds_sel()[lo] == j) ? lo : lo - 1;
}

__device__ __forceinline__ int tix(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
// square root for the triangular-index decodes (x >= 1; the callers correct the last unit): x * rsqrt(x) by Newton steps
__device__ __forceinline__ double tri_root(double x) { return x * mqs::rsqrt_d(x); }

// every workgroup of the launch has arrived (and everything it stored with stg before is visible to ldg afterwards); false: a
// wait gave up somewhere -- the caller returns
__device__ __forceinline__ bool grid_barrier(Cx &c)
{
    mqs_stores_landed();
    __syncthreads();
    c.epoch += 1;
    if (c.tid == 0) {
        atomicAdd(c.b.sync, 1u);
        const uint32_t target = c.epoch * (uint32_t)c.G;
        // the poll is ONE sc1 load per round; the abort word and the clock (a scalar-memory round trip of its own) only every 64th
        long long t0 = 0;
        int ok = 1;
        for (unsigned spins = 0; ldg(c.b.sync) < target; ++spins) {
            if ((spins & 63u) == 63u) {
                if (t0 == 0) t0 = wall_clock64();
                if (ldg(c.b.sync + 1) != 0u || wall_clock64() - t0 > kSpinTicks) { ok = 0; break; }
            }
        }
        if (!ok) stg(c.b.sync + 1, 1u);
        *lds_flag() = ok;
    }
    __syncthreads();
    return *lds_flag() != 0;
}

// profiling: workgroup 0 notes (phase id, wall clock) -- only when the caller asked for stamps
__device__ __forceinline__ void stamp(Cx &c, int id)
{
    if (c.b.stamps && c.wg == 0 && c.tid == 0 && c.n_stamp < kStamps) {
        c.b.stamps[2 * c.n_stamp] = id; c.b.stamps[2 * c.n_stamp + 1] = wall_clock64();
        c.n_stamp += 1;
        c.b.stamps[2 * kStamps] = c.n_stamp;
    }
}

// a counter nobody has touched yet in this launch (the same one in every workgroup)
__device__ __forceinline__ int32_t *fresh_counter(Cx &c) { int32_t *p = c.b.ctr + (c.ctr_next < kCtr ? c.ctr_next : kCtr - 1); c.ctr_next += 1; return p; }

__device__ __forceinline__ double block_sum(Cx &c, double v)
{
    v = mqs::wave::sum1(v);
    __syncthreads();
    if (c.lane == 0) lds_red()[c.wave] = v;
    __syncthreads();
    return (lds_red()[0] + lds_red()[1]) + (lds_red()[2] + lds_red()[3]);
}

// [R | t] world -> camera (12, row-major 3 x 4)  ->  camera-to-world pose12 (R^T row-major, centre)
__device__ __forceinline__ void w2c_to_pose12(const double *M, double *o)
{
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) o[3 * a + b] = M[4 * b + a];
    for (int a = 0; a < 3; ++a) o[9 + a] = -(M[a] * M[3] + M[4 + a] * M[7] + M[8 + a] * M[11]);
}
__device__ __forceinline__ void pose12_to_w2c(const double *q, double *M)
{
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) M[4 * a + b] = q[3 * b + a];
        M[4 * a + 3] = -(q[a] * q[9] + q[3 + a] * q[10] + q[6 + a] * q[11]);
    }
}

__device__ __forceinline__ void so3_log(const double *R, double w[3])
{
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    const double k = mqs::so3_log_factor(0.5 * (R[0] + R[4] + R[8] - 1.0), 0.25 * fma(vx, vx, fma(vy, vy, vz * vz)));
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
}

__device__ __forceinline__ void retract_pose(const double *T, const double *dp, double *O)
{
    const double th2 = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
    double a, bq;
    mqs::so3_exp_factors(th2, a, bq);
    const double K[9] = {0, -dp[2], dp[1], dp[2], 0, -dp[0], -dp[1], dp[0], 0};
    double E[9];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
            const double k2 = K[3 * r] * K[q] + K[3 * r + 1] * K[3 + q] + K[3 * r + 2] * K[6 + q];
            E[3 * r + q] = ((r == q) ? 1.0 : 0.0) + a * K[3 * r + q] + bq * k2;
        }
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) O[3 * r + q] = T[3 * r] * E[q] + T[3 * r + 1] * E[3 + q] + T[3 * r + 2] * E[6 + q];
    for (int r = 0; r < 3; ++r) O[9 + r] = T[9 + r] + T[3 * r] * dp[3] + T[3 * r + 1] * dp[4] + T[3 * r + 2] * dp[5];
}

// the adjuster's camera model fx fy s u0 v0 k1 k2 p1 p2 (IO.hpp:230-236) from the frame loop's fx fy cx cy k1 k2 p1 p2 k3 (k3 = 0: the
// host side refuses anything else with the adjuster on; no skew in the loop's model)
__device__ __forceinline__ void load_calib(const Cx &c, double cal[9])
{
    const double *I = c.d.intr;                      // written before this launch
    cal[0] = I[0]; cal[1] = I[1]; cal[2] = 0.0; cal[3] = I[2]; cal[4] = I[3]; cal[5] = I[4]; cal[6] = I[5]; cal[7] = I[6]; cal[8] = I[7];
}

// the cameras of `poses` (global, this launch's) into LDS: 24 doubles each (ba_math.h: stage_camera)
__device__ __forceinline__ void stage_cams(Cx &c, const double *poses, double *sC)
{
    __syncthreads();
    for (int j = c.tid; j < c.P; j += kT) {
        double q[12], cal[9];
        for (int k = 0; k < 12; ++k) q[k] = ldg(poses + 12 * j + k);
        load_calib(c, cal);
        stage_camera(sC + j * kCamStride, q, cal, c.p.sigma_px);
    }
    __syncthreads();
}

// odometry factor (GTSAM 3.2.1 BetweenFactor<Pose3>; ba_sparse.hip: sparse_between_kernel) between the camera-to-world poses T1 -> T2
// (R row-major 9, t 3) with measurement Tm: e = (Log(Rm^T Rh), Rm^T (th - tm)), h = T1^-1 T2; also Rh, th for the Jacobian
__device__ __forceinline__ void between_error(const double *T1, const double *T2, const double *Tm, double e[6], double Rh[9], double th[3])
{
    double Re[9], w[3];
    const double dt[3] = {T2[9] - T1[9], T2[10] - T1[10], T2[11] - T1[11]};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rh[3 * i + j] = T1[i] * T2[j] + T1[3 + i] * T2[3 + j] + T1[6 + i] * T2[6 + j];
        th[i] = T1[i] * dt[0] + T1[3 + i] * dt[1] + T1[6 + i] * dt[2];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Re[3 * i + j] = Tm[i] * Rh[j] + Tm[3 + i] * Rh[3 + j] + Tm[6 + i] * Rh[6 + j];
    so3_log(Re, w);
    for (int i = 0; i < 3; ++i) {
        e[i] = w[i];
        e[3 + i] = Tm[i] * (th[0] - Tm[9]) + Tm[3 + i] * (th[1] - Tm[10]) + Tm[6 + i] * (th[2] - Tm[11]);
    }
}

// the prior on the first pose T at T0 (ba_sparse.hip: sparse_priors_kernel): e = (Log(R0^T R), R0^T (t - t0))
__device__ __forceinline__ void prior_error(const double *T0, const double *T, double e[6])
{
    double Rr[9], w[3];
    for (int a = 0; a < 3; ++a)
        for (int bq = 0; bq < 3; ++bq) Rr[3 * a + bq] = T0[a] * T[bq] + T0[3 + a] * T[3 + bq] + T0[6 + a] * T[6 + bq];
    so3_log(Rr, w);
    const double dt[3] = {T[9] - T0[9], T[10] - T0[10], T[11] - T0[11]};
    for (int a = 0; a < 3; ++a) {
        e[a] = w[a];
        e[3 + a] = T0[a] * dt[0] + T0[3 + a] * dt[1] + T0[6 + a] * dt[2];
    }
}

// the pose-dependent factors' cost at the poses in `sC` (LDS: a staged camera begins with its pose12): the odometry edges whose two
// ends are poses of the problem + the prior on its first pose (+ on the second anchor); every thread of the workgroup gets the sum
__device__ __forceinline__ const double *anchor_target(const Cx &c, int which)      // where an anchor's prior sits
{
    // pose 0 of the run: at its start-up estimate (bundle_adjust.cpp:273); any other pose: where this adjustment found it
    if (which == 0) return c.p.sel0 == 0 ? c.b.pose0 : c.b.poses_init;
    return c.b.poses_init + 12 * c.a2;
}

__device__ __forceinline__ double extras_cost(Cx &c, const double *sC)
{
    double v = 0.0;
    for (int t = c.tid; t < c.p.n_odo; t += kT) {
        const int kf = map_pose(c, ldg(c.b.odo_from + t)), kt = map_pose(c, ldg(c.b.odo_to + t));
        if (kf < 0 || kt < 0) continue;
        double Tm[12], e[6], Rh[9], th[3];
        for (int k = 0; k < 12; ++k) Tm[k] = ldg(c.b.odo_meas + 12 * t + k);
        between_error(sC + kf * kCamStride, sC + kt * kCamStride, Tm, e, Rh, th);
        for (int i = 0; i < 6; ++i) v += 0.5 * c.p.odo_w[i] * e[i] * e[i];
    }
    if (c.tid == 0 || (c.tid == 1 && c.a2 > 0)) {
        double e[6], T0[12];
        const double *tg = anchor_target(c, c.tid);
        for (int k = 0; k < 12; ++k) T0[k] = ldg(tg + k);
        prior_error(T0, sC + (c.tid == 0 ? 0 : c.a2) * kCamStride, e);
        for (int a = 0; a < 6; ++a) v += 0.5 * c.p.pose_w[a] * e[a] * e[a];
    }
    return block_sum(c, v);
}

// their blocks at the poses in `sC`, for the system phase: per edge [0..35] H1^T W H1 (from, from), [36..71] H1^T W (from, to),
// [72..77] W (to, to: diagonal), [78..83] -H1^T W e (g, from), [84..89] -W e (g, to), H1 = -Ad(h^-1) (the Jacobians of `between` only,
// as GTSAM 3.2.1 has them); the priors: [0..5] 1 / sigma^2, [6..11] -w e (first pose), [12..23] the same for the second anchor
__device__ __forceinline__ void extras_records(Cx &c, const double *sC)
{
    for (int t = c.tid; t < c.p.n_odo; t += kT) {
        const int kf = map_pose(c, ldg(c.b.odo_from + t)), kt = map_pose(c, ldg(c.b.odo_to + t));
        if (kf < 0 || kt < 0) continue;
        double Tm[12], e[6], Rh[9], th[3];
        for (int k = 0; k < 12; ++k) Tm[k] = ldg(c.b.odo_meas + 12 * t + k);
        between_error(sC + kf * kCamStride, sC + kt * kCamStride, Tm, e, Rh, th);
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
        double *rec = c.b.erec + (size_t)t * kERec;
        for (int i = 0; i < 6; ++i) {
            double gi = 0.0;
            for (int k = 0; k < 6; ++k) gi += H1[6 * k + i] * c.p.odo_w[k] * e[k];
            stg(rec + 78 + i, -gi);
            stg(rec + 84 + i, -c.p.odo_w[i] * e[i]);
            stg(rec + 72 + i, c.p.odo_w[i]);
            for (int j = 0; j < 6; ++j) {
                double sij = 0.0;
                for (int k = 0; k < 6; ++k) sij += H1[6 * k + i] * c.p.odo_w[k] * H1[6 * k + j];
                stg(rec + 6 * i + j, sij);
                stg(rec + 36 + 6 * i + j, H1[6 * j + i] * c.p.odo_w[j]);
            }
        }
    }
    if (c.tid == 0 || (c.tid == 1 && c.a2 > 0)) {
        double e[6], T0[12];
        const double *tg = anchor_target(c, c.tid);
        for (int k = 0; k < 12; ++k) T0[k] = ldg(tg + k);
        prior_error(T0, sC + (c.tid == 0 ? 0 : c.a2) * kCamStride, e);
        double *o = c.b.prec + 12 * c.tid;
        for (int a = 0; a < 6; ++a) { stg(o + a, c.p.pose_w[a]); stg(o + 6 + a, -c.p.pose_w[a] * e[a]); }
    }
}

// ---- landmark phases: kLanes lanes per landmark, each takes every kLanes-th pose of the landmark's [pmin, pmax] ---------------
struct LmWalk { int i, l8, p0, p1, q0, q1; bool in, live; double px, py, pz, pw, dx, dy, dz, tx, ty, tz; };   // (tx, ty, tz): where the landmark's prior sits (pw > 0)   // [p0, p1]: poses of a landmark in use; [q0, q1]: of any landmark

__device__ __forceinline__ LmWalk lm_begin(const Cx &c, int base, const double *pts)
{
    LmWalk w;
    w.i = base + c.tid / kLanes;
    w.l8 = c.tid % kLanes;
    w.in = w.i < c.N;
    w.live = w.in && ldg(c.b.use + (w.in ? w.i : 0)) != 0;
    const int ii = w.in ? w.i : 0;
    w.px = ldg(pts + 3 * ii); w.py = ldg(pts + 3 * ii + 1); w.pz = ldg(pts + 3 * ii + 2);
    w.pw = 0.0; w.dx = w.dy = w.dz = 0.0; w.tx = w.ty = w.tz = 0.0;
    if (w.live && w.i < c.n0) {
        // a start-up landmark: the gauge, at its given position (bundle_adjust.cpp:277-281)
        w.pw = c.p.prior_w;
        w.tx = c.b.objp0[3 * w.i]; w.ty = c.b.objp0[3 * w.i + 1]; w.tz = c.b.objp0[3 * w.i + 2];
        w.dx = w.px - w.tx; w.dy = w.py - w.ty; w.dz = w.pz - w.tz;
    } else if (w.live && c.p.out_prior_w > 0.0 && ldg(c.b.out_cnt + w.i) > 0) {
        // a selection: what the frames that are not poses of this problem know about the landmark stays with it as a prior at the
        // value this adjustment found it with
        w.pw = c.p.out_prior_w;
        w.tx = ldg(c.b.pts_init + 3 * w.i); w.ty = ldg(c.b.pts_init + 3 * w.i + 1); w.tz = ldg(c.b.pts_init + 3 * w.i + 2);
        w.dx = w.px - w.tx; w.dy = w.py - w.ty; w.dz = w.pz - w.tz;
    }
    w.q1 = w.in ? ldg(c.b.pmax + ii) : -1;
    w.q0 = (w.in && w.q1 >= 0) ? ldg(c.b.pmin + ii) : 0;            // (a landmark nobody observes: pmin is still INT_MAX)
    w.p0 = w.live ? w.q0 : 0;
    w.p1 = w.live ? w.q1 : -1;
    return w;
}

// sum / max / min over the 32 lanes of a landmark's half wavefront, on every lane of it: the rows of 16 exchanged by a
// v_permlane16_swap of the value with itself, the rest DPP moves (wave_reduce.h) -- no LDS round trips
__device__ __forceinline__ double half_sum(double v)
{
    double a = v, bq = v;
    mqs::wave::swap16(a, bq);
    double t = a + bq;
    t += mqs::wave::xor_lane<8>(t); t += mqs::wave::xor_lane<4>(t); t += mqs::wave::xor_lane<2>(t); t += mqs::wave::xor_lane<1>(t);
    return t;
}
__device__ __forceinline__ double half_max(double v)
{
    double a = v, bq = v;
    mqs::wave::swap16(a, bq);
    double t = fmax(a, bq);
    t = fmax(t, mqs::wave::xor_lane<8>(t)); t = fmax(t, mqs::wave::xor_lane<4>(t)); t = fmax(t, mqs::wave::xor_lane<2>(t)); t = fmax(t, mqs::wave::xor_lane<1>(t));
    return t;
}
#define MQS_LM_REDUCE(v) { v = half_sum(v); }

// A: records of every observation of every landmark in use, at (sCam, pts), damping lambda
__device__ __forceinline__ void phase_records(Cx &c, const double *pts, double lambda)
{
    for (int base = c.wg * (kT / kLanes); base < c.N; base += c.G * (kT / kLanes)) {
        const LmWalk w = lm_begin(c, base, pts);
        if (w.in && !w.live) {
            // a landmark that sits out: records that contribute nothing (the system phase does not look at `use`)
            for (int pp = w.q0 + w.l8; pp <= w.q1; pp += kLanes) {
                const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
                if (e < 0) continue;
                double *o = c.b.rec + e;
                const long long rs = c.b.rec_stride;
                stg(o, 0.0); stg(o + rs, 0.0); stg(o + 2 * rs, 1.0);
                for (int k = 3; k < kRec; ++k) stg(o + k * rs, 0.0);
            }
        }
        PointSystem ps;
        ps.H = mqs::Sym3{0, 0, 0, 0, 0, 0};
        ps.g = mqs::Vec3{0, 0, 0};
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const double *cam = lds_cam() + pp * kCamStride;
            const Factor fc = make_factor(cam, w.px, w.py, w.pz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
            double PR[2][3];
            make_PR(cam, fc.x, fc.y, PR);
            point_add_factor(ps, fc, PR);
        }
        MQS_LM_REDUCE(ps.H.xx) MQS_LM_REDUCE(ps.H.xy) MQS_LM_REDUCE(ps.H.xz) MQS_LM_REDUCE(ps.H.yy) MQS_LM_REDUCE(ps.H.yz) MQS_LM_REDUCE(ps.H.zz)
        MQS_LM_REDUCE(ps.g.x) MQS_LM_REDUCE(ps.g.y) MQS_LM_REDUCE(ps.g.z)
        point_finish(ps, w.pw, w.dx, w.dy, w.dz, lambda);
        const double m = ps.ok ? 1.0 : 0.0;
        double w0 = ps.g.x * ps.i00;
        double w1 = fma(-ps.l10, w0, ps.g.y) * ps.i11;
        double w2 = fma(-ps.l21, w1, fma(-ps.l20, w0, ps.g.z)) * ps.i22;
        w0 *= m; w1 *= m; w2 *= m;
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const double *cam = lds_cam() + pp * kCamStride;
            const Factor fc = make_factor(cam, w.px, w.py, w.pz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
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
            double *o = c.b.rec + e;
            const long long rs = c.b.rec_stride;
            stg(o, fc.x); stg(o + rs, fc.y); stg(o + 2 * rs, fc.Z);
            stg(o + 3 * rs, U[0][0]); stg(o + 4 * rs, U[0][1]); stg(o + 5 * rs, U[0][2]); stg(o + 6 * rs, U[1][0]); stg(o + 7 * rs, U[1][1]); stg(o + 8 * rs, U[1][2]);
            stg(o + 9 * rs, fc.F00 - (U[0][0] * U[0][0] + U[0][1] * U[0][1] + U[0][2] * U[0][2]));
            stg(o + 10 * rs, fc.F01 - (U[0][0] * U[1][0] + U[0][1] * U[1][1] + U[0][2] * U[1][2]));
            stg(o + 11 * rs, fc.F11 - (U[1][0] * U[1][0] + U[1][1] * U[1][1] + U[1][2] * U[1][2]));
            stg(o + 12 * rs, -fc.f0 - (U[0][0] * w0 + U[0][1] * w1 + U[0][2] * w2));
            stg(o + 13 * rs, -fc.f1 - (U[1][0] * w0 + U[1][1] * w1 + U[1][2] * w2));
        }
    }
}

// the cost of (sC, pts): every landmark in use; returns this workgroup's sum on every thread
__device__ __forceinline__ double phase_cost(Cx &c, const double *pts, const double *sC)
{
    double cost = 0.0;
    for (int base = c.wg * (kT / kLanes); base < c.N; base += c.G * (kT / kLanes)) {
        const LmWalk w = lm_begin(c, base, pts);
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const Factor fc = make_factor(sC + pp * kCamStride, w.px, w.py, w.pz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
            cost += fc.half_e2;
        }
        if (w.l8 == 0 && w.live) cost += 0.5 * w.pw * (w.dx * w.dx + w.dy * w.dy + w.dz * w.dz);
    }
    return block_sum(c, cost);
}

// worst pixel residual (+inf: an observation behind its camera) and smallest depth per landmark at (sC, pts) (ba_sparse.hip:
// sparse_worst_residual_kernel); landmarks not in use: 0 / +inf
__device__ __forceinline__ double phase_worst(Cx &c, const double *pts, const double *sC, bool with_depth, bool with_cost = false)
{
    double cost = 0.0;
    for (int base = c.wg * (kT / kLanes); base < c.N; base += c.G * (kT / kLanes)) {
        const LmWalk w = lm_begin(c, base, pts);
        double worst = 0.0, zmin = HUGE_VAL;
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const double *cam = sC + pp * kCamStride;
            const Factor fc = make_factor(cam, w.px, w.py, w.pz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
            worst = fmax(worst, fc.valid ? sqrt(2.0 * fc.half_e2) * c.p.sigma_px : HUGE_VAL);
            zmin = fmin(zmin, fma(cam[2], w.px - cam[9], fma(cam[5], w.py - cam[10], cam[8] * (w.pz - cam[11]))));
            cost += fc.half_e2;
        }
        if (w.l8 == 0 && w.live) cost += 0.5 * w.pw * (w.dx * w.dx + w.dy * w.dy + w.dz * w.dz);
        worst = half_max(worst);
        zmin = -half_max(-zmin);
        if (w.l8 == 0 && w.i < c.N) {
            stg(c.b.worst + w.i, worst);
            if (with_depth) stg(c.b.zmin + w.i, zmin);
        }
    }
    return with_cost ? block_sum(c, cost) : 0.0;
}

// The hit lists (once per adjustment, behind the per-pose lists): for every pose pair ja < jb the observations of the landmarks both
// see, as (log index in ja, log index in jb) in landmark order.  A task is pose ja with up to 32 consecutive poses jb: one wavefront
// loads ja's list ONCE (registers: 8 entries per lane), then looks the cells of eight jb at a time up in the table -- 64 loads in flight --
// counts their hits, takes the eight segments of the hit array with ONE atomic add and writes them from the registers.  (The first form
// -- one pair per task, one atomic add per pair -- spent 240 us of a 200-pose adjustment serialised on the cursor: 20 100 adds to one
// address at ~12 ns.)  The segments lie in the order the wavefronts arrive; inside a segment the order is the list's, so every sum over a
// segment is reproducible.  A pose with more than 512 observations: the slower pair-by-pair walk below.
__device__ __forceinline__ void build_hits_pair_slow(Cx &c, int ja, int jb, int na, int32_t *cursor, int32_t *overflow)
{
    const int N = c.N;
    const int t = jb * (jb + 1) / 2 + ja;
    const long long *list = c.b.plist + (size_t)ja * kListCap;
    int total = 0;
    for (int k0 = 0; k0 < na; k0 += 64) {
        const int k = k0 + c.lane;
        int eb = -1;
        if (k < na) eb = ldg(c.b.T + (size_t)jb * N + (int)(ldg(list + k) >> 32));
        total += __popcll(__ballot(eb >= 0));
    }
    int off = 0;
    if (c.lane == 0 && total > 0) off = atomicAdd(cursor, total);
    off = __builtin_amdgcn_readfirstlane(off);
    if (total > 0 && (long long)off + total > c.b.hits_cap) { if (c.lane == 0) atomicAdd(overflow, 1); total = 0; }
    if (c.lane == 0) { stg(c.b.blk_off + t, off); stg(c.b.blk_cnt + t, total); }
    if (total == 0) return;
    int w = off;
    for (int k0 = 0; k0 < na; k0 += 64) {
        const int k = k0 + c.lane;
        int ea = -1, eb = -1;
        if (k < na) { const long long el = ldg(list + k); ea = (int)(el & 0xffffffffll); eb = ldg(c.b.T + (size_t)jb * N + (int)(el >> 32)); }
        const unsigned long long bal = __ballot(eb >= 0);
        if (eb >= 0) stg(c.b.hits + w + __popcll(bal & ((1ull << c.lane) - 1ull)), (long long)(unsigned)ea | ((long long)eb << 32));
        w += __popcll(bal);
    }
}

__device__ __forceinline__ void build_hits(Cx &c, int32_t *cursor, int32_t *overflow)
{
    const int P = c.P, N = c.N;
    constexpr int kChunk = 32, kB = 8, kJ = 8;           // jb per task; list entries per lane in registers; jb per look-up batch
    // task -> (ja, chunk): pose ja has ceil((P - 1 - ja) / kChunk) chunks
    int ntask = 0;
    for (int ja = 0; ja + 1 < P; ++ja) ntask += (P - 1 - ja + kChunk - 1) / kChunk;
    for (int t = c.wg * 4 + c.wave; t < ntask; t += c.G * 4) {
        int ja = 0, rem = t;
        for (;; ++ja) { const int nc = (P - 1 - ja + kChunk - 1) / kChunk; if (rem < nc) break; rem -= nc; }
        const int jb0 = ja + 1 + rem * kChunk, jb1 = (jb0 + kChunk < P) ? jb0 + kChunk : P;
        const int na = ldg(c.b.pcount + ja);
        if (na > 64 * kB) {
            for (int jb = jb0; jb < jb1; ++jb) build_hits_pair_slow(c, ja, jb, na, cursor, overflow);
            continue;
        }
        const long long *list = c.b.plist + (size_t)ja * kListCap;
        long long el[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) { const int k = 64 * u + c.lane; el[u] = ldg(list + (k < na ? k : (na > 0 ? na - 1 : 0))); }
        for (int jq = jb0; jq < jb1; jq += kJ) {
            int eb[kJ][kB];
#pragma unroll
            for (int v = 0; v < kJ; ++v) {
                const int jb = jq + v < jb1 ? jq + v : jb1 - 1;          // (clamped: straight-line loads; the surplus is dropped below)
#pragma unroll
                for (int u = 0; u < kB; ++u) eb[v][u] = ldg(c.b.T + (size_t)jb * N + (int)(el[u] >> 32));
            }
            int cnt[kJ], total = 0;
#pragma unroll
            for (int v = 0; v < kJ; ++v) {
                cnt[v] = 0;
#pragma unroll
                for (int u = 0; u < kB; ++u) {
                    if (64 * u + c.lane >= na || jq + v >= jb1) eb[v][u] = -1;
                    cnt[v] += __popcll(__ballot(eb[v][u] >= 0));
                }
                total += cnt[v];
            }
            int off = 0;
            if (c.lane == 0 && total > 0) off = atomicAdd(cursor, total);
            off = __builtin_amdgcn_readfirstlane(off);
            const bool fits = (long long)off + total <= c.b.hits_cap;
            if (total > 0 && !fits && c.lane == 0) atomicAdd(overflow, 1);
            int w = off;
#pragma unroll
            for (int v = 0; v < kJ; ++v) {
                const int jb = jq + v;
                if (jb < jb1) {                                          // (wave-uniform)
                    if (c.lane == 0) { stg(c.b.blk_off + jb * (jb + 1) / 2 + ja, w); stg(c.b.blk_cnt + jb * (jb + 1) / 2 + ja, fits ? cnt[v] : 0); }
                    if (fits) {
#pragma unroll
                        for (int u = 0; u < kB; ++u) {
                            const unsigned long long bal = __ballot(eb[v][u] >= 0);
                            if (eb[v][u] >= 0) stg(c.b.hits + w + __popcll(bal & ((1ull << c.lane) - 1ull)), (long long)(unsigned)(int)(el[u] & 0xffffffffll) | ((long long)eb[v][u] << 32));
                            w += __popcll(bal);
                        }
                    }
                }
            }
        }
    }
}

// B: the reduced camera system.  Task (ja <= jb) = one wavefront: the block from the records of the pair's hit list (ja < jb) or of
// the pose's own observations (ja == jb).  Lower triangle of the augmented matrix, tile-major; row n = the right-hand side.
__device__ __forceinline__ double *s_entry(const Cx &c, int r, int q)      // r >= q
{
    return c.b.S + (size_t)tix(r >> 5, q >> 5) * (TB * TB) + (r & 31) * TB + (q & 31);
}

__device__ __forceinline__ void phase_system(Cx &c, double lambda)
{
    const int P = c.P;
    const int ntask = P * (P + 1) / 2;
    // what a task looks up about its poses -- the odometry edge that starts / ends there, that edge's other end, the length of the
    // pose's list -- once per workgroup into LDS: fetched per task these were up to three DEPENDENT round trips behind the block's sums
    int *sEo = reinterpret_cast<int *>(lds_cam_new()), *sEi = sEo + kMaxPoses, *sTo = sEi + kMaxPoses, *sCnt = sTo + kMaxPoses;
    __syncthreads();
    for (int j = c.tid; j < P; j += kT) {
        // (an edge counts when both its ends are poses of the problem; e_in / e_out are kept per trajectory pose)
        const int tj = pose_of(c, j);
        int eo = ldg(c.b.e_out + tj), ei = ldg(c.b.e_in + tj);
        int to = -1;
        if (eo >= 0 && eo < c.p.n_odo) { to = map_pose(c, ldg(c.b.odo_to + eo)); if (to < 0) eo = -1; } else eo = -1;
        if (ei >= 0 && ei < c.p.n_odo) { if (map_pose(c, ldg(c.b.odo_from + ei)) < 0) ei = -1; } else ei = -1;
        sEo[j] = eo;
        sEi[j] = ei;
        sTo[j] = to;
        sCnt[j] = ldg(c.b.pcount + j);
    }
    __syncthreads();
    int t = c.wg * 4 + c.wave;
    int off_next = 0, cnt_next = 0;
    if (t < ntask) { off_next = ldg(c.b.blk_off + t); cnt_next = ldg(c.b.blk_cnt + t); }      // (a diagonal task's pair is not used)
    for (; t < ntask; t += c.G * 4) {
        // t -> (jb, ja), ja <= jb: t = jb (jb + 1) / 2 + ja
        int jb = (int)((tri_root(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (jb * (jb + 1) / 2 > t) --jb;
        while ((jb + 1) * (jb + 2) / 2 <= t) ++jb;
        const int ja = t - jb * (jb + 1) / 2;
        stamp(c, 20);
        const int off = off_next, cnt = cnt_next;
        if (t + c.G * 4 < ntask) { off_next = ldg(c.b.blk_off + t + c.G * 4); cnt_next = ldg(c.b.blk_cnt + t + c.G * 4); }
        // the entries this lane will add to its sums from the odometry / prior records: requested now, used behind the reduction
        const int e_mine = (c.lane & 1) ? 32 + (c.lane >> 1) : (c.lane >> 1);
        const int mi = e_mine / 6, mj = e_mine % 6;
        double x_add = 0.0, x_gadd = 0.0;
        if (ja == jb) {
            if (sEo[ja] >= 0) x_add += ldg(c.b.erec + (size_t)sEo[ja] * kERec + 6 * mi + mj);
            if (sEi[ja] >= 0 && mi == mj) x_add += ldg(c.b.erec + (size_t)sEi[ja] * kERec + 72 + mi);
            if (ja == 0 && mi == mj) x_add += ldg(c.b.prec + mi);
            if (ja == c.a2 && mi == mj) x_add += ldg(c.b.prec + 12 + mi);
            if (c.lane < 6) {
                if (sEo[ja] >= 0) x_gadd += ldg(c.b.erec + (size_t)sEo[ja] * kERec + 78 + c.lane);
                if (sEi[ja] >= 0) x_gadd += ldg(c.b.erec + (size_t)sEi[ja] * kERec + 84 + c.lane);
                if (ja == 0) x_gadd += ldg(c.b.prec + 6 + c.lane);
                if (ja == c.a2) x_gadd += ldg(c.b.prec + 18 + c.lane);
            }
        } else if (sEo[ja] >= 0 && sTo[ja] == jb) x_add += ldg(c.b.erec + (size_t)sEo[ja] * kERec + 36 + 6 * mi + mj);
        double acc[36], gacc[6];
#pragma unroll
        for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e) gacc[e] = 0.0;
        // Loads are issued unconditionally and in straight-line batches (an index beyond the segment is clamped to its last entry, the
        // lane's contribution dropped afterwards): a load inside a divergent branch makes the wait-count bookkeeping give up and
        // every batch waited for the one before it (vmcnt(0) between them, seen in the listing) -- 8.8 us per task where the
        // arithmetic is 1
        constexpr int kU = 4;
        if (ja == jb) {
            // the pose's own observations: Jg^T (F - U U^T) Jg and the right-hand side
            const int na = sCnt[ja];
            const long long *list = c.b.plist + (size_t)ja * kListCap;
            for (int k0 = 0; k0 < na; k0 += 64 * kU) {
                int ea[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) { const int k = k0 + 64 * u + c.lane; ea[u] = (int)(ldg(list + (k < na ? k : na - 1)) & 0xffffffffll); }
                double r[kU][8];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const double *pa = c.b.rec + ea[u];
                    const long long rs = c.b.rec_stride;
                    r[u][0] = ldg(pa); r[u][1] = ldg(pa + rs); r[u][2] = ldg(pa + 2 * rs);
#pragma unroll
                    for (int q = 0; q < 5; ++q) r[u][3 + q] = ldg(pa + (9 + q) * rs);
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (k0 + 64 * u >= na) break;                       // (wave-uniform)
                    const bool mine = k0 + 64 * u + c.lane < na;
                    const JgA A = make_JgA(r[u][0], r[u][1], r[u][2]);
                    const double k00 = mine ? r[u][3] : 0.0, k01 = mine ? r[u][4] : 0.0, k11 = mine ? r[u][5] : 0.0;
                    const double r0 = mine ? r[u][6] : 0.0, r1 = mine ? r[u][7] : 0.0;
                    double Tm[2][6];
                    k_times_Jg(k00, k01, k01, k11, A, Tm);
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = i; j < 6; ++j) acc[i * 6 + j] += JgT_T(A, Tm, i, j);
#pragma unroll
                    for (int i = 0; i < 6; ++i) gacc[i] += JgT_r(A, r0, r1, i);
                }
            }
        } else {
            // the landmarks both poses see: -Jg_a^T (U_a U_b^T) Jg_b over the pair's hit list (built once per adjustment), four hits per
            // lane at a time: their index pairs, then all their records, in flight together
            for (int k0 = 0; k0 < cnt; k0 += 64 * kU) {
                long long h[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) { const int k = k0 + 64 * u + c.lane; h[u] = ldg(c.b.hits + off + (k < cnt ? k : cnt - 1)); }
                double ra[kU][9], rb[kU][9];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const double *pa = c.b.rec + (int)(h[u] & 0xffffffffll), *pb = c.b.rec + (int)(h[u] >> 32);
                    const long long rs = c.b.rec_stride;
#pragma unroll
                    for (int q = 0; q < 9; ++q) { ra[u][q] = ldg(pa + q * rs); rb[u][q] = ldg(pb + q * rs); }
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (k0 + 64 * u >= cnt) break;                      // (wave-uniform)
                    const bool mine = k0 + 64 * u + c.lane < cnt;
                    const JgA A = make_JgA(ra[u][0], ra[u][1], ra[u][2]), B = make_JgA(rb[u][0], rb[u][1], rb[u][2]);
                    const double k00 = mine ? -(ra[u][3] * rb[u][3] + ra[u][4] * rb[u][4] + ra[u][5] * rb[u][5]) : 0.0;
                    const double k01 = mine ? -(ra[u][3] * rb[u][6] + ra[u][4] * rb[u][7] + ra[u][5] * rb[u][8]) : 0.0;
                    const double k10 = mine ? -(ra[u][6] * rb[u][3] + ra[u][7] * rb[u][4] + ra[u][8] * rb[u][5]) : 0.0;
                    const double k11 = mine ? -(ra[u][6] * rb[u][6] + ra[u][7] * rb[u][7] + ra[u][8] * rb[u][8]) : 0.0;
                    double Tm[2][6];
                    k_times_Jg(k00, k01, k10, k11, B, Tm);
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[i * 6 + j] += JgT_T(A, Tm, i, j);
                }
            }
        }
        stamp(c, 21);
        // 36 sums over the wavefront (ba_sparse.hip: sparse_pair_groups_kernel): lane l ends with entry l >> 1 of the first 32,
        // the last four by butterflies
        double first32[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) first32[e] = acc[e];
        const double t32 = mqs::wave::wave_reduce32(first32, c.lane);
#pragma unroll
        for (int e = 32; e < 36; ++e) acc[e] = mqs::wave::sum1(acc[e]);
        if (ja == jb) {
#pragma unroll
            for (int e = 0; e < 6; ++e) gacc[e] = mqs::wave::sum1(gacc[e]);
        }
        const int e_out = (c.lane & 1) ? 32 + (c.lane >> 1) : (c.lane >> 1);
        const bool writer = !(c.lane & 1) || c.lane < 8;
        double v = t32;
        if (c.lane & 1) {
            v = acc[32];
#pragma unroll
            for (int e = 33; e < 36; ++e) v = (e_out == e) ? acc[e] : v;
        }
        const int i = writer ? e_out / 6 : 0, j = writer ? e_out % 6 : 0;
        stamp(c, 22);
        // odometry edges and the pose prior (their records were written in the phase before; requested at the task's start), damping
        if (writer) {
            v += x_add;
            if (ja == jb && i == j) v = (lambda >= 0.0) ? v * (1.0 + lambda) : v - lambda;
            // entry (6 ja + i, 6 jb + j) of the symmetric system; kept: the lower triangle
            if (ja != jb) stg(s_entry(c, 6 * jb + j, 6 * ja + i), v);
            else if (i <= j) stg(s_entry(c, 6 * ja + j, 6 * ja + i), v);
        }
        if (ja == jb && c.lane < 6) {
            double gv = gacc[0];
#pragma unroll
            for (int e = 1; e < 6; ++e) gv = (c.lane == e) ? gacc[e] : gv;
            stg(s_entry(c, c.n, 6 * ja + c.lane), gv + x_gadd);
        }
    }
    if (c.wg == 0 && c.tid == 0) stg(s_entry(c, c.n, c.n), kAugDiag);
}

// ---- C: blocked Cholesky of the augmented matrix over its tiles -------------------------------------------------------------
__device__ __forceinline__ void load_tile(Cx &c, const double *g, double *sT)          // 32 x 32 row-major (global, this launch's) -> LDS, stride TLD
{
    for (int e = c.tid; e < TB * TB; e += kT) sT[(e >> 5) * TLD + (e & 31)] = ldg(g + e);
}

// The diagonal tile's factor: the four-wave form with a barrier per pivot (6.3 us per tile; the default), or one wavefront working in
// panels of four pivots on the matrix pipe (chol_block.h: factor_diag_block_mfma_wave; -DMQS_SLAM_BA_DIAG_MFMA=1).  The blocked form was
// built in round 6 and measured on the example sequence's 107 factorisations: 71 against 61.5 us per factorisation (7.7 against 6.3 us per
// tile) -- a lone wavefront pays ~0.96 us per panel of dependent chain (four reciprocal square roots, three LDS hand-overs, the gather
// for the inverse) where the four-wave form pays four barriers.  It stays as the A/B form (tools/probes/ab_diag_factor_r06.sh) and under
// its test (mqs_debug_factor32).
#ifndef MQS_SLAM_BA_DIAG_MFMA
#define MQS_SLAM_BA_DIAG_MFMA 0
#endif
__device__ __forceinline__ void factor_diag_tile(Cx &c, const double *sT, double *tile, int32_t *badw, double *sM)
{
#if MQS_SLAM_BA_DIAG_MFMA
    if (c.wave == 0) mqs::chol::factor_diag_block_mfma_wave<true>(sT, tile, badw, c.lane, sM);
#else
    mqs::chol::factor_diag_block_from_lds_4w<true>(sT, tile, TB, 0, badw, c.tid, sM);
#endif
}

__device__ __forceinline__ void chol_diag0(Cx &c, double *sT, double *sM, int32_t *badw)
{
    double *tile = c.b.S + (size_t)tix(0, 0) * (TB * TB);
    __syncthreads();
    load_tile(c, tile, sT);
    __syncthreads();
    // (a single-wavefront factor without a barrier per pivot -- the column through wave-private LDS, the next pivot's entry by
    // v_readlane -- was built and measured in round 5: 10.8 us per block against this form's 6.3; one wave alone issues a dependent
    // fp64 instruction every ~16 cycles and the 32 reciprocal square roots sit on that chain)
    factor_diag_tile(c, sT, tile, badw, sM);
}

// tile (bi, bj) of step k (k < bj <= bi): A(bi, bj) -= X_i X_j^T with X = A(., k) inv(L_kk)^T formed here; the tiles of column
// k + 1 also store their X_i (the finished block L(bi, k)); tile (k + 1, k + 1) is factored on the spot
__device__ __forceinline__ void chol_task(Cx &c, int k, int bi, int bj, double *sBuf, int32_t *badw)
{
    double *sLi = sBuf, *sAi = sBuf + TB * TLD, *sAj = sBuf + 2 * TB * TLD, *sXi = sBuf + 3 * TB * TLD, *sXj = sBuf + 4 * TB * TLD;
    double *sM = sBuf + 5 * TB * TLD;
    const double *D = c.b.S + (size_t)tix(k, k) * (TB * TB);
    const double *Ai = c.b.S + (size_t)tix(bi, k) * (TB * TB), *Aj = c.b.S + (size_t)tix(bj, k) * (TB * TB);
    const int wr = c.wave >> 1, wc = c.wave & 1;
    double *tile = c.b.S + (size_t)tix(bi, bj) * (TB * TB);
    // every load of the task up front, unconditional and row-major (a wavefront's 64 entries = 4 lines): the three panel inputs AND the
    // tile to update -- its round trip used to follow the products
    double dv[4], ai[4], aj[4], tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = c.tid + u * kT;
        dv[u] = ldg(D + e); ai[u] = ldg(Ai + e); aj[u] = ldg(Aj + e);
    }
    {
        const int q = mqs::chol::quadrant_col(wc, c.lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) tv[v] = ldg(tile + mqs::chol::quadrant_row(wr, c.lane, v) * TB + q);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = c.tid + u * kT, a = e >> 5, bq = e & 31;
        // the diagonal tile's strictly upper triangle is inv(L)'s strictly lower one, transposed; its diagonal is L's
        sLi[bq * TLD + a] = (bq > a) ? dv[u] : ((bq == a) ? mqs::rcp(dv[u]) : 0.0);      // (Newton reciprocal: the IEEE division sat in front of every task's first product)
        sAi[a * TLD + bq] = ai[u];
        sAj[a * TLD + bq] = aj[u];
    }
    __syncthreads();
    {
        const mqs::chol::double4v xi = mqs::chol::tile_quadrant_mfma(sAi, sLi, wr, wc, c.lane);
        const mqs::chol::double4v xj = (bi == bj) ? xi : mqs::chol::tile_quadrant_mfma(sAj, sLi, wr, wc, c.lane);
        const int q = mqs::chol::quadrant_col(wc, c.lane);
        double *Lout = c.b.Lp + (size_t)tix(bi, k) * (TB * TB);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = mqs::chol::quadrant_row(wr, c.lane, v);
            sXi[r * TLD + q] = xi[v];
            sXj[r * TLD + q] = xj[v];
            if (bj == k + 1) stg(Lout + r * TB + q, xi[v]);
        }
    }
    __syncthreads();
    const mqs::chol::double4v acc = mqs::chol::tile_quadrant_mfma(sXi, sXj, wr, wc, c.lane);
    const bool next_diag = bi == k + 1 && bj == k + 1;
    double *sT = sAi;
    __syncthreads();
    {
        const int q = mqs::chol::quadrant_col(wc, c.lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = mqs::chol::quadrant_row(wr, c.lane, v);
            const double val = tv[v] - acc[v];
            if (next_diag) sT[r * TLD + q] = val;
            else stg(tile + r * TB + q, val);
        }
    }
    if (!next_diag) return;
    __syncthreads();
    factor_diag_tile(c, sT, tile, badw, sM);
}

__device__ __forceinline__ bool phase_cholesky(Cx &c, double *sBuf, int32_t *badw)
{
    if (c.wg == 0) chol_diag0(c, sBuf, sBuf + 5 * TB * TLD, badw);
    if (!grid_barrier(c)) return false;
    for (int k = 0; k + 1 < c.nt; ++k) {
        const int m = c.nt - 1 - k, ntask = m * (m + 1) / 2;
        // task 0 = the next diagonal tile (the long one: it ends with a factorisation) on workgroup 0 alone, the others spread
        for (int t = (c.G == 1 ? 0 : (c.wg == 0 ? 0 : c.wg)); t < ntask; t += (c.G == 1 ? 1 : (c.wg == 0 ? ntask : c.G - 1))) {
            int a = (int)((tri_root(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while (a * (a + 1) / 2 > t) --a;
            while ((a + 1) * (a + 2) / 2 <= t) ++a;
            const int bq = t - a * (a + 1) / 2;
            chol_task(c, k, k + 1 + a, k + 1 + bq, sBuf, badw);
        }
        if (!grid_barrier(c)) return false;
    }
    return true;
}

// D: L^T x = y, y = the augmented row of the factor; x -> dpose (global).
// Block rows from the last to the first, in SUPER-BLOCKS of kSuper tile rows (256 unknowns):
//   * inside a super-block workgroup 0 substitutes: x_kb = inv(L_kb)^T y_kb (the inverse lies in the diagonal tile's upper triangle), then
//     y_j -= L(kb, j)^T x_kb for the blocks of the SAME super-block left of it -- at most 224 columns, one per thread.  A load of another
//     XCD's write-through data takes ~3.3 us here and the compiler waits for a batch where it is issued, not where it is used (listing and
//     phase stamps: prefetching one row ahead bought nothing), so the rows go four at a time: all their loads in flight together;
//   * the super-block's x then goes to global memory, and ALL workgroups apply it to the columns left of the super-block -- one 32-column
//     tile column per workgroup, its eight tiles (64 KB) in flight at once, the eight partial sums added in a fixed order -- between two
//     grid barriers.  (Round 5's first form did everything in workgroup 0: at 200 poses the columns beyond the first 256 were fetched in
//     a loop of dependent round trips, 15 us per block row, 581 us per solve -- more than the factorisation.)
// Systems of up to 256 unknowns (42 poses) are one super-block: no barrier inside.
constexpr int kSuper = 8;

struct BsRow { double diag[4]; double col[TB]; };

__device__ __forceinline__ void bs_load(const Cx &c, int kb, int lo, BsRow &r)
{
    const double *D = c.b.S + (size_t)tix(kb, kb) * (TB * TB);
#pragma unroll
    for (int u = 0; u < 4; ++u) r.diag[u] = ldg(D + c.tid + u * kT);
    // unconditional, the column index clamped (a load inside a divergent branch serialises the batches behind it); kb = lo: a tile
    // nobody needs -- the values are not used
    int q = lo * TB + c.tid;
    if (q >= kb * TB) q = kb * TB - 1;
    if (q < 0) q = 0;
    const double *Lt = c.b.Lp + (size_t)tix(kb > 0 ? kb : 1, q >> 5) * (TB * TB) + (q & 31);
#pragma unroll
    for (int rr = 0; rr < TB; ++rr) r.col[rr] = ldg(Lt + rr * TB);
}

// sX: this super-block's y -> x, indexed from the super-block's first row (lo * TB)
__device__ __forceinline__ void bs_step(Cx &c, int kb, int lo, const BsRow &r, double *sX, double *sLi)
{
    const int n = c.n;
    // the diagonal tile's upper triangle = inv(L)^T rows; its diagonal = L_jj
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = c.tid + u * kT, a = e >> 5, bq = e & 31;
        sLi[a * TLD + bq] = (bq > a) ? r.diag[u] : ((bq == a) ? mqs::rcp(r.diag[u]) : 0.0);
    }
    __syncthreads();
    if (c.tid < TB) {
        const int a = c.tid;
        double xa = 0.0;
#pragma unroll 8
        for (int bq = 0; bq < TB; ++bq) xa = fma(sLi[a * TLD + bq], sX[(kb - lo) * TB + bq], xa);      // zero below the diagonal
        if (kb * TB + a >= n) xa = 0.0;
        sLi[TB * TLD + a] = xa;
    }
    __syncthreads();
    if (c.tid < TB) sX[(kb - lo) * TB + c.tid] = sLi[TB * TLD + c.tid];
    if (lo * TB + c.tid < kb * TB) {
        double sacc = 0.0;
#pragma unroll
        for (int rr = 0; rr < TB; ++rr) sacc = fma(r.col[rr], sLi[TB * TLD + rr], sacc);
        sX[c.tid] -= sacc;
    }
    __syncthreads();
}

__device__ __forceinline__ bool phase_backsolve(Cx &c, double *sX, double *sLi, int &barriers)
{
    const int n = c.n, nt = c.nt;
    const int br = n >> 5, rr = n & 31;                   // the right-hand side's row: tile row br (= nt - 1), local row rr
    constexpr int kRows = 4;
    for (int hi = nt; hi > 0;) {
        const int lo = hi > kSuper ? hi - kSuper : 0;
        if (c.wg == 0) {
            BsRow rows[kRows];
            bool first = true;
            for (int kb = hi - 1; kb >= lo; kb -= kRows) {
#pragma unroll
                for (int u = 0; u < kRows; ++u) bs_load(c, kb - u >= lo ? kb - u : lo, lo, rows[u]);
                if (first) {
                    first = false;
                    __syncthreads();
                    // this super-block's right-hand side: the factor's augmented row (the last super-block: nothing has touched it yet; it also
                    // seeds the global copy the other workgroups will subtract from), else the global copy
                    if (hi == nt) {
                        for (int q = c.tid; q < nt * TB; q += kT) {
                            double y = 0.0;
                            if (q < n) {
                                const int bc = q >> 5;
                                y = (bc < br) ? ldg(c.b.Lp + (size_t)tix(br, bc) * (TB * TB) + rr * TB + (q & 31))
                                              : ldg(c.b.S + (size_t)tix(br, br) * (TB * TB) + rr * TB + (q & 31));
                            }
                            if (q >= lo * TB) sX[q - lo * TB] = y;
                            else stg(c.b.ysol + q, y);
                        }
                    } else {
                        for (int q = c.tid; q < (hi - lo) * TB; q += kT) sX[q] = ldg(c.b.ysol + lo * TB + q);
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int u = 0; u < kRows; ++u)
                    if (kb - u >= lo) bs_step(c, kb - u, lo, rows[u], sX, sLi);
            }
            for (int q = c.tid; q < (hi - lo) * TB; q += kT)
                if (lo * TB + q < n) stg(c.b.dpose + lo * TB + q, sX[q]);
        }
        if (lo == 0) break;
        if (!grid_barrier(c)) return false;
        barriers += 1;
        // the super-block applied to the columns left of it: workgroup -> tile column j, thread -> (tile row lo + g, column cc)
        {
            double *sXs = sX;                             // [kSuper * TB] the super-block's x
            double *sPart = sLi;                          // [kSuper][TB]
            __syncthreads();
            for (int q = c.tid; q < (hi - lo) * TB; q += kT) sXs[q] = (lo * TB + q < n) ? ldg(c.b.dpose + lo * TB + q) : 0.0;
            __syncthreads();
            const int g = c.tid >> 5, cc = c.tid & 31;
            for (int j = c.wg; j < lo; j += c.G) {
                double part = 0.0;
                const int i = lo + g;
                if (i < hi) {
                    const double *Lt = c.b.Lp + (size_t)tix(i, j) * (TB * TB) + cc;
                    double lv[TB];
#pragma unroll
                    for (int r2 = 0; r2 < TB; ++r2) lv[r2] = ldg(Lt + r2 * TB);
#pragma unroll
                    for (int r2 = 0; r2 < TB; ++r2) part = fma(lv[r2], sXs[g * TB + r2], part);
                }
                __syncthreads();
                sPart[g * TB + cc] = part;
                __syncthreads();
                if (c.tid < TB) {
                    double t = 0.0;
#pragma unroll
                    for (int g2 = 0; g2 < kSuper; ++g2) t += sPart[g2 * TB + c.tid];
                    const int q = j * TB + c.tid;
                    stg(c.b.ysol + q, ldg(c.b.ysol + q) - t);
                }
            }
        }
        if (!grid_barrier(c)) return false;
        barriers += 1;
        hi = lo;
    }
    return true;
}

// E: landmarks back-substituted at the linearisation point (sCam, pts), the step's poses retracted by every workgroup for itself
// (sCamN), then the cost of the trial estimate; returns the workgroup's cost sum
__device__ __forceinline__ double phase_backsub_cost(Cx &c, const double *poses, const double *pts, double *poses_new, double *pts_new, double lambda,
                                     bool want_cost)
{
    __syncthreads();
    for (int q = c.tid; q < c.n; q += kT) lds_step()[q] = ldg(c.b.dpose + q);
    __syncthreads();
    for (int j = c.tid; j < c.P; j += kT) {
        double q[12], o[12], cal[9];
        for (int k = 0; k < 12; ++k) q[k] = ldg(poses + 12 * j + k);
        retract_pose(q, lds_step() + 6 * j, o);
        load_calib(c, cal);
        stage_camera(lds_cam_new() + j * kCamStride, o, cal, c.p.sigma_px);
        if (j % c.G == c.wg)
            for (int k = 0; k < 12; ++k) stg(poses_new + 12 * j + k, o[k]);
    }
    __syncthreads();
    double cost = 0.0;
    for (int base = c.wg * (kT / kLanes); base < c.N; base += c.G * (kT / kLanes)) {
        const LmWalk w = lm_begin(c, base, pts);
        PointSystem ps;
        ps.H = mqs::Sym3{0, 0, 0, 0, 0, 0};
        ps.g = mqs::Vec3{0, 0, 0};
        double rx = 0, ry = 0, rz = 0;
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const double *cam = lds_cam() + pp * kCamStride;
            const Factor fc = make_factor(cam, w.px, w.py, w.pz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
            double PR[2][3];
            make_PR(cam, fc.x, fc.y, PR);
            point_add_factor(ps, fc, PR);
            double Jg[2][6];
            make_Jg(fc.x, fc.y, fc.Z, Jg);
            const double *dp = lds_step() + 6 * pp;
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) { s0 = fma(Jg[0][q], dp[q], s0); s1 = fma(Jg[1][q], dp[q], s1); }
            const double t0 = fma(fc.F00, s0, fc.F01 * s1), t1 = fma(fc.F01, s0, fc.F11 * s1);
            rx = fma(PR[0][0], t0, fma(PR[1][0], t1, rx));
            ry = fma(PR[0][1], t0, fma(PR[1][1], t1, ry));
            rz = fma(PR[0][2], t0, fma(PR[1][2], t1, rz));
        }
        MQS_LM_REDUCE(ps.H.xx) MQS_LM_REDUCE(ps.H.xy) MQS_LM_REDUCE(ps.H.xz) MQS_LM_REDUCE(ps.H.yy) MQS_LM_REDUCE(ps.H.yz) MQS_LM_REDUCE(ps.H.zz)
        MQS_LM_REDUCE(ps.g.x) MQS_LM_REDUCE(ps.g.y) MQS_LM_REDUCE(ps.g.z)
        MQS_LM_REDUCE(rx) MQS_LM_REDUCE(ry) MQS_LM_REDUCE(rz)
        point_finish(ps, w.pw, w.dx, w.dy, w.dz, lambda);
        double v0 = ps.g.x - rx, v1 = ps.g.y - ry, v2 = ps.g.z - rz;
        v0 = v0 * ps.i00;
        v1 = fma(-ps.l10, v0, v1) * ps.i11;
        v2 = fma(-ps.l21, v1, fma(-ps.l20, v0, v2)) * ps.i22;
        apply_Lt_inv(ps, v0, v1, v2);
        const double m = (w.live && ps.ok) ? 1.0 : 0.0;
        const double nx = w.px + m * v0, ny = w.py + m * v1, nz = w.pz + m * v2;
        if (w.l8 == 0 && w.i < c.N) { stg(pts_new + 3 * w.i, nx); stg(pts_new + 3 * w.i + 1, ny); stg(pts_new + 3 * w.i + 2, nz); }
        if (!want_cost) continue;
        for (int pp = w.p0 + w.l8; pp <= w.p1; pp += kLanes) {
            const int e = ldg(c.b.T + (size_t)pp * c.N + w.i);
            if (e < 0) continue;
            const Factor fc = make_factor(lds_cam_new() + pp * kCamStride, nx, ny, nz, c.d.log_uv[2 * e], c.d.log_uv[2 * e + 1], true);
            cost += fc.half_e2;
        }
        if (w.l8 == 0 && w.live && w.pw > 0.0) {
            const double ex = nx - w.tx, ey = ny - w.ty, ez = nz - w.tz;
            cost += 0.5 * w.pw * (ex * ex + ey * ey + ez * ez);
        }
    }
    return want_cost ? block_sum(c, cost) : 0.0;
}

// every workgroup's cost piece (published in front of a barrier that has been passed) added in one fixed order: lane g takes the
// pieces g, g + 64, ..., the lanes' sums go through the wavefront butterfly; on every thread
__device__ __forceinline__ double sum_partials(Cx &c)
{
    double s = 0.0;
    if (c.wave == 0) {
        for (int g = c.lane; g < c.G; g += 64) s += ldg(c.b.partials + 4 * g);
        s = mqs::wave::sum1(s);
    }
    __syncthreads();
    if (c.tid == 0) lds_red()[8] = s;
    __syncthreads();
    return lds_red()[8];
}

// this workgroup's cost piece published, the barrier, every workgroup's pieces added in the same order
__device__ __forceinline__ bool reduce_cost(Cx &c, double mine, double &total)
{
    if (c.tid == 0) stg(c.b.partials + 4 * c.wg, mine);
    if (!grid_barrier(c)) return false;
    total = sum_partials(c);
    return true;
}

__device__ __forceinline__ bool count_barrier(Cx &c, int32_t *ctr, int &value)
{
    if (!grid_barrier(c)) return false;
    value = ldg(ctr);
    return true;
}

// one Levenberg-Marquardt trial at damping l from (poses, pts): phases A - E; cost of the trial estimate in `fresh`, ok = the
// factorisation met no non-positive pivot
__device__ __forceinline__ bool lm_trial(Cx &c, const double *poses, const double *pts, double *poses_new, double *pts_new, double l, bool want_cost,
                         double &fresh, bool &ok, int &barriers)
{
    stamp(c, 10);
    stage_cams(c, poses, lds_cam());
    phase_records(c, pts, l);
    if (c.wg == c.G - 1) extras_records(c, lds_cam());      // the last workgroup: its share of the landmarks is the smallest
    int32_t *badw = fresh_counter(c);
    stamp(c, 11);
    if (!grid_barrier(c)) return false;
    stamp(c, 12);
    phase_system(c, l);
    stamp(c, 13);
    if (!grid_barrier(c)) return false;
    stamp(c, 14);
    if (!phase_cholesky(c, lds_cam_new(), badw)) return false;
    stamp(c, 15);
    if (!phase_backsolve(c, lds_cam_new(), lds_cam_new() + kMaxPoses * 6 + 64, barriers)) return false;
    stamp(c, 16);
    if (!grid_barrier(c)) return false;
    stamp(c, 17);
    double mine = phase_backsub_cost(c, poses, pts, poses_new, pts_new, l, want_cost);
    stamp(c, 18);
    barriers += 4 + c.nt;
    if (!want_cost) { fresh = 0.0; ok = true; return grid_barrier(c); }
    if (c.wg == c.G - 1) mine += extras_cost(c, lds_cam_new());         // every workgroup has retracted all poses for itself
    if (!reduce_cost(c, mine, fresh)) return false;
    stamp(c, 19);
    barriers += 1;
    ok = ldg(badw) == 0;
    return true;
}

__global__ __launch_bounds__(kT) void slam_ba_kernel(BaDev b, SlamDev d, BaParams p)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { stg(b.sync_next, 0u); stg(b.sync_next + 1, 0u); }      // the next launch's barrier words
    Cx c;
    c.b = b; c.d = d; c.p = p;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.wg = blockIdx.x; c.G = gridDim.x;
    c.epoch = 0; c.ctr_next = 0; c.n_stamp = 0;
    const int P = p.P;
    const int N = d.cnt[C_NLAND], nlog = d.cnt[C_NLOG];
    c.P = P; c.N = N; c.nlog = nlog; c.n = 6 * P; c.nt = (6 * P + 1 + TB - 1) / TB;
    c.n0 = p.sel0 == 0 ? b.n0 : 0;
    c.a2 = -1;
    const int gtid = c.wg * kT + c.tid, gthreads = c.G * kT;
    int status = 0;
    if (d.cnt[C_LOG_OVERFLOW]) status = 3;
    else if (P < 1 || P > kMaxPoses || P > b.P_cap || N > b.N_cap || tix(c.nt - 1, c.nt - 1) >= b.ntile_cap || N < 1) status = 2;
    if (!status && !p.identity) {
        // the selection, from the pinned host block into this workgroup's LDS (1 KB over the fabric, once)
        for (int k = c.tid; k < P; k += kT) lds_sel()[k] = b.sel_host[k];
        __syncthreads();
        if (p.anchor2 >= 0) c.a2 = map_pose(c, p.anchor2);
        if (c.a2 == 0) c.a2 = -1;
    }
    if (status) {
        if (gtid == 0) { b.report[0] = (double)status; b.report[1] = (double)P; b.report[2] = (double)N; }
        return;
    }
    int barriers = 0;
    stamp(c, 0);

    // ---- 0: clear the tables, the working estimate, the new odometry edge ------------------------------------------------------
    for (int64_t e = gtid; e < (int64_t)P * N; e += gthreads) stg(b.T + e, -1);
    for (int l = gtid; l < N; l += gthreads) {
        stg(b.per_lm + l, 0); stg(b.pmin + l, 0x7fffffff); stg(b.pmax + l, -1); stg(b.out_cnt + l, 0);
        for (int k = 0; k < 3; ++k) { const double v = d.map[3 * l + k]; stg(b.pts_a + 3 * l + k, v); stg(b.pts_init + 3 * l + k, v); }
    }
    for (int k = gtid; k < kCtr; k += gthreads) stg(b.ctr + k, 0);
    if (gtid < 2) stg(b.med + gtid, (double)NAN);
    for (int j = gtid; j < P; j += gthreads) {
        double q[12];
        const double *row = d.traj + 12 * (size_t)pose_of(c, j);
        if (j == 0 && p.sel0 == 0) for (int k = 0; k < 12; ++k) q[k] = b.pose0[k];       // the prior sits at the start-up estimate (bundle_adjust.cpp:273)
        else w2c_to_pose12(row, q);
        for (int k = 0; k < 12; ++k) { stg(b.poses_a + 12 * j + k, q[k]); stg(b.poses_init + 12 * j + k, q[k]); stg(b.traj_old + 12 * j + k, row[k]); }
    }
    {
        // the system's tiles: zero, ones on the diagonal beyond the augmented row (rows that act as identity)
        const int64_t tot = (int64_t)tix(c.nt - 1, c.nt - 1) + 1;
        for (int64_t e = gtid; e < tot * TB * TB; e += gthreads) stg(b.S + e, 0.0);
    }
    if (p.add_edge && c.wg == 0 && c.tid == 0) {
        // slam2.py:681-687: base keyframe -> keyframe, from the poses as they stand now: M = P1 inv(P0), as a camera-to-world pose12
        const int t = p.n_odo - 1;
        const double *P0 = d.traj + 12 * (size_t)p.edge_from, *P1 = d.traj + 12 * (size_t)p.edge_to;
        double M[12];
        for (int a = 0; a < 3; ++a) {
            for (int bq = 0; bq < 3; ++bq) M[4 * a + bq] = P1[4 * a] * P0[4 * bq] + P1[4 * a + 1] * P0[4 * bq + 1] + P1[4 * a + 2] * P0[4 * bq + 2];
        }
        for (int a = 0; a < 3; ++a) M[4 * a + 3] = P1[4 * a + 3] - (M[4 * a] * P0[3] + M[4 * a + 1] * P0[7] + M[4 * a + 2] * P0[11]);
        double q[12];
        w2c_to_pose12(M, q);
        for (int k = 0; k < 12; ++k) stg(b.odo_meas + 12 * t + k, q[k]);
        stg(b.odo_from + t, p.edge_from); stg(b.odo_to + t, p.edge_to);
        stg(b.e_out + p.edge_from, t); stg(b.e_in + p.edge_to, t);
    }
    if (!grid_barrier(c)) { if (c.tid == 0 && c.wg == 0) b.report[0] = 1.0; return; }
    barriers += 1;
    {
        const int64_t nd = (int64_t)c.nt * TB;
        for (int r = c.n + 1 + gtid; r < nd; r += gthreads) stg(s_entry(c, r, r), 1.0);
    }
    stamp(c, 1);
    // ---- 1: the log into the table --------------------------------------------------------------------------------------------
    int32_t *dup_ctr = fresh_counter(c), *cap_ctr = fresh_counter(c);
    for (int e = gtid; e < nlog; e += gthreads) {
        int lm = d.log_lm[e];
        if (lm <= -2) { const int t = -2 - lm; lm = t < d.tid_cap ? d.tid2lm[t] : -1; }
        const int tj = d.log_pose[e];
        const int pj = map_pose(c, tj);
        if (pj < 0 && tj >= 0 && tj < p.P_all && lm >= 0 && lm < N) atomicAdd(b.out_cnt + lm, 1);      // seen from a frame that is not a pose of this problem
        const double u = d.log_uv[2 * e], v = d.log_uv[2 * e + 1];
        // an observation nearer to the image border than the margin was tracked on a window that read the border-extended
        // pyramid: good enough for RANSAC, not for least squares (slam_device.py)
        const bool inside = p.border <= 0.0 || (u >= p.border && u <= p.W - 1 - p.border && v >= p.border && v <= p.H - 1 - p.border);
        const bool known = lm >= 0 && lm < N && pj >= 0 && inside;
        stg(b.lm_res + e, known ? lm : -1);
        if (known) {
            const int old = atomicMax(b.T + (size_t)pj * N + lm, e);
            if (old < 0) { atomicAdd(b.per_lm + lm, 1); atomicMin(b.pmin + lm, pj); atomicMax(b.pmax + lm, pj); }
            else atomicAdd(dup_ctr, 1);               // the later log entry of a (frame, landmark) cell is the one the table keeps
        }
    }
    if (!grid_barrier(c)) { if (c.tid == 0 && c.wg == 0) b.report[0] = 1.0; return; }
    barriers += 1;
    stamp(c, 2);
    // ---- 2: which landmarks take part; every pose's observations compacted in landmark order --------------------------------------
    for (int l = gtid; l < N; l += gthreads) {
        // (a selection: the observations at frames outside the problem count towards "seen often enough", one inside it is needed)
        const int n_in = ldg(b.per_lm + l);
        const bool use = ((n_in >= 1 && n_in + ldg(b.out_cnt + l) >= p.min_observations) || l < c.n0) && ldg(b.bad + l) == 0;
        stg(b.use + l, use ? 1 : 0);
    }
    for (int pj = c.wg * 4 + c.wave; pj < P; pj += c.G * 4) {
        int cnt = 0, lo = 0x7fffffff, hi = -1;
        for (int l0 = 0; l0 < N; l0 += 64) {
            const int l = l0 + c.lane;
            const int e = l < N ? ldg(b.T + (size_t)pj * N + l) : -1;
            const unsigned long long bal = __ballot(e >= 0);
            const int r = cnt + __popcll(bal & ((1ull << c.lane) - 1ull));
            if (e >= 0 && r < kListCap) stg(b.plist + (size_t)pj * kListCap + r, (long long)(unsigned)e | ((long long)l << 32));
            if (bal) { if (lo == 0x7fffffff) lo = l0 + __ffsll((long long)bal) - 1; hi = l0 + 63 - __clzll((long long)bal); }
            cnt += __popcll(bal);
        }
        if (c.lane == 0) {
            if (cnt > kListCap) { atomicAdd(cap_ctr, 1); cnt = kListCap; }
            stg(b.pcount + pj, cnt); stg(b.pl_lo + pj, lo); stg(b.pl_hi + pj, hi);
        }
    }
    if (!grid_barrier(c)) { if (c.tid == 0 && c.wg == 0) b.report[0] = 1.0; return; }
    barriers += 1;
    if (ldg(cap_ctr) != 0) { if (gtid == 0) { b.report[0] = 2.0; b.report[1] = (double)P; b.report[2] = (double)N; } return; }
    const int dups = ldg(dup_ctr);
    stamp(c, 3);
    {
        int32_t *cursor = fresh_counter(c), *overflow = fresh_counter(c);
        build_hits(c, cursor, overflow);
        if (!grid_barrier(c)) { if (c.tid == 0 && c.wg == 0) b.report[0] = 1.0; return; }
        barriers += 1;
        if (ldg(overflow) != 0) { if (gtid == 0) { b.report[0] = 2.0; b.report[1] = (double)P; b.report[2] = (double)N; b.report[3] = -1.0; } return; }
    }
    stamp(c, 8);

    // ---- the passes ---------------------------------------------------------------------------------------------------------------
    const double sgn = p.damping == MQS_SBA_DAMPING_MARQUARDT ? 1.0 : -1.0;
    double *poses_cur = b.poses_a, *poses_new = b.poses_b, *pts_cur = b.pts_a, *pts_new = b.pts_b;
    int passes = 0, dropped = 0, trials = 0, lm_iters = 0;
    bool screened = !(p.gross_px > 0.0), dirty = false, have_before = false, have_start_cost = false;
    double start_cost = 0.0;
    double cost_before = 0.0, cost_after = 0.0;
    bool failed = false;
    for (;;) {
        if (dirty) {
            // the adjustment is redone from where it started
            for (int j = gtid; j < 12 * P; j += gthreads) stg(poses_cur + j, ldg(b.poses_init + j));
            for (int l = gtid; l < 3 * N; l += gthreads) stg(pts_cur + l, ldg(b.pts_init + l));
            if (!grid_barrier(c)) { failed = true; break; }
            barriers += 1;
            dirty = false;
        }
        stage_cams(c, poses_cur, lds_cam());
        if (!screened) {
            // before anything is adjusted: an observation that misses the current estimate by tens of pixels is not noise an
            // adjustment averages out; a landmark at a camera centre has no depth to adjust (slam_device.py, round 4)
            screened = true;
            double mine = phase_worst(c, pts_cur, lds_cam(), true, true);       // ... and the cost, should nothing be dropped
            if (c.wg == c.G - 1) mine += extras_cost(c, lds_cam());
            if (c.tid == 0) stg(b.partials + 4 * c.wg, mine);
            if (!grid_barrier(c)) { failed = true; break; }
            barriers += 1;
            // the median depth of the landmarks in use that have one, by rank counting against a copy in LDS (NaN = not a candidate):
            // every workgroup for itself while the map is small, each its share of the candidates (and one more barrier) beyond
            const bool local_median = N <= 4096;
            double med = NAN;
            if (local_median) {
                // every workgroup for itself: the candidates (+inf for the rest, padded to a power of two) sorted in LDS by a bitonic
                // network -- 55 compare-exchange stages at 1 024 entries; ranking every candidate against every other was 28 us
                double *sZ = lds_cam_new();
                int n2 = 64;
                while (n2 < N) n2 *= 2;
                __syncthreads();
                int mine_cnt = 0;
                for (int j = c.tid; j < n2; j += kT) {
                    double z = HUGE_VAL;
                    if (j < N) { const double zz = ldg(b.zmin + j); if (ldg(b.use + j) != 0 && isfinite(zz)) { z = zz; mine_cnt += 1; } }
                    sZ[j] = z;
                }
                const int cnt = (int)(block_sum(c, (double)mine_cnt) + 0.5);
                for (int k2 = 2; k2 <= n2; k2 <<= 1)
                    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                        __syncthreads();
                        for (int i = c.tid; i < n2; i += kT) {
                            const int ix = i ^ j2;
                            if (ix > i) {
                                const double a = sZ[i], bq = sZ[ix];
                                const bool up = (i & k2) == 0;
                                if ((a > bq) == up) { sZ[i] = bq; sZ[ix] = a; }
                            }
                        }
                    }
                __syncthreads();
                if (cnt > 0) med = 0.5 * (sZ[(cnt - 1) / 2] + sZ[cnt / 2]);
                __syncthreads();
            } else {
                double *sZ = lds_cam_new();
                constexpr int kChunk = kMaxPoses * kCamStride;
                for (int i0 = gtid; i0 < ((N + kT - 1) / kT) * kT; i0 += gthreads) {
                    const int ii = i0;
                    double zi = NAN;
                    if (ii < N) { const double z = ldg(b.zmin + ii); if (ldg(b.use + ii) != 0 && isfinite(z)) zi = z; }
                    int rank = 0, cnt = 0;
                    for (int j0 = 0; j0 < N; j0 += kChunk) {
                        const int m = (N - j0) < kChunk ? (N - j0) : kChunk;
                        __syncthreads();
                        for (int j = c.tid; j < m; j += kT) {
                            const double z = ldg(b.zmin + j0 + j);
                            sZ[j] = (ldg(b.use + j0 + j) != 0 && isfinite(z)) ? z : (double)NAN;
                        }
                        __syncthreads();
#pragma unroll 8
                        for (int j = 0; j < m; ++j) {
                            const double zj = sZ[j];
                            cnt += (zj == zj) ? 1 : 0;
                            rank += (zj < zi || (zj == zi && j0 + j < ii)) ? 1 : 0;
                        }
                    }
                    if (zi == zi) {
                        if (rank == (cnt - 1) / 2) stg(b.med + 0, zi);
                        if (rank == cnt / 2) stg(b.med + 1, zi);
                    }
                }
                if (!grid_barrier(c)) { failed = true; break; }
                barriers += 1;
                med = 0.5 * (ldg(b.med + 0) + ldg(b.med + 1));
            }
            // (NaN while no landmark has a depth: no depth screen then)
            int32_t *gross_ctr = fresh_counter(c);
            for (int l = gtid; l < N; l += gthreads) {
                if (ldg(b.use + l) == 0 || l < c.n0) continue;
                const double wv = ldg(b.worst + l), zv = ldg(b.zmin + l);
                const bool gross = !(wv <= p.gross_px) || (isfinite(zv) && zv < p.min_depth_ratio * med);
                if (gross) { stg(b.bad + l, 1); stg(b.use + l, 0); atomicAdd(gross_ctr, 1); }
            }
            int ng = 0;
            if (!count_barrier(c, gross_ctr, ng)) { failed = true; break; }
            barriers += 1;
            if (ng > 0) { dropped += ng; continue; }
            start_cost = sum_partials(c);
            have_start_cost = true;
        }
        // Levenberg-Marquardt, GTSAM 3.2.1's default schedule (ba_sparse.hip: mqs_sba_optimize_lm_dev)
        stamp(c, 4);
        double cur = 0.0;
        if (have_start_cost) { cur = start_cost; have_start_cost = false; }      // the screen's sweep has it
        else {
            double mine = phase_cost(c, pts_cur, lds_cam());
            if (c.wg == c.G - 1) mine += extras_cost(c, lds_cam());
            if (!reduce_cost(c, mine, cur)) { failed = true; break; }
            barriers += 1;
        }
        if (!have_before) { cost_before = cur; have_before = true; }
        // The iterations run in LEGS of at most `screen_iters` (0: one leg of max_iterations -- the adjustment as round 5 ran it).  Behind a
        // leg that has neither met the stop rule nor used the iterations up, the residual screen looks at the estimate as it stands: a
        // mistracked corner shows after three or four iterations (its residual stays at tens of pixels while the cost is still falling by
        // per cent), and a pass that carries one used to run to the iteration cap before the screen behind it threw the pass away --
        // ten trials where four tell the same.  Outliers found: retired, the adjustment redone from its start (a pass, as behind the
        // final screen); none: the next leg goes on from the estimate, its damping back at the initial value (GTSAM: a second optimize()).
        int nh = 1;                                           // accepted iterations of this pass + 1, over its legs
        bool early_out = false;
        for (;;) {
            double lam = p.lam0;
            const int left = p.max_iterations - (nh - 1);
            const int leg_cap = (p.screen_iters > 0 && p.screen_iters < left) ? p.screen_iters : left;
            bool ended = false;                               // the stop rule was met, or no trial at any damping was accepted
            for (int it = 0; it < leg_cap && !failed; ++it) {
                bool improved = false;
                double fresh = 0.0;
                while (lam <= p.lam_upper) {
                    bool ok = true;
                    if (!lm_trial(c, poses_cur, pts_cur, poses_new, pts_new, sgn * lam, true, fresh, ok, barriers)) { failed = true; break; }
                    trials += 1;
                    if (!ok) fresh = HUGE_VAL;
                    if (fresh <= cur) {
                        double *t1 = poses_cur; poses_cur = poses_new; poses_new = t1;
                        double *t2 = pts_cur; pts_cur = pts_new; pts_new = t2;
                        lam = lam / p.lam_factor;
                        if (lam < 1e-20) lam = 1e-20;
                        improved = true;
                        break;
                    }
                    lam *= p.lam_factor;
                }
                if (failed || !improved) { ended = true; break; }
                nh += 1;
                const double dec = fabs(cur - fresh);
                const bool done = dec < p.abs_tol || dec / (cur > 1e-300 ? cur : 1e-300) < p.rel_tol;
                cur = fresh;
                if (done) { ended = true; break; }
            }
            if (failed || ended || nh - 1 >= p.max_iterations || p.screen_iters <= 0) break;
            if (passes + 1 >= p.max_passes) continue;         // (no pass left to redo the adjustment in: the legs just go on)
            stage_cams(c, poses_cur, lds_cam());
            phase_worst(c, pts_cur, lds_cam(), false);
            if (!grid_barrier(c)) { failed = true; break; }
            barriers += 1;
            int32_t *early_ctr = fresh_counter(c);
            for (int l = gtid; l < N; l += gthreads) {
                const bool cand = ldg(b.use + l) != 0 && l >= c.n0 && !(ldg(b.worst + l) <= p.outlier_px);
                stg(b.cand + l, cand ? 1 : 0);
                if (cand) atomicAdd(early_ctr, 1);
            }
            int ne = 0;
            if (!count_barrier(c, early_ctr, ne)) { failed = true; break; }
            barriers += 1;
            if (ne > 0) {
                for (int l = gtid; l < N; l += gthreads)
                    if (ldg(b.cand + l) != 0) { stg(b.bad + l, 1); stg(b.use + l, 0); }
                dropped += ne;
                early_out = true;
                break;
            }
            if (c.ctr_next >= kCtr - 8) break;
        }
        if (failed) break;
        if (early_out) { passes += 1; dirty = true; if (c.ctr_next >= kCtr - 8) break; continue; }
        if (failed) break;
        lm_iters = nh - 1;
        cost_after = cur;
        if (nh == 1 && passes + 1 < p.max_passes) {
            // no trial at any damping was accepted: if that is cheirality -- a first step that puts a landmark behind one of its
            // cameras -- the landmark shows at the trial estimate of the smallest damping and sits out; redone without it
            double fr; bool ok;
            if (!lm_trial(c, poses_cur, pts_cur, poses_new, pts_new, sgn * p.lam0, false, fr, ok, barriers)) { failed = true; break; }
            trials += 1;
            phase_worst(c, pts_new, lds_cam_new(), false);
            if (!grid_barrier(c)) { failed = true; break; }
            barriers += 1;
            int32_t *flip_ctr = fresh_counter(c);
            for (int l = gtid; l < N; l += gthreads) {
                if (ldg(b.use + l) == 0 || l < c.n0) continue;
                if (isinf(ldg(b.worst + l))) { stg(b.bad + l, 1); stg(b.use + l, 0); atomicAdd(flip_ctr, 1); }
            }
            int nf = 0;
            if (!count_barrier(c, flip_ctr, nf)) { failed = true; break; }
            barriers += 1;
            if (nf > 0) { dropped += nf; passes += 1; continue; }
        }
        passes += 1;
        stamp(c, 5);
        // the screen behind an adjustment: a landmark whose worst residual exceeds the bound is a mistracked corner
        stage_cams(c, poses_cur, lds_cam());
        phase_worst(c, pts_cur, lds_cam(), false);
        if (!grid_barrier(c)) { failed = true; break; }
        barriers += 1;
        int32_t *bad_ctr = fresh_counter(c);
        for (int l = gtid; l < N; l += gthreads) {
            const bool cand = ldg(b.use + l) != 0 && l >= c.n0 && !(ldg(b.worst + l) <= p.outlier_px);
            stg(b.cand + l, cand ? 1 : 0);
            if (cand) atomicAdd(bad_ctr, 1);
        }
        int nb = 0;
        if (!count_barrier(c, bad_ctr, nb)) { failed = true; break; }
        barriers += 1;
        if (passes >= p.max_passes || nb == 0) break;
        for (int l = gtid; l < N; l += gthreads)
            if (ldg(b.cand + l) != 0) { stg(b.bad + l, 1); stg(b.use + l, 0); }
        dropped += nb;
        dirty = true;
        if (c.ctr_next >= kCtr - 8) break;
    }
    if (failed) { if (c.tid == 0 && c.wg == 0) b.report[0] = 1.0; return; }

    stamp(c, 6);
    // ---- the result back into the live state (slam2.py:19: landmarks are float32 values) ------------------------------------------
    int n_used = 0, n_obs = 0;
    for (int l = gtid; l < N; l += gthreads) {
        if (ldg(b.use + l) == 0) continue;
        for (int k = 0; k < 3; ++k) d.map[3 * l + k] = (double)(float)ldg(pts_cur + 3 * l + k);
    }
    for (int j = gtid; j < P; j += gthreads) {
        double q[12], M[12];
        for (int k = 0; k < 12; ++k) q[k] = ldg(poses_cur + 12 * j + k);
        pose12_to_w2c(q, M);
        const int tj = pose_of(c, j), row = tj - p.sel0;
        for (int k = 0; k < 12; ++k) d.traj[12 * (size_t)tj + k] = M[k];
        if (row < b.poses_host_cap) for (int k = 0; k < 12; ++k) b.poses_host[12 * (size_t)row + k] = M[k];
        if (j == P - 1) for (int k = 0; k < 12; ++k) d.pose_prev[k] = M[k];
        if (tj == p.key_pose) for (int k = 0; k < 12; ++k) d.pose_key[k] = M[k];
    }
    if (!p.identity) {
        // the accepted frames behind sel[0] that are not poses of this problem: each keeps its pose RELATIVE to the last problem pose
        // in front of it (M_j <- M_j inv(A_old) A_new, A = that pose's [R | t] before / after this adjustment) -- or stays (carry off);
        // either way its row goes to the host with the others
        for (int tj = p.sel0 + gtid; tj < p.P_all; tj += gthreads) {
            const int a = anchor_of(c, tj);
            if (lds_sel()[a] == tj) continue;
            double M[12];
            for (int k = 0; k < 12; ++k) M[k] = d.traj[12 * (size_t)tj + k];
            if (p.carry) {
                double Ao[12], q[12], An[12], Rr[9], tr[3];
                for (int k = 0; k < 12; ++k) { Ao[k] = ldg(b.traj_old + 12 * a + k); q[k] = ldg(poses_cur + 12 * a + k); }
                pose12_to_w2c(q, An);
                // D = M inv(Ao): rotation R_m R_o^T, translation t_m - D_R t_o; then M' = D An
                for (int r = 0; r < 3; ++r)
                    for (int s2 = 0; s2 < 3; ++s2) Rr[3 * r + s2] = M[4 * r] * Ao[4 * s2] + M[4 * r + 1] * Ao[4 * s2 + 1] + M[4 * r + 2] * Ao[4 * s2 + 2];
                for (int r = 0; r < 3; ++r) tr[r] = M[4 * r + 3] - (Rr[3 * r] * Ao[3] + Rr[3 * r + 1] * Ao[7] + Rr[3 * r + 2] * Ao[11]);
                for (int r = 0; r < 3; ++r) {
                    for (int s2 = 0; s2 < 3; ++s2) M[4 * r + s2] = Rr[3 * r] * An[s2] + Rr[3 * r + 1] * An[4 + s2] + Rr[3 * r + 2] * An[8 + s2];
                    M[4 * r + 3] = Rr[3 * r] * An[3] + Rr[3 * r + 1] * An[7] + Rr[3 * r + 2] * An[11] + tr[r];
                }
                for (int k = 0; k < 12; ++k) d.traj[12 * (size_t)tj + k] = M[k];
            }
            const int row = tj - p.sel0;
            if (row < b.poses_host_cap) for (int k = 0; k < 12; ++k) b.poses_host[12 * (size_t)row + k] = M[k];
        }
    }
    if (c.wg == 0) {
        for (int l = c.tid; l < N; l += kT)
            if (ldg(b.use + l) != 0) { n_used += 1; n_obs += ldg(b.per_lm + l); }
        const double su = block_sum(c, (double)n_used), so = block_sum(c, (double)n_obs);
        if (c.tid == 0) {
            double *r = b.report;
            r[0] = 0.0; r[1] = (double)P; r[2] = (double)N; r[3] = su; r[4] = so; r[5] = (double)passes; r[6] = (double)dropped;
            r[7] = (double)lm_iters; r[8] = cost_before; r[9] = cost_after; r[10] = (double)trials; r[11] = (double)dups;
            r[12] = (double)p.n_odo; r[13] = (double)barriers; r[14] = (double)p.sel0; r[15] = (double)(p.P_all - p.sel0);
        }
        stamp(c, 7);
    }
}

}  // namespace

// the adjuster's resident state
struct mqs_slam_ba {
    char *arena;
    size_t arena_bytes;
    BaDev dev;
    int n_odo;
    double *host;                       // pinned: report [MQS_SLAM_BA_REPORT], the selection [kMaxPoses int32], poses [host_rows][12] (the kernel writes report and poses,
    double *host_dev;                   // reads the selection; host_dev: the device's view of it)
    int host_rows;
    int fail_next;                      // test hook (mqs_debug_slam_ba_fail_next): the next adjustment returns this code, nothing launched
    uint32_t *sync_base;                // two 64-byte blocks of barrier words, used alternately (parity)
    int parity;
    char *fixed;                        // the part of the state that lives across re-allocations of the arena
    long long *stamps;                  // device, allocated by mqs_debug_slam_ba_stamps
};

namespace {

constexpr size_t up256(size_t v) { return (v + 255) & ~size_t(255); }
constexpr size_t kHostPoses = MQS_SLAM_BA_REPORT + kMaxPoses / 2;      // doubles in front of the pose rows of the pinned block: report, selection

int ba_fixed_alloc(mqs_slam *s)
{
    mqs_slam_ba *ba = new (std::nothrow) mqs_slam_ba();
    if (!ba) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
    const size_t L = (size_t)s->p.max_landmarks;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = up256(off + bytes); return o; };
    const size_t o_sync = take(128), o_ctr = take(kCtr * 4), o_rep = take(MQS_SLAM_BA_REPORT * 8), o_p0 = take(96), o_o0 = take(kMaxTracks * 24),
                 o_bad = take(L * 4), o_of = take(kMaxEdges * 4), o_ot = take(kMaxEdges * 4), o_om = take(kMaxEdges * 96),
                 o_ei = take((size_t)s->d.traj_cap * 4), o_eo = take((size_t)s->d.traj_cap * 4), o_er = take((size_t)kMaxEdges * kERec * 8), o_pr = take(256),
                 o_to = take(kMaxPoses * 96),
                 o_med = take(64), o_part = take(kMaxGroups * 32), o_dp = take((6 * kMaxPoses + 64) * 8), o_ys = take((6 * kMaxPoses + 64) * 8),
                 o_pa = take(kMaxPoses * 96), o_pb = take(kMaxPoses * 96), o_pi = take(kMaxPoses * 96),
                 o_pc = take(kMaxPoses * 4), o_lo = take(kMaxPoses * 4), o_hi = take(kMaxPoses * 4),
                 o_pl = take((size_t)kMaxPoses * kListCap * 8), o_bo = take((size_t)kMaxPoses * (kMaxPoses + 1) / 2 * 4),
                 o_bc = take((size_t)kMaxPoses * (kMaxPoses + 1) / 2 * 4), o_lr = take((size_t)s->d.log_cap * 4), o_rec = take((size_t)s->d.log_cap * kRec * 8);
    hipError_t e = hipMalloc((void **)&ba->fixed, off);
    if (e != hipSuccess) { delete ba; mqs_set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); return MQS_E_NOMEM; }
    ba->host_rows = 1024; ba->fail_next = 0;
    e = hipHostMalloc((void **)&ba->host, (kHostPoses + 12 * (size_t)ba->host_rows) * 8, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(ba->fixed); delete ba; mqs_set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return MQS_E_NOMEM; }
    char *a = ba->fixed;
    BaDev &b = ba->dev;
    ba->sync_base = (uint32_t *)(a + o_sync); ba->parity = 0;
    e = hipHostGetDevicePointer((void **)&ba->host_dev, ba->host, 0);
    if (e == hipSuccess) e = hipMemsetAsync(ba->sync_base, 0, 128, s->stream);
    if (e != hipSuccess) { (void)hipFree(ba->fixed); (void)hipHostFree(ba->host); delete ba; mqs_set_error("mqs_slam_bundle_adjust: %s", hipGetErrorString(e)); return MQS_E_HIP; }
    (void)o_rep;
    b.sync = ba->sync_base; b.sync_next = ba->sync_base + 16; b.poses_host = ba->host_dev + kHostPoses; b.poses_host_cap = 0;
    b.sel_host = reinterpret_cast<const int32_t *>(ba->host_dev + MQS_SLAM_BA_REPORT); b.traj_old = (double *)(a + o_to);
    b.ctr = (int32_t *)(a + o_ctr); b.report = ba->host_dev; b.pose0 = (double *)(a + o_p0);
    b.objp0 = (double *)(a + o_o0); b.bad = (int32_t *)(a + o_bad); b.odo_from = (int32_t *)(a + o_of); b.odo_to = (int32_t *)(a + o_ot);
    b.odo_meas = (double *)(a + o_om); b.e_in = (int32_t *)(a + o_ei); b.e_out = (int32_t *)(a + o_eo); b.erec = (double *)(a + o_er);
    b.prec = (double *)(a + o_pr); b.med = (double *)(a + o_med); b.partials = (double *)(a + o_part); b.dpose = (double *)(a + o_dp); b.ysol = (double *)(a + o_ys);
    b.poses_a = (double *)(a + o_pa); b.poses_b = (double *)(a + o_pb); b.poses_init = (double *)(a + o_pi);
    b.pcount = (int32_t *)(a + o_pc); b.pl_lo = (int32_t *)(a + o_lo); b.pl_hi = (int32_t *)(a + o_hi); b.plist = (long long *)(a + o_pl);
    b.lm_res = (int32_t *)(a + o_lr); b.rec = (double *)(a + o_rec); b.rec_stride = s->d.log_cap; b.blk_off = (int32_t *)(a + o_bo); b.blk_cnt = (int32_t *)(a + o_bc);
    b.N_cap = 0; b.P_cap = 0; b.ntile_cap = 0; b.n0 = 0;
    ba->arena = nullptr; ba->arena_bytes = 0; ba->n_odo = 0; ba->stamps = nullptr; b.stamps = nullptr;
    // retired flags 0, no edges
    e = hipMemsetAsync(b.bad, 0, L * 4, s->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b.e_in, 0xff, (size_t)s->d.traj_cap * 4, s->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b.e_out, 0xff, (size_t)s->d.traj_cap * 4, s->stream);
    if (e != hipSuccess) { (void)hipFree(ba->fixed); (void)hipHostFree(ba->host); delete ba; mqs_set_error("mqs_slam_bundle_adjust: %s", hipGetErrorString(e)); return MQS_E_HIP; }
    s->ba = ba;
    return MQS_OK;
}

// the arrays whose size follows the problem (landmarks, poses): grown geometrically; growing waits for the stream
int ba_reserve(mqs_slam *s, int P, int N_ub)
{
    mqs_slam_ba *ba = s->ba;
    const int nt = (6 * P + 1 + TB - 1) / TB;
    const int ntiles = nt * (nt + 1) / 2;
    if (ba->arena && N_ub <= ba->dev.N_cap && P <= ba->dev.P_cap && ntiles <= ba->dev.ntile_cap) return MQS_OK;
    int N_cap = ba->dev.N_cap > 0 ? ba->dev.N_cap : 2048;
    while (N_cap < N_ub) N_cap *= 2;
    int P_cap = ba->dev.P_cap > 0 ? ba->dev.P_cap : 64;
    while (P_cap < P) P_cap *= 2;
    if (P_cap > kMaxPoses) P_cap = kMaxPoses;
    const int nt_cap = (6 * P_cap + 1 + TB - 1) / TB, ntile_cap = nt_cap * (nt_cap + 1) / 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = up256(off + bytes); return o; };
    const size_t o_T = take((size_t)P_cap * N_cap * 4), o_pl = take((size_t)N_cap * 4), o_use = take((size_t)N_cap * 4), o_mn = take((size_t)N_cap * 4),
                 o_mx = take((size_t)N_cap * 4), o_cd = take((size_t)N_cap * 4), o_oc = take((size_t)N_cap * 4), o_a = take((size_t)N_cap * 24), o_b = take((size_t)N_cap * 24),
                 o_i = take((size_t)N_cap * 24), o_w = take((size_t)N_cap * 8), o_z = take((size_t)N_cap * 8),
                 o_S = take((size_t)ntile_cap * TB * TB * 8), o_L = take((size_t)ntile_cap * TB * TB * 8);
    // hit lists: a landmark seen from k poses has k (k - 1) / 2 of them; ~300 tracks per frame living ~30 frames: 4 500 P
    const long long hits_cap = (long long)160 * P_cap * P_cap > 262144 ? (long long)160 * P_cap * P_cap : 262144;
    const size_t o_h = take((size_t)hits_cap * 8);
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (ba->arena) { (void)hipFree(ba->arena); ba->arena = nullptr; }
    hipError_t e = hipMalloc((void **)&ba->arena, off);
    if (e != hipSuccess) { ba->arena = nullptr; ba->dev.N_cap = 0; ba->dev.P_cap = 0; mqs_set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); return MQS_E_NOMEM; }
    ba->arena_bytes = off;
    char *a = ba->arena;
    BaDev &b = ba->dev;
    b.T = (int32_t *)(a + o_T); b.per_lm = (int32_t *)(a + o_pl); b.use = (int32_t *)(a + o_use); b.pmin = (int32_t *)(a + o_mn); b.pmax = (int32_t *)(a + o_mx);
    b.cand = (int32_t *)(a + o_cd); b.out_cnt = (int32_t *)(a + o_oc); b.pts_a = (double *)(a + o_a); b.pts_b = (double *)(a + o_b); b.pts_init = (double *)(a + o_i);
    b.worst = (double *)(a + o_w); b.zmin = (double *)(a + o_z); b.S = (double *)(a + o_S); b.Lp = (double *)(a + o_L);
    b.hits = (long long *)(a + o_h); b.hits_cap = hits_cap;
    b.N_cap = N_cap; b.P_cap = P_cap; b.ntile_cap = ntile_cap;
    return MQS_OK;
}

mqs_lds_opt_in g_ba_lds_opt;

__global__ void slam_ba_anchor_kernel(BaDev b, SlamDev d, int n0)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * n0) b.objp0[t] = d.map[t];           // the start-up landmarks as given (exact float32 values)
    if (t == 0) {
        double q[12];
        w2c_to_pose12(d.traj, q);
        for (int k = 0; k < 12; ++k) b.pose0[k] = q[k];
    }
}

}  // namespace

void mqs_slam_ba_release(mqs_slam *s)
{
    if (!s || !s->ba) return;
    if (s->ba->arena) (void)hipFree(s->ba->arena);
    if (s->ba->fixed) (void)hipFree(s->ba->fixed);
    if (s->ba->host) (void)hipHostFree(s->ba->host);
    if (s->ba->stamps) (void)hipFree(s->ba->stamps);
    delete s->ba;
    s->ba = nullptr;
}

// slam_frame.hip calls this at the end of mqs_slam_start (the log is on): the gauge anchors are taken while the start-up
// landmarks and the first pose still have their given / first values
int mqs_slam_ba_anchor(mqs_slam *s, int n0)
{
    if (!s->d.log_lm) return MQS_OK;
    if (!s->ba) { const int rc = ba_fixed_alloc(s); if (rc != MQS_OK) return rc; }
    s->ba->dev.n0 = n0;
    { const int rc = ba_reserve(s, 1, n0); if (rc != MQS_OK) return rc; }      // the first sizes (64 poses, 2 048 landmarks) now, not inside the loop
    hipLaunchKernelGGL(slam_ba_anchor_kernel, dim3((3 * n0 + 255) / 256), dim3(256), 0, s->stream, s->ba->dev, s->d, n0);
    // The adjuster's first launch in a process costs ~8 ms beyond its work (the queue's scratch for its few spilled registers, the
    // 112 KB LDS opt-in): paid here, at start-up, by a launch that returns at once (zero poses: status 2, nothing touched), not behind
    // the run's first keyframe
    {
        BaParams p;
        memset(&p, 0, sizeof(p));
        p.G = 1;
        p.identity = 1;
        MQS_HIP_CHECK(mqs_lds_opt_in_once(g_ba_lds_opt, reinterpret_cast<const void *>(slam_ba_kernel), kLdsBytes));
        hipLaunchKernelGGL(slam_ba_kernel, dim3(1), dim3(kT), kLdsBytes, s->stream, s->ba->dev, s->d, p);
    }
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

namespace {

// How many workgroups of the adjuster can be RESIDENT at once on this device: its grid-wide barriers spin, so a workgroup that is
// never dispatched is a launch that waits 2 s and gives up.  One workgroup per CU (the LDS enforces it); cached per device.
int ba_resident_groups(int device, int *out)
{
    static int cached[64];
    if (device >= 0 && device < 64 && cached[device] > 0) { *out = cached[device]; return MQS_OK; }
    hipDeviceProp_t prop;
    MQS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    int per_cu = 0;
    MQS_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, slam_ba_kernel, kT, kLdsBytes));
    const long long g = (long long)per_cu * prop.multiProcessorCount;
    *out = g > kMaxGroups ? kMaxGroups : (int)g;                      // (the kernel's own bound: its per-workgroup cost pieces)
    if (device >= 0 && device < 64) cached[device] = *out;
    return MQS_OK;
}

// one adjuster launch at a time per device in this process: two launches whose workgroups wait for each other's CUs would both spin
std::mutex g_ba_launch_mutex[64];

int ba_grow_host(mqs_slam *s, int rows)
{
    mqs_slam_ba *ba = s->ba;
    if (rows <= ba->host_rows) return MQS_OK;
    int cap = ba->host_rows;
    while (cap < rows) cap *= 2;
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    double *h = nullptr, *hd = nullptr;
    hipError_t e = hipHostMalloc((void **)&h, (kHostPoses + 12 * (size_t)cap) * 8, hipHostMallocDefault);
    if (e != hipSuccess) { mqs_set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return MQS_E_NOMEM; }
    e = hipHostGetDevicePointer((void **)&hd, h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); mqs_set_error("hipHostGetDevicePointer failed: %s", hipGetErrorString(e)); return MQS_E_HIP; }
    (void)hipHostFree(ba->host);
    ba->host = h; ba->host_dev = hd; ba->host_rows = cap;
    ba->dev.report = hd; ba->dev.poses_host = hd + kHostPoses;
    ba->dev.sel_host = reinterpret_cast<const int32_t *>(hd + MQS_SLAM_BA_REPORT);
    return MQS_OK;
}

int ba_run(mqs_slam *s, const mqs_slam_ba_params *q, const mqs_slam_ba_window *w, double *report, double *poses_out, int32_t poses_cap)
{
    MQS_ARG_CHECK(s != nullptr && q != nullptr && report != nullptr, "handle, params, report must not be null");
    MQS_ARG_CHECK(s->started && s->log_arena != nullptr && s->ba != nullptr, "mqs_slam_log_enable before mqs_slam_start");
    MQS_ARG_CHECK(q->max_iterations >= 0 && q->max_iterations <= 100 && q->min_observations >= 1 && q->max_passes >= 1 && q->max_passes <= 16,
                  "max_iterations in [0, 100], min_observations >= 1, max_passes in [1, 16]");
    MQS_ARG_CHECK(q->point_sigma > 0.0 && q->pixel_sigma > 0.0 && q->outlier_px > 0.0 && q->lambda_factor > 1.0 && q->lambda_initial > 0.0, "sigmas, bounds, LM parameters");
    MQS_ARG_CHECK(poses_out == nullptr || poses_cap >= 0, "poses_cap >= 0");
    for (int k = 0; k < 6; ++k) MQS_ARG_CHECK(q->pose_sigmas[k] > 0.0 && q->odometry_sigmas[k] > 0.0, "sigmas > 0");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    const int P_all = s->accepted;
    const bool windowed = w != nullptr && w->n_poses > 0;
    const int P = windowed ? w->n_poses : P_all;
    mqs_slam_ba *ba = s->ba;
    BaParams p;
    memset(&p, 0, sizeof(p));
    p.anchor2 = -1;
    if (windowed) {
        MQS_ARG_CHECK(w->poses != nullptr && w->n_poses <= P_all, "window: poses must not be null, n_poses <= accepted frames");
        MQS_ARG_CHECK(w->seen_outside_sigma >= 0.0, "window: seen_outside_sigma >= 0");
        if (P > kMaxPoses) {
            mqs_set_error("mqs_slam_bundle_adjust_window: %d poses selected; the resident adjuster takes at most %d", P, kMaxPoses);
            return MQS_E_ARG;
        }
        bool asc = w->poses[0] >= 0, has_key = false;
        for (int k = 0; k < P; ++k) {
            if (k > 0 && w->poses[k] <= w->poses[k - 1]) asc = false;
            if (w->poses[k] == s->key_pose) has_key = true;
        }
        MQS_ARG_CHECK(asc && w->poses[P - 1] == P_all - 1, "window: pose indices ascending, the last one = the last accepted frame");
        MQS_ARG_CHECK(has_key, "window: the base keyframe of the live tracks must be one of the poses");
        MQS_ARG_CHECK(w->second_anchor < P_all, "window: second_anchor < accepted frames");
        bool ident = true;
        for (int k = 0; k < P && ident; ++k) ident = w->poses[k] == k;
        p.identity = (ident && P == P_all) ? 1 : 0;
        p.sel0 = w->poses[0];
        p.anchor2 = w->second_anchor >= 0 ? w->second_anchor : -1;
        p.carry = w->carry_unselected != 0;
        p.out_prior_w = w->seen_outside_sigma > 0.0 ? 1.0 / (w->seen_outside_sigma * w->seen_outside_sigma) : 0.0;
    } else {
        if (P > kMaxPoses) {
            mqs_set_error("mqs_slam_bundle_adjust: %d accepted frames; the resident adjuster takes at most %d poses (mqs_slam_bundle_adjust_window selects them)", P, kMaxPoses);
            return MQS_E_ARG;
        }
        p.identity = 1;
    }
    if (q->add_odometry_edge) {
        MQS_ARG_CHECK(q->edge_from >= 0 && q->edge_from < q->edge_to && q->edge_to < P_all, "0 <= edge_from < edge_to < accepted frames");
        MQS_ARG_CHECK(ba->n_odo < kMaxEdges, "too many odometry edges");
        p.add_edge = 1; p.edge_from = q->edge_from; p.edge_to = q->edge_to;
    }
    if (ba->fail_next) {
        const int code = ba->fail_next;
        ba->fail_next = 0;
        mqs_set_error("mqs_slam_bundle_adjust: failure injected by mqs_debug_slam_ba_fail_next (%d); nothing was launched or written", code);
        return code;
    }
    int rc = ba_reserve(s, P, s->land_ub > 0 ? s->land_ub : 1);
    if (rc != MQS_OK) return rc;
    rc = ba_grow_host(s, P_all - p.sel0);
    if (rc != MQS_OK) return rc;
    int G = q->workgroups > 0 ? q->workgroups : (P > 96 ? 256 : 128);     // the pose-pair phases have P^2 / 2 tasks; the chain of block factors does not care
    const bool g_forced = q->workgroups > 0;
    if (const char *e = getenv("MQS_SLAM_BA_GROUPS")) { const int g = atoi(e); if (g > 0) G = g; }
    // every workgroup has to be resident (the barriers spin): the library's own choice is clamped to what the device holds, a
    // caller's explicit request beyond that is an error at once -- not a 2 s wait that gives up
    int resident = 0;
    rc = ba_resident_groups(s->device, &resident);
    if (rc != MQS_OK) return rc;
    if (resident < 1) { mqs_set_error("mqs_slam_bundle_adjust: no workgroup of the adjuster fits a compute unit of device %d (%zu bytes of LDS)", s->device, kLdsBytes); return MQS_E_ARG; }
    if (G > resident) {
        if (g_forced || getenv("MQS_SLAM_BA_GROUPS")) {
            mqs_set_error("mqs_slam_bundle_adjust: %d workgroups requested, device %d holds %d of them at once (one per compute unit: %zu bytes of LDS each); "
                          "the adjustment's grid-wide barriers need all of them resident", G, s->device, resident, kLdsBytes);
            return MQS_E_ARG;
        }
        G = resident;
    }
    p.P = P; p.P_all = P_all; p.G = G; p.key_pose = s->key_pose;
    p.max_iterations = q->max_iterations; p.min_observations = q->min_observations; p.max_passes = q->max_passes; p.damping = q->damping;
    p.screen_iters = q->screen_iterations > 0 ? q->screen_iterations : 0;
    p.n_odo = ba->n_odo + (p.add_edge ? 1 : 0);
    p.outlier_px = q->outlier_px; p.gross_px = q->gross_px; p.border = q->border_margin_px; p.min_depth_ratio = q->min_depth_ratio;
    p.prior_w = 1.0 / (q->point_sigma * q->point_sigma); p.sigma_px = q->pixel_sigma; p.isigma_px = 1.0 / q->pixel_sigma;
    for (int k = 0; k < 6; ++k) {
        p.pose_w[k] = 1.0 / (q->pose_sigmas[k] * q->pose_sigmas[k]);
        p.odo_w[k] = 1.0 / (q->odometry_sigmas[k] * q->odometry_sigmas[k]);
    }
    p.lam0 = q->lambda_initial; p.lam_factor = q->lambda_factor; p.lam_upper = q->lambda_upper; p.abs_tol = q->abs_tol; p.rel_tol = q->rel_tol;
    p.W = s->p.W; p.H = s->p.H;
    MQS_HIP_CHECK(mqs_lds_opt_in_once(g_ba_lds_opt, reinterpret_cast<const void *>(slam_ba_kernel), kLdsBytes));
    // No fill or copy launch around the adjustment: the barrier words alternate between two blocks (a launch zeroes the next one's), the
    // report and the adjusted poses are written by the kernel into pinned host memory (NaN there: a launch that wrote no report), the
    // selection is read from it.
    const int rows = P_all - p.sel0;
    const int np = poses_out ? (rows < poses_cap ? rows : poses_cap) : 0;
    memset(ba->host, 0xff, MQS_SLAM_BA_REPORT * 8);
    if (!p.identity) memcpy(ba->host + MQS_SLAM_BA_REPORT, w->poses, (size_t)P * 4);
    {
        std::lock_guard<std::mutex> lock(g_ba_launch_mutex[(s->device >= 0 && s->device < 64) ? s->device : 0]);
        ba->dev.sync = ba->sync_base + 16 * ba->parity; ba->dev.sync_next = ba->sync_base + 16 * (1 - ba->parity);
        ba->dev.poses_host_cap = np;
        ba->parity ^= 1;
        hipLaunchKernelGGL(slam_ba_kernel, dim3(G), dim3(kT), kLdsBytes, s->stream, ba->dev, s->d, p);
        MQS_HIP_CHECK(hipGetLastError());
        MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    }
    memcpy(report, ba->host, MQS_SLAM_BA_REPORT * 8);
    if (np > 0) memcpy(poses_out, ba->host + kHostPoses, (size_t)np * 96);
    if (!(report[0] == 0.0)) {
        // (a launch that gave up leaves its barrier words as they were: both blocks start from zero again; an edge this call brought is
        // not kept -- the kernel may have returned before writing it)
        (void)hipMemsetAsync(ba->sync_base, 0, 128, s->stream);
        (void)hipStreamSynchronize(s->stream);
        ba->parity = 0;
        if (report[0] == 1.0) { mqs_set_error("mqs_slam_bundle_adjust: a grid-wide wait of the adjustment gave up (2 s); nothing was written back"); return MQS_E_TIMEOUT; }
        if (report[0] == 3.0) { mqs_set_error("mqs_slam_bundle_adjust: the observation log is full (%d entries): observations have been dropped", s->d.log_cap); return MQS_E_ARG; }
        mqs_set_error("mqs_slam_bundle_adjust: capacity (poses %g of %d, landmarks %g, a frame with more than %d observations, or more co-observations than the hit lists hold)", report[1], kMaxPoses, report[2], kListCap);
        return MQS_E_CAPACITY;
    }
    if (p.add_edge) ba->n_odo += 1;
    return MQS_OK;
}

__global__ __launch_bounds__(kT) void debug_factor32_kernel(const double *A, double *out, int32_t *bad, int form)
{
    __shared__ double sT[TB * TLD];
    __shared__ double sM[256];
    const int tid = threadIdx.x;
    for (int e = tid; e < TB * TB; e += kT) sT[(e >> 5) * TLD + (e & 31)] = A[e];
    __syncthreads();
    if (form == 0) mqs::chol::factor_diag_block_from_lds_4w<true>(sT, out, TB, 0, bad, tid, sM);
    else if (tid < 64) mqs::chol::factor_diag_block_mfma_wave<true>(sT, out, bad, tid, sM);
}

}  // namespace

extern "C" {

int mqs_slam_bundle_adjust(mqs_slam *s, const mqs_slam_ba_params *q, double *report, double *poses_out, int32_t poses_cap)
{
    return ba_run(s, q, nullptr, report, poses_out, poses_cap);
}

int mqs_slam_bundle_adjust_window(mqs_slam *s, const mqs_slam_ba_params *q, const mqs_slam_ba_window *w, double *report, double *poses_out, int32_t poses_cap)
{
    MQS_ARG_CHECK(w != nullptr, "window must not be null");
    return ba_run(s, q, w, report, poses_out, poses_cap);
}

int mqs_debug_slam_ba_fail_next(mqs_slam *s, int code)
{
    MQS_ARG_CHECK(s != nullptr && s->ba != nullptr, "handle (started with the log on)");
    MQS_ARG_CHECK(code == MQS_E_TIMEOUT || code == MQS_E_CAPACITY || code == 0, "code: MQS_E_TIMEOUT, MQS_E_CAPACITY or 0");
    s->ba->fail_next = code;
    return MQS_OK;
}

int mqs_slam_ba_resident_groups(mqs_slam *s, int32_t *groups)
{
    MQS_ARG_CHECK(s != nullptr && groups != nullptr, "handle, groups must not be null");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int g = 0;
    const int rc = ba_resident_groups(s->device, &g);
    *groups = g;
    return rc;
}

// Profiling hook: switches the phase stamps of the adjuster's launches on (the first call allocates them) and returns those of the
// LAST mqs_slam_bundle_adjust: out [cap][2] int64 = (phase id, ticks of the 100 MHz wall clock) as workgroup 0 passed them; *n = stamps
// taken.  Phase ids: 0 start, 1 tables cleared, 2 log in the table, 3 lists built, 4 a pass's LM begins, 10..19 inside a trial (records,
// barrier, system, barrier, Cholesky, back-substitution, barrier, landmarks + cost, cost reduced), 5 screen, 6 write-back, 7 end.
// test hook: the 32 x 32 diagonal-tile factor by itself (form 0: four wavefronts, a barrier per pivot; 1: one wavefront, panels of four
// pivots on the matrix pipe).  A, out: HOST, 32 x 32 row-major; out = L in the lower triangle, inv(L)'s strictly lower part transposed above it.
int mqs_debug_factor32(const double *A, double *out, int form, int32_t *not_positive_definite)
{
    MQS_ARG_CHECK(A != nullptr && out != nullptr && not_positive_definite != nullptr && (form == 0 || form == 1), "A, out, flag; form 0 or 1");
    double *dA = nullptr;
    MQS_HIP_CHECK(hipMalloc((void **)&dA, (2 * TB * TB + 8) * sizeof(double)));
    hipError_t e = hipMemcpy(dA, A, TB * TB * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dA + TB * TB, 0, (TB * TB + 8) * sizeof(double));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_factor32_kernel, dim3(1), dim3(kT), 0, nullptr, dA, dA + TB * TB, reinterpret_cast<int32_t *>(dA + 2 * TB * TB), form);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dA + TB * TB, TB * TB * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(not_positive_definite, dA + 2 * TB * TB, 4, hipMemcpyDeviceToHost);
    (void)hipFree(dA);
    MQS_HIP_CHECK(e);
    return MQS_OK;
}

int mqs_debug_slam_ba_stamps(mqs_slam *s, int64_t *out, int cap, int32_t *n)
{
    MQS_ARG_CHECK(s != nullptr && s->ba != nullptr && n != nullptr && cap >= 0, "handle (started with the log on), n");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n = 0;
    if (!s->ba->stamps) {
        MQS_HIP_CHECK(hipMalloc((void **)&s->ba->stamps, (2 * kStamps + 2) * sizeof(long long)));
        MQS_HIP_CHECK(hipMemset(s->ba->stamps, 0, (2 * kStamps + 2) * sizeof(long long)));
        s->ba->dev.stamps = s->ba->stamps;
        return MQS_OK;
    }
    long long cnt = 0;
    MQS_HIP_CHECK(hipMemcpy(&cnt, s->ba->stamps + 2 * kStamps, sizeof(cnt), hipMemcpyDeviceToHost));
    *n = (int32_t)cnt;
    const int m = cnt < cap ? (int)cnt : cap;
    if (m > 0 && out) MQS_HIP_CHECK(hipMemcpy(out, s->ba->stamps, (size_t)m * 16, hipMemcpyDeviceToHost));
    return MQS_OK;
}

// The odometry edges the adjuster holds (one per keyframe it stood behind): pose indices and the measured relative pose (camera-to-world
// pose12 of P_to inv(P_from)), for a caller that takes the adjustment over (slam_device.py beyond MQS_SLAM_BA_MAX_POSES frames).
int mqs_slam_read_ba_edges(mqs_slam *s, int32_t *from, int32_t *to, double *meas, int cap, int32_t *n)
{
    MQS_ARG_CHECK(s != nullptr && n != nullptr && cap >= 0, "handle, n");
    MQS_ARG_CHECK(s->ba != nullptr, "mqs_slam_log_enable before mqs_slam_start");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    *n = s->ba->n_odo;
    const int m = s->ba->n_odo < cap ? s->ba->n_odo : cap;
    if (m > 0) {
        MQS_ARG_CHECK(from && to && meas, "from, to, meas must not be null");
        MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
        MQS_HIP_CHECK(hipMemcpy(from, s->ba->dev.odo_from, (size_t)m * 4, hipMemcpyDeviceToHost));
        MQS_HIP_CHECK(hipMemcpy(to, s->ba->dev.odo_to, (size_t)m * 4, hipMemcpyDeviceToHost));
        MQS_HIP_CHECK(hipMemcpy(meas, s->ba->dev.odo_meas, (size_t)m * 96, hipMemcpyDeviceToHost));
    }
    return MQS_OK;
}

int mqs_slam_read_ba_flags(mqs_slam *s, uint8_t *retired, int cap, int32_t *n)
{
    MQS_ARG_CHECK(s != nullptr && n != nullptr && cap >= 0 && (cap == 0 || retired != nullptr), "handle, n, retired");
    MQS_ARG_CHECK(s->ba != nullptr, "mqs_slam_log_enable before mqs_slam_start");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int32_t cnt[C_COUNT];
    MQS_HIP_CHECK(hipMemcpyAsync(cnt, s->d.cnt, sizeof(cnt), hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n = cnt[C_NLAND];
    const int m = cnt[C_NLAND] < cap ? cnt[C_NLAND] : cap;
    if (m > 0) {
        int32_t *tmp = new (std::nothrow) int32_t[m];
        if (!tmp) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
        hipError_t e = hipMemcpyAsync(tmp, s->ba->dev.bad, (size_t)m * 4, hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e == hipSuccess)
            for (int k = 0; k < m; ++k) retired[k] = tmp[k] != 0;
        delete[] tmp;
        MQS_HIP_CHECK(e);
    }
    return MQS_OK;
}

}  // extern "C"
