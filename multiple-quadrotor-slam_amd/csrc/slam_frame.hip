// The per-frame loop of the reference resident on the device (gfx950): Work/SLAM/application/own/slam2.py
// handle_new_frame (:360-695) with the state -- live tracks, their base-keyframe positions, the map, the poses -- kept in
// device memory, so that a frame is ONE library call that enqueues
//     pyramids + Scharr derivatives -> pyramidal LK of the live tracks                          (:381)      features.hip
//     frame_filter_kernel: status / error filter, lost-tracks gate, >= 8 landmarks gate,        (:382-437)
//                          the 3D-2D correspondences of the triangulated tracks, RANSAC samples
//     all RANSAC hypotheses -> selection + inliers -> solvePnP on the inliers                   (:453-490)  pnp.hip
//     frame_decide_kernel: outlier-ratio and reprojection gates, the kept tracks committed,     (:461-522, 43-59)
//                          keyframe test (homography base keyframe -> frame, w0 / w2 > 1.04)
// reads back one 320-byte result block (decision, pose, counts) and, on a keyframe, enqueues WITHOUT waiting
//     keyframe_step_kernel: triangulate the free tracks, refine the pose, re-triangulate       (:541-590)  pnp.hip
//     keyframe_commit_kernel: new landmarks into the map (float32, :19), failed tracks dropped  (:589-612)
//     coverage mask -> goodFeaturesToTrack -> append_tracks_kernel: top-up, rebase             (:657-674)  features.hip
// The host-driven loop (slam_loop.py) spends 1.2-1.4 ms per frame on ~10 host-pointer calls for ~0.25 ms of kernels; here the
// host's share is one call and one wait.
//
// Round 6: WHEN the kernels of a frame are enqueued, and where.  With the next image named before the call (mqs_slam_set_next) the pyramid
// of the pair (this, next) and -- behind this frame's hypotheses -- its tracker run on a side stream, ahead of the next frame; with
// mqs_slam_pipeline the next frame's hypothesis and decision launches are enqueued behind this frame's decision before the call waits for
// this frame's result, and look at that decision on the device (they do nothing behind a keyframe or a rejected frame).  The decision kernel is
// two workgroups (pose refinement | keyframe test); the result block goes into a ring of pinned slots with a ticket the host polls for.  Both
// streams are created with the highest stream priority: hardware queues no default-priority stream of the process can land on.  Same kernels,
// same inputs, same results as the order above -- a frame takes ~100 us instead of ~160.
//
// Deviations from the reference, as in slam_loop.py (no image set / OpenCV run of the reference exists to compare with): the
// homography of the keyframe test is the normalised DLT over ALL kept tracks by default (cv2.findHomography on a random quarter of
// them there: mqs_slam_set_thresholds(.., max_homography_points) switches that on, drawn from the device generator), RANSAC samples come from a counter-based generator on the device (splitmix64 of seed, frame, hypothesis).
#include "mqs_common.h"
#include "pnp_math.h"
#include "pnp_block.h"
#include "cam_math.h"
#include "wave_reduce.h"
#include "slam_state.h"
#include <new>
#include <chrono>

namespace {

using namespace mqs::slamst;

// rank of this thread among the flagged threads of the workgroup (thread order), and their number; two barriers
__device__ __forceinline__ int block_rank(bool flag, int tid, int *sWave, int &total)
{
    const unsigned long long bal = __ballot(flag);
    const int lane = tid & 63, wave = tid >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) sWave[wave] = __popcll(bal);
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) off += sWave[w];
        total += sWave[w];
    }
    __syncthreads();
    return off + before;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long &x)
{
    unsigned long long z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------------------------------
// after LK: which tracks survive (slam2.py:382), the two early gates (:385-387, :437), the pose problem, the samples
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_filter_kernel(SlamDev d, SlamParams p)
{
    __shared__ int sWave[4];
    const int tid = threadIdx.x;
    const int n = d.cnt[C_N];
    int n_keep = 0;
    for (int b = 0; b < n; b += 256) {
        const int i = b + tid;
        // (the tracker ran ahead of this frame: its results lie in the frame-before's kept order, spec_map says where)
        const int ii = (p.spec && i < n) ? d.spec_map[i] : i;
        const float *lkp = p.spec ? d.lk_pts_b : d.lk_pts, *lke = p.spec ? d.lk_err_b : d.lk_err;
        const uint8_t *lks = p.spec ? d.lk_st_b : d.lk_st;
        const bool keep = i < n && lks[ii] == 1 && lke[ii] < (float)p.max_of_error;
        int total;
        const int r = n_keep + block_rank(keep, tid, sWave, total);
        if (keep) {
            d.t_pts[2 * r] = lkp[2 * ii]; d.t_pts[2 * r + 1] = lkp[2 * ii + 1];
            d.t_base[2 * r] = d.base[2 * i]; d.t_base[2 * r + 1] = d.base[2 * i + 1];
            d.t_lm[r] = d.lm[i]; d.t_tid[r] = d.tid[i];
        }
        n_keep += total;
    }
    __threadfence_block();
    __syncthreads();
    const double lost = n > 0 ? 1.0 - (double)n_keep / (double)n : 1.0;
    int n_tri = 0;
    for (int b = 0; b < n_keep; b += 256) {
        const int k = b + tid;
        const bool tri = k < n_keep && d.t_lm[k] >= 0;
        int total;
        const int j = n_tri + block_rank(tri, tid, sWave, total);
        if (tri) {
            const int l = d.t_lm[k];
            d.objp_t[3 * j] = d.map[3 * l]; d.objp_t[3 * j + 1] = d.map[3 * l + 1]; d.objp_t[3 * j + 2] = d.map[3 * l + 2];
            d.imgp_t[2 * j] = (double)d.t_pts[2 * k]; d.imgp_t[2 * j + 1] = (double)d.t_pts[2 * k + 1];
            d.tri_pos[j] = k;
        }
        n_tri += total;
    }
    int reason = 0;
    if (lost > p.max_lost_ratio) reason = 1;                       // "lost track of too many points"
    else if (n_tri < 8) reason = 2;                                // fewer than 8 triangulated tracks
    if (tid == 0) {
        d.cnt[C_NKEEP] = n_keep;
        d.cnt[C_NTRI] = reason ? 0 : n_tri;                        // 0 switches the pose kernels off
        d.res[R_DECISION] = reason ? 0.0 : -1.0;                   // -1: undecided
        d.res[R_REASON] = (double)reason;
        d.res[R_LOST] = lost;
        d.res[R_NTRI] = (double)n_tri;
    }
    if (reason) return;
    // minimal samples without replacement, one hypothesis per thread and pass.  x mod n_tri of the 64-bit draws through three exact
    // small remainders in double arithmetic (n_tri <= kMaxTracks: every operand below 2^33) -- the same values as the 64-bit `%`, whose
    // software division was ~150 instructions per pick on each thread's chain.
    const int frame = d.cnt[C_FRAME];
    const double inv_nt = 1.0 / (double)n_tri;
    auto small_mod = [&](unsigned long long y) {                   // y < 2^53
        long long r = (long long)y - (long long)((unsigned long long)((double)y * inv_nt)) * (long long)n_tri;
        r += r < 0 ? n_tri : 0;
        r -= r >= n_tri ? n_tri : 0;
        return (unsigned long long)r;
    };
    const unsigned long long two32_mod = small_mod(1ull << 32);
    for (int h = tid; h < kHyp; h += 256) {
        unsigned long long s = p.seed ^ ((unsigned long long)frame << 24) ^ ((unsigned long long)h * 0x632be59bd9b4e019ull);
        int pick[kSample];
#pragma unroll
        for (int j = 0; j < kSample; ++j) {
            int v;
            bool dup;
            do {
                const unsigned long long x = splitmix64(s);
                v = (int)small_mod(small_mod(x >> 32) * two32_mod + small_mod(x & 0xffffffffull));
                dup = false;
#pragma unroll
                for (int q = 0; q < kSample; ++q) dup = dup || (q < j && pick[q] == v);
            } while (dup);
            pick[j] = v;
        }
#pragma unroll
        for (int j = 0; j < kSample; ++j) d.samples[h * kSample + j] = pick[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same filter and the RANSAC hypotheses in ONE launch (round 5): every hypothesis' wavefront compacts the tracked points for
// itself into LDS -- 300 tracks are five ballots -- instead of waiting for a one-workgroup kernel to do it once (8 us + a launch gap on
// a frame of ~160), draws its own sample, and counts its inliers on its LDS copy.  Workgroup 0 alone writes the frame's state (the kept
// tracks, the pose problem, the counters, the result block): the same arrays frame_filter_kernel leaves.  MQS_SLAM_FUSED_FILTER=0 keeps
// the two launches (A/B, tests).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void frame_hypothesis_kernel(SlamDev d, SlamParams p)
{
    __shared__ double sI[9];
    __shared__ double sA[132];
    __shared__ double sObjAll[3 * kMaxTracks], sImgAll[2 * kMaxTracks];
    __shared__ double sObj[3 * 64], sImg[2 * 64];
    __shared__ int sPick[kSample];
    const int lane = threadIdx.x, h = blockIdx.x;
    const bool writer = h == 0;
    // enqueued before the frame in front was decided (mqs_slam_pipeline): that frame became a keyframe or was rejected -- nothing to do, the
    // host issues this frame again behind the keyframe's branch / from the right previous image
    if (p.gated && d.cnt[C_LAST_DECISION] != 1) return;
    const int n = d.cnt[C_N];
    if (lane < 9) sI[lane] = d.intr[lane];
    int n_keep = 0, n_tri = 0;
    for (int b = 0; b < n; b += 64) {
        const int i = b + lane;
        const int ii = (p.spec && i < n) ? d.spec_map[i] : i;           // (the tracker ran ahead of this frame: see frame_filter_kernel)
        const float *lkp = p.spec ? d.lk_pts_b : d.lk_pts, *lke = p.spec ? d.lk_err_b : d.lk_err;
        const uint8_t *lks = p.spec ? d.lk_st_b : d.lk_st;
        const bool keep = i < n && lks[ii] == 1 && lke[ii] < (float)p.max_of_error;
        const int l = keep ? d.lm[i] : -1;
        const bool tri = l >= 0;
        const unsigned long long km = __ballot(keep), tm = __ballot(tri), below = (1ull << lane) - 1ull;
        const int r = n_keep + __popcll(km & below), j = n_tri + __popcll(tm & below);
        if (keep) {
            const float px = lkp[2 * ii], py = lkp[2 * ii + 1];
            if (writer) {
                d.t_pts[2 * r] = px; d.t_pts[2 * r + 1] = py;
                d.t_base[2 * r] = d.base[2 * i]; d.t_base[2 * r + 1] = d.base[2 * i + 1];
                d.t_lm[r] = l; d.t_tid[r] = d.tid[i];
            }
            if (tri) {
                const double X = d.map[3 * l], Y = d.map[3 * l + 1], Z = d.map[3 * l + 2];
                sObjAll[3 * j] = X; sObjAll[3 * j + 1] = Y; sObjAll[3 * j + 2] = Z;
                sImgAll[2 * j] = (double)px; sImgAll[2 * j + 1] = (double)py;
                if (writer) {
                    d.objp_t[3 * j] = X; d.objp_t[3 * j + 1] = Y; d.objp_t[3 * j + 2] = Z;
                    d.imgp_t[2 * j] = (double)px; d.imgp_t[2 * j + 1] = (double)py;
                    d.tri_pos[j] = r;
                }
            }
        }
        n_keep += __popcll(km); n_tri += __popcll(tm);
    }
    const double lost = n > 0 ? 1.0 - (double)n_keep / (double)n : 1.0;
    int reason = 0;
    if (lost > p.max_lost_ratio) reason = 1;                       // "lost track of too many points"
    else if (n_tri < 8) reason = 2;                                // fewer than 8 triangulated tracks
    if (writer && lane == 0) {
        d.cnt[C_NKEEP] = n_keep;
        d.cnt[C_NTRI] = reason ? 0 : n_tri;                        // 0 switches the pose step off
        d.res[R_DECISION] = reason ? 0.0 : -1.0;                   // -1: undecided
        d.res[R_REASON] = (double)reason;
        d.res[R_LOST] = lost;
        d.res[R_NTRI] = (double)n_tri;
    }
    if (reason) {
        if (lane == 0) d.pnp_counts[h] = -1;
        return;
    }
    // this hypothesis' minimal sample without replacement (the generator and the picks of frame_filter_kernel)
    if (lane == 0) {
        const int frame = d.cnt[C_FRAME];
        const double inv_nt = 1.0 / (double)n_tri;
        auto small_mod = [&](unsigned long long y) {
            long long r = (long long)y - (long long)((unsigned long long)((double)y * inv_nt)) * (long long)n_tri;
            r += r < 0 ? n_tri : 0;
            r -= r >= n_tri ? n_tri : 0;
            return (unsigned long long)r;
        };
        const unsigned long long two32_mod = small_mod(1ull << 32);
        unsigned long long s = p.seed ^ ((unsigned long long)frame << 24) ^ ((unsigned long long)h * 0x632be59bd9b4e019ull);
        int pick[kSample];
#pragma unroll
        for (int j = 0; j < kSample; ++j) {
            int v;
            bool dup;
            do {
                const unsigned long long x = splitmix64(s);
                v = (int)small_mod(small_mod(x >> 32) * two32_mod + small_mod(x & 0xffffffffull));
                dup = false;
#pragma unroll
                for (int q = 0; q < kSample; ++q) dup = dup || (q < j && pick[q] == v);
            } while (dup);
            pick[j] = v;
        }
#pragma unroll
        for (int j = 0; j < kSample; ++j) { sPick[j] = pick[j]; d.samples[h * kSample + j] = pick[j]; }
    }
    mqs_wave_lds_sync();
    if (lane < kSample) {
        const int i = sPick[lane];
        sObj[3 * lane] = sObjAll[3 * i]; sObj[3 * lane + 1] = sObjAll[3 * i + 1]; sObj[3 * lane + 2] = sObjAll[3 * i + 2];
        sImg[2 * lane] = sImgAll[2 * i]; sImg[2 * lane + 1] = sImgAll[2 * i + 1];
    }
    mqs_wave_lds_sync();
    const mqs::pnpblk::Problem pr = {sObj, sImg, nullptr, 0, kSample};
    double P[12];
    const int count = mqs::pnpblk::hypothesis_wave(pr, sObjAll, sImgAll, n_tri, sI, kSampleIters, p.max_reproj * p.max_reproj, sA, lane, P);
    if (lane < 12) d.pnp_poses[12 * h + lane] = P[lane];
    if (lane == 0) d.pnp_counts[h] = count;
}

// ---------------------------------------------------------------------------------------------------------------------
// 9 x 9 symmetric eigen-decomposition by parallel cyclic Jacobi (one wavefront; lanes = 4 disjoint rotations x 9 rows), for the
// null vector of the homography system; 3 x 3 by a single lane for the singular values of the homography.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_angle(double app, double aqq, double apq, double &c, double &s)
{
    if (fabs(apq) <= 1e-300) { c = 1.0; s = 0.0; return; }
    // (reciprocal and reciprocal square root by Newton steps: the IEEE divisions and square roots were most of a rotation's chain)
    const double tau = (aqq - app) * mqs::rcp(2.0 * apq);
    const double q1 = fma(tau, tau, 1.0);
    const double t = fabs(tau) > 1e100 ? 0.5 * mqs::rcp(tau) : (tau >= 0.0 ? 1.0 : -1.0) * mqs::rcp(fabs(tau) + q1 * mqs::rsqrt_d(q1));
    c = mqs::rsqrt_d(fma(t, t, 1.0));
    s = t * c;
}

// A, V: LDS [9][9]; on return A's diagonal holds the eigenvalues, V's columns the eigenvectors.  Called by one full wavefront.
__device__ void jacobi9(double *A, double *V, int lane)
{
    const int g = lane / 9, k = lane - 9 * g;                      // rotation slot 0..3 (lanes 36..63 idle), row / column index
    const bool act = lane < 36;
    if (lane < 81) V[lane] = (lane % 10 == 0) ? 1.0 : 0.0;
    if (lane + 64 < 81) V[lane + 64] = ((lane + 64) % 10 == 0) ? 1.0 : 0.0;
    mqs_wave_lds_sync();
    for (int sweep = 0; sweep < 10; ++sweep) {
        // converged when the off-diagonal mass is at rounding level of the diagonal (usually after 5 or 6 sweeps)
        double off = 0.0, dia = 0.0;
        if (act) {
#pragma unroll
            for (int j = 0; j < 9; ++j) { const double a = A[k * 9 + j]; if (g == 0) { if (j == k) dia += a * a; else off += a * a; } }
        }
        mqs::wave::sum2(off, dia, off, dia);                       // (the butterfly's pairs, without its ds_bpermute round trips)
        if (off <= 1e-30 * dia) break;
        for (int r = 0; r < 9; ++r) {
            // round-robin round r of 9 players: slot g pairs (r + g + 1, r - g - 1) mod 9, player r rests
            int pa = (r + g + 1) % 9, pb = (r + 9 - g - 1) % 9;
            const int pp = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
            double c = 1.0, s = 0.0;
            if (act) jacobi_angle(A[pp * 9 + pp], A[q * 9 + q], A[pp * 9 + q], c, s);
            mqs_wave_lds_sync();
            if (act) {                                             // A <- A J: columns pp, q of row k
                const double akp = A[k * 9 + pp], akq = A[k * 9 + q];
                A[k * 9 + pp] = c * akp - s * akq;
                A[k * 9 + q] = s * akp + c * akq;
                const double vkp = V[k * 9 + pp], vkq = V[k * 9 + q];
                V[k * 9 + pp] = c * vkp - s * vkq;
                V[k * 9 + q] = s * vkp + c * vkq;
            }
            mqs_wave_lds_sync();
            if (act) {                                             // A <- J^T A: rows pp, q of column k
                const double apk = A[pp * 9 + k], aqk = A[q * 9 + k];
                A[pp * 9 + k] = c * apk - s * aqk;
                A[q * 9 + k] = s * apk + c * aqk;
            }
            mqs_wave_lds_sync();
        }
    }
}

// The same null vector by inverse iteration, every lane for itself (uniform control flow, no LDS traffic beyond reading A): LDL^T of
// A (positive semi-definite; the last pivot kept away from zero), a few plain steps from L^-T e_8, then -- the homography fits a
// non-planar scene badly and the two smallest eigenvalues may lie within a factor of two -- Rayleigh-quotient shifts (A - mu (1 - 2^-10) I
// factored again, two steps per shift).  Settled = the direction changes by less than 1e-11 between two steps, which at the shifted
// phase's rate bounds the error near 1e-14.  Returns false (clustered smallest eigenvalues, a leading pivot that is not positive):
// the caller falls back to jacobi9.  2-5 us where the Jacobi sweeps take 26 (phase stamps of frame_decide_kernel, round 5).
struct Ldlt9 { double l[36]; double inv[9]; bool ok; };           // l: unit lower factor, row-major over the strict lower triangle

__device__ __forceinline__ constexpr int ltri(int i, int j) { return i * (i - 1) / 2 + j; }      // i > j

__device__ __forceinline__ void ldlt9_shifted(const double *A, double shift, double tiny, Ldlt9 &f)
{
    double t[36];                                                  // t(i, q) = l(i, q) d_q
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double dj = A[j * 9 + j] - shift;
#pragma unroll
        for (int q = 0; q < j; ++q) dj = fma(-f.l[ltri(j, q)], t[ltri(j, q)], dj);
        if (j < 8) ok = ok && (dj > 0.0);
        else if (!(fabs(dj) > tiny)) dj = tiny;                    // exact data: the smallest eigenvalue is zero to rounding
        const double ij = mqs::rcp(dj);
        f.inv[j] = ij;
#pragma unroll
        for (int i = j + 1; i < 9; ++i) {
            double a = A[i * 9 + j];
#pragma unroll
            for (int q = 0; q < j; ++q) a = fma(-f.l[ltri(i, q)], t[ltri(j, q)], a);
            t[ltri(i, j)] = a;
            f.l[ltri(i, j)] = a * ij;
        }
    }
    f.ok = ok;
}

// one step v <- (L D L^T)^-1 v on the power-of-two normalised iterate; true when the direction has settled
__device__ __forceinline__ bool invit9_step(const Ldlt9 &f, double (&v)[9])
{
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) m = fmax(m, fabs(v[i]));
    const double sc = ldexp(1.0, -ilogb(m));
    double p[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { v[i] *= sc; p[i] = v[i]; }
#pragma unroll
    for (int i = 1; i < 9; ++i)
#pragma unroll
        for (int q = 0; q < i; ++q) v[i] = fma(-f.l[ltri(i, q)], v[q], v[i]);
#pragma unroll
    for (int i = 8; i >= 0; --i) {
        double a = v[i] * f.inv[i];
#pragma unroll
        for (int q = i + 1; q < 9; ++q) a = fma(-f.l[ltri(q, i)], v[q], a);
        v[i] = a;
    }
    double pp = 0.0, vp = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { pp = fma(p[i], p[i], pp); vp = fma(v[i], p[i], vp); }
    const double a = vp * mqs::rcp(pp);
    double err = 0.0, mag = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { err = fmax(err, fabs(fma(-a, p[i], v[i]))); mag = fmax(mag, fabs(v[i])); }
    return err <= 1e-11 * mag;
}

__device__ bool null_vector9_invit(const double *A /*LDS [9][9]*/, double (&v)[9])
{
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) tr += A[i * 9 + i];
    const double tiny = 1e-30 * tr;
    Ldlt9 f;
    ldlt9_shifted(A, 0.0, tiny, f);
    if (!f.ok) return false;
    // first iterate: the direction of A^-1 e_8 = L^-T e_8 / d_8
#pragma unroll
    for (int i = 8; i >= 0; --i) {
        double a = (i == 8) ? 1.0 : 0.0;
#pragma unroll
        for (int q = i + 1; q < 9; ++q) a = fma(-f.l[ltri(q, i)], v[q], a);
        v[i] = a;
    }
    bool done = false;
    for (int it = 0; it < 3 && !done; ++it) done = invit9_step(f, v);
    for (int round = 0; round < 4 && !done; ++round) {
        double vv = 0.0, vav = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double w = 0.0;
#pragma unroll
            for (int j = 0; j < 9; ++j) w = fma(A[i * 9 + j], v[j], w);
            vv = fma(v[i], v[i], vv); vav = fma(v[i], w, vav);
        }
        const double mu = vav * mqs::rcp(vv);
        ldlt9_shifted(A, mu * (1.0 - 0x1p-10), tiny, f);
        if (!f.ok) return false;
        for (int it = 0; it < 2 && !done; ++it) done = invit9_step(f, v);
    }
    return done;
}

// eigenvalues of a symmetric 3 x 3 (a: 9 doubles, destroyed), single thread
__device__ void jacobi3(double *a, double *w)
{
    for (int sweep = 0; sweep < 10; ++sweep) {
        const double off3 = a[1] * a[1] + a[2] * a[2] + a[5] * a[5], dia3 = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off3 <= 1e-30 * dia3) break;
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double c, s;
            jacobi_angle(a[p * 3 + p], a[q * 3 + q], a[p * 3 + q], c, s);
            for (int k = 0; k < 3; ++k) {
                const double akp = a[k * 3 + p], akq = a[k * 3 + q];
                a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
                const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk;
            }
        }
    }
    w[0] = a[0]; w[1] = a[4]; w[2] = a[8];
}

// sum over the workgroup, the same value in every thread, fixed order (wave butterflies, then the four waves in order)
// The second half of cv2.findHomography(method = 0) (OpenCV 2.4 fundam.cpp: `estimator.refine(M, m, &matH, 10)` whenever there are
// more than four pairs): Levenberg-Marquardt on the eight free entries of H (h33 = 1) over the transfer error
// sum |u2 - proj(H u1)|^2 with CvLevMarq's schedule -- lambda 1e-3, J^T J with its diagonal scaled by (1 + lambda), a step that
// raises the error retried with lambda x 10, an accepted one divides lambda by 10, at most ten accepted steps (OpenCV stops
// early only below DBL_EPSILON; here below 1e-12 relative: slam_loop.homography_refine, the host twin).  ONE wavefront: the pairs
// dealt over the lanes, the 29 distinct sums of J^T J, J^T r in one transposed wave reduction, the 8 x 8 system solved by every
// lane (uniform control flow, no barrier).  sH [9]: H on entry (h33 = 1) and on return; sT [32]: scratch.
__device__ void homography_refine_wave(const double *__restrict__ u1, const double *__restrict__ u2, int n, double *sH, double *sT, int lane)
{
    double h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = sH[i];
    auto transfer_error = [&](const double (&g)[8]) {
        double e = 0.0;
        for (int k = lane; k < n; k += 64) {
            const double ax = u1[2 * k], ay = u1[2 * k + 1];
            const double w = mqs::rcp(fma(g[6], ax, fma(g[7], ay, 1.0)));
            const double rx = fma(g[0], ax, fma(g[1], ay, g[2])) * w - u2[2 * k], ry = fma(g[3], ax, fma(g[4], ay, g[5])) * w - u2[2 * k + 1];
            e = fma(rx, rx, fma(ry, ry, e));
        }
        double s, unused;
        mqs::wave::sum2(e, 0.0, s, unused);
        return s;
    };
    double err = transfer_error(h);
    int lam = -3;
    for (int it = 0; it < 10; ++it) {
        double v[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = 0.0;
        for (int k = lane; k < n; k += 64) {
            const double ax = u1[2 * k], ay = u1[2 * k + 1];
            const double w = mqs::rcp(fma(h[6], ax, fma(h[7], ay, 1.0)));
            const double x = fma(h[0], ax, fma(h[1], ay, h[2])) * w, y = fma(h[3], ax, fma(h[4], ay, h[5])) * w;
            const double rx = x - u2[2 * k], ry = y - u2[2 * k + 1];
            // rows of J: [a0 a1 a2 0 0 0 c0 c1] (x) and [0 0 0 a0 a1 a2 d0 d1] (y)
            const double a0 = ax * w, a1 = ay * w, a2 = w, c0 = -a0 * x, c1 = -a1 * x, d0 = -a0 * y, d1 = -a1 * y;
            v[0] = fma(a0, a0, v[0]); v[1] = fma(a0, a1, v[1]); v[2] = fma(a0, a2, v[2]);
            v[3] = fma(a1, a1, v[3]); v[4] = fma(a1, a2, v[4]); v[5] = fma(a2, a2, v[5]);
            v[6] = fma(a0, c0, v[6]); v[7] = fma(a0, c1, v[7]); v[8] = fma(a1, c0, v[8]);
            v[9] = fma(a1, c1, v[9]); v[10] = fma(a2, c0, v[10]); v[11] = fma(a2, c1, v[11]);
            v[12] = fma(a0, d0, v[12]); v[13] = fma(a0, d1, v[13]); v[14] = fma(a1, d0, v[14]);
            v[15] = fma(a1, d1, v[15]); v[16] = fma(a2, d0, v[16]); v[17] = fma(a2, d1, v[17]);
            v[18] = fma(c0, c0, fma(d0, d0, v[18])); v[19] = fma(c0, c1, fma(d0, d1, v[19])); v[20] = fma(c1, c1, fma(d1, d1, v[20]));
            v[21] = fma(a0, rx, v[21]); v[22] = fma(a1, rx, v[22]); v[23] = fma(a2, rx, v[23]);
            v[24] = fma(a0, ry, v[24]); v[25] = fma(a1, ry, v[25]); v[26] = fma(a2, ry, v[26]);
            v[27] = fma(c0, rx, fma(d0, ry, v[27])); v[28] = fma(c1, rx, fma(d1, ry, v[28]));
        }
        const double t = mqs::wave::wave_reduce32(v, lane);
        mqs_wave_lds_sync();                                  // (the previous round's readers of sT are done)
        if (!(lane & 1)) sT[lane >> 1] = t;
        mqs_wave_lds_sync();
        double A[8][8], g[8];
        {
            const double S[3][3] = {{sT[0], sT[1], sT[2]}, {sT[1], sT[3], sT[4]}, {sT[2], sT[4], sT[5]}};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { A[i][j] = S[i][j]; A[3 + i][3 + j] = S[i][j]; A[i][3 + j] = 0.0; A[3 + i][j] = 0.0; }
                A[i][6] = A[6][i] = sT[6 + 2 * i]; A[i][7] = A[7][i] = sT[7 + 2 * i];
                A[3 + i][6] = A[6][3 + i] = sT[12 + 2 * i]; A[3 + i][7] = A[7][3 + i] = sT[13 + 2 * i];
            }
            A[6][6] = sT[18]; A[6][7] = A[7][6] = sT[19]; A[7][7] = sT[20];
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = sT[21 + i];
        }
        bool accepted = false;
        double hn[8], err2 = 0.0, dn = 0.0;
        while (lam <= 16) {
            // 10^|lam| by its binary digits (|lam| <= 16; every factor and product exact, as the loop of |lam| multiplications was)
            const int al = lam < 0 ? -lam : lam;
            double scale = (al & 1) ? 10.0 : 1.0;
            if (al & 2) scale *= 100.0;
            if (al & 4) scale *= 1e4;
            if (al & 8) scale *= 1e8;
            if (al & 16) scale *= 1e16;
            const double damp = 1.0 + (lam < 0 ? mqs::rcp(scale) : scale);
            // Cholesky of A with the diagonal scaled; the step solves (A + lambda diag A) step = g.  Reciprocal square roots of the
            // pivots, no division anywhere: the fifty fp64 divisions of the textbook form were most of the refinement's 30 us
            double Lm[8][8], ri[8], yv[8], st[8];
            bool pd = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double piv = A[j][j] * damp;
#pragma unroll
                for (int q = 0; q < j; ++q) piv = fma(-Lm[j][q], Lm[j][q], piv);
                pd = pd && (piv > 0.0);
                ri[j] = rsqrt(piv > 0.0 ? piv : 1.0);
#pragma unroll
                for (int i = j + 1; i < 8; ++i) {
                    double sacc = A[i][j];
#pragma unroll
                    for (int q = 0; q < j; ++q) sacc = fma(-Lm[i][q], Lm[j][q], sacc);
                    Lm[i][j] = sacc * ri[j];
                }
            }
            if (!pd) { ++lam; continue; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double sacc = g[i];
#pragma unroll
                for (int q = 0; q < i; ++q) sacc = fma(-Lm[i][q], yv[q], sacc);
                yv[i] = sacc * ri[i];
            }
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                double sacc = yv[i];
#pragma unroll
                for (int q = i + 1; q < 8; ++q) sacc = fma(-Lm[q][i], st[q], sacc);
                st[i] = sacc * ri[i];
            }
            dn = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { hn[i] = h[i] - st[i]; dn = fma(st[i], st[i], dn); }
            err2 = transfer_error(hn);
            if (!(err2 <= err)) { ++lam; continue; }
            accepted = true;
            break;
        }
        if (!accepted) break;
        double hh = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { hh = fma(h[i], h[i], hh); h[i] = hn[i]; }
        err = err2;
        lam = lam - 1 < -16 ? -16 : lam - 1;
        if (dn < 1e-24 * hh) break;                           // relative step below 1e-12
    }
    mqs_wave_lds_sync();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sH[i] = h[i];
        sH[8] = 1.0;
    }
    mqs_wave_lds_sync();
}

__device__ __forceinline__ double block_sum(double v, int tid, double *sRed /*[4]*/)
{
    v = mqs::wave::sum1(v);
    if ((tid & 63) == 0) sRed[tid >> 6] = v;
    __syncthreads();
    const double t = ((sRed[0] + sRed[1]) + sRed[2]) + sRed[3];
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// after the pose: gates (slam2.py:461-468, 493-497), commit of the kept tracks (:499-522), keyframe test (:43-59)
// ---------------------------------------------------------------------------------------------------------------------
// A/B builds only (-DMQS_DECIDE_STAMPS): the kernel's phases in 100 MHz ticks over the keyframe-pose slots of the result block
// (tools/probes/decide_phases.py reads them behind frames that are no keyframes)
#ifdef MQS_DECIDE_STAMPS
#define MQS_DSTAMP(i) do { if (tid == 0) d.res[R_KF_POSE + (i)] = (double)(long long)(wall_clock64() - dst0); } while (0)
#else
#define MQS_DSTAMP(i) do { } while (0)
#endif
// The keyframe test's number (slam2.py:43-59: the ratio of the largest to the smallest singular value of the homography between the base
// keyframe's and this frame's positions of the accepted tracks) by ONE workgroup that needs nothing of the refined pose: it picks the best
// hypothesis and marks its inliers as select_refine_block does (the same comparisons on the same numbers), lists the accepted tracks --
// kept by the filter, and no outlier of the pose -- in the commit's order, undistorts both ends, and runs findHomography on them.  In the
// two-workgroup form of frame_decide_kernel this runs BESIDE the pose refinement (30 us) instead of behind it (22 us of the kernel's 63).
// frame: the frame counter this frame's sample is seeded with.  Returns the ratio (1: fewer than four tracks / no model); n_acc_out: tracks.
__device__ __forceinline__ double decide_keyframe_ratio(const SlamDev &d, const SlamParams &p, int frame_no, int &n_acc_out)
{
#ifdef MQS_DECIDE_STAMPS
    const unsigned long long dst0 = wall_clock64();
#endif
    __shared__ int sWave[4];
    __shared__ double sRed[4];
    __shared__ uint8_t sInl[kMaxTracks];
    __shared__ double sU1[kMaxTracks * 2], sU2[kMaxTracks * 2];
    __shared__ double sAcc[4][48];
    __shared__ double sA[81], sV[81];
    __shared__ double sI[9], sP[12], sH[9];
    __shared__ int sBestC[256], sBestH[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tri = d.cnt[C_NTRI], n_keep = d.cnt[C_NKEEP];
    n_acc_out = 0;
    if (n_tri <= 0) return 1.0;                                    // rejected by the filter
    if (tid < 9) sI[tid] = d.intr[tid];
    // the hypothesis with the most inliers, the lowest index on ties (pnp_block.h: select_refine_block)
    int bc = -1, bh = -1;
    for (int h = tid; h < kHyp; h += 256)
        if (d.pnp_counts[h] > bc) { bc = d.pnp_counts[h]; bh = h; }
    sBestC[tid] = bc; sBestH[tid] = bh;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
            const int c2 = sBestC[tid + st], h2 = sBestH[tid + st];
            if (c2 > sBestC[tid] || (c2 == sBestC[tid] && h2 >= 0 && (sBestH[tid] < 0 || h2 < sBestH[tid]))) { sBestC[tid] = c2; sBestH[tid] = h2; }
        }
        __syncthreads();
    }
    const int best = sBestH[0];
    if (best < 0) return 1.0;
    if (tid < 12) sP[tid] = d.pnp_poses[12 * best + tid];
    for (int k = tid; k < n_keep; k += 256) sInl[k] = 1;
    __syncthreads();
    const double thr2 = p.max_reproj * p.max_reproj;
    for (int j = tid; j < n_tri; j += 256) {
        const double X = d.objp_t[3 * j], Y = d.objp_t[3 * j + 1], Z = d.objp_t[3 * j + 2], u = d.imgp_t[2 * j], v = d.imgp_t[2 * j + 1];
        const double Zc = fma(sP[8], X, fma(sP[9], Y, fma(sP[10], Z, sP[11])));
        const double e2 = mqs::pnp::reproj_sqerr(sP, sI, X, Y, Z, u, v);
        sInl[d.tri_pos[j]] = (Zc > 0.0 && e2 <= thr2) ? 1 : 0;
    }
    __syncthreads();
    int n_acc = 0;
    for (int b = 0; b < n_keep; b += 256) {
        const int k = b + tid;
        const bool in = k < n_keep && sInl[k] != 0;
        int tot;
        const int r = n_acc + block_rank(in, tid, sWave, tot);
        if (in) {                                                  // keyframe test inputs: both ends of the track, undistorted
            mqs::cam::undistort_pixel(sI, (double)d.t_base[2 * k], (double)d.t_base[2 * k + 1], sU1[2 * r], sU1[2 * r + 1]);
            mqs::cam::undistort_pixel(sI, (double)d.t_pts[2 * k], (double)d.t_pts[2 * k + 1], sU2[2 * r], sU2[2 * r + 1]);
        }
        n_acc += tot;
    }
    __syncthreads();
    n_acc_out = n_acc;
    MQS_DSTAMP(1);
    // keyframe_test's random sample of the kept tracks (slam2.py:48: np.random.permutation(n)[:max_num_homography_points]): a
    // counter-based hash of (seed, frame, track position) per track, the tracks with the max_homography_points smallest hashes are
    // the sample (a uniformly random subset; ties by position).  The sample is compacted to the front of sU1 / sU2 in track order.
    int n_hom = n_acc;
    if (p.max_homography_points > 0 && n_acc > p.max_homography_points) {
        __shared__ unsigned sHash[kMaxTracks];
        __shared__ uint8_t sPick[kMaxTracks];
        const unsigned long long frame = (unsigned long long)frame_no;
        for (int k = tid; k < n_acc; k += 256) {
            unsigned long long st = p.seed ^ (frame << 24) ^ ((unsigned long long)(k + 1) * 0x9e3779b97f4a7c15ull) ^ 0x5851f42d4c957f2dull;
            sHash[k] = (unsigned)(splitmix64(st) >> 32);
        }
        __syncthreads();
        for (int k = tid; k < n_acc; k += 256) {
            const unsigned hk = sHash[k];
            int rank = 0;
            for (int j = 0; j < n_acc; ++j) rank += (sHash[j] < hk || (sHash[j] == hk && j < k)) ? 1 : 0;
            sPick[k] = rank < p.max_homography_points ? 1 : 0;
        }
        __syncthreads();
        int n_sel = 0;
        for (int b = 0; b < n_acc; b += 256) {                       // ordered compaction, in place (the target never runs ahead of the source)
            const int k = b + tid;
            const bool in = k < n_acc && sPick[k] != 0;
            double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (in) { a0 = sU1[2 * k]; a1 = sU1[2 * k + 1]; b0 = sU2[2 * k]; b1 = sU2[2 * k + 1]; }
            int tot;
            const int r = n_sel + block_rank(in, tid, sWave, tot);
            if (in) { sU1[2 * r] = a0; sU1[2 * r + 1] = a1; sU2[2 * r] = b0; sU2[2 * r + 1] = b1; }
            n_sel += tot;
            __syncthreads();
        }
        n_hom = n_sel;
    }
    MQS_DSTAMP(2);
    // normalised DLT homography u2 ~ H u1 over the sample
    double ratio = 1.0;
    if (n_acc >= 4) {
        const int n_acc_all = n_acc;
        const int n_acc = n_hom;                                      // (shadows: the loops below run over the sample)
        (void)n_acc_all;
        double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
        for (int k = tid; k < n_acc; k += 256) { c1x += sU1[2 * k]; c1y += sU1[2 * k + 1]; c2x += sU2[2 * k]; c2y += sU2[2 * k + 1]; }
        const double inv_n = 1.0 / (double)n_acc;
        c1x = block_sum(c1x, tid, sRed) * inv_n; c1y = block_sum(c1y, tid, sRed) * inv_n;
        c2x = block_sum(c2x, tid, sRed) * inv_n; c2y = block_sum(c2y, tid, sRed) * inv_n;
        double d1 = 0, d2 = 0;
        for (int k = tid; k < n_acc; k += 256) {
            d1 += sqrt((sU1[2 * k] - c1x) * (sU1[2 * k] - c1x) + (sU1[2 * k + 1] - c1y) * (sU1[2 * k + 1] - c1y));
            d2 += sqrt((sU2[2 * k] - c2x) * (sU2[2 * k] - c2x) + (sU2[2 * k + 1] - c2y) * (sU2[2 * k + 1] - c2y));
        }
        const double s1 = 1.4142135623730951 / fmax(block_sum(d1, tid, sRed) * inv_n, 1e-12);
        const double s2 = 1.4142135623730951 / fmax(block_sum(d2, tid, sRed) * inv_n, 1e-12);
        // the 45 distinct sums of A^T A (rows [a 1 0 0 0 -bx a -bx], [0 0 0 a 1 -by a -by]), per thread, then over the workgroup
        double acc[45];
#pragma unroll
        for (int e = 0; e < 45; ++e) acc[e] = 0.0;
        for (int k = tid; k < n_acc; k += 256) {
            const double ax = (sU1[2 * k] - c1x) * s1, ay = (sU1[2 * k + 1] - c1y) * s1;
            const double bx = (sU2[2 * k] - c2x) * s2, by = (sU2[2 * k + 1] - c2y) * s2;
            const double r1[9] = {ax, ay, 1.0, 0.0, 0.0, 0.0, -bx * ax, -bx * ay, -bx};
            const double r2[9] = {0.0, 0.0, 0.0, ax, ay, 1.0, -by * ax, -by * ay, -by};
            int e = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = i; j < 9; ++j, ++e) acc[e] = fma(r1[i], r1[j], fma(r2[i], r2[j], acc[e]));
        }
        {
            double v[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] = acc[e];
            const double t0 = mqs::wave::wave_reduce32(v, lane);
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] = e < 13 ? acc[32 + e] : 0.0;
            const double t1 = mqs::wave::wave_reduce32(v, lane);
            if (!(lane & 1)) { sAcc[wave][lane >> 1] = t0; if ((lane >> 1) < 13) sAcc[wave][32 + (lane >> 1)] = t1; }
        }
        __syncthreads();
        if (tid < 45) {
            const double t = ((sAcc[0][tid] + sAcc[1][tid]) + sAcc[2][tid]) + sAcc[3][tid];
            // entry tid = (i, j), i <= j, row-major over the upper triangle
            int i = 0, rem = tid;
            while (rem >= 9 - i) { rem -= 9 - i; ++i; }
            const int j = i + rem;
            sA[i * 9 + j] = t; sA[j * 9 + i] = t;
        }
        __syncthreads();
        MQS_DSTAMP(3);
        if (wave == 0) {
            double hn[9];
            if (p.null_vector_jacobi || !null_vector9_invit(sA, hn)) {                          // (wave-uniform: every lane computed the same)
                jacobi9(sA, sV, lane);
                int im = 0;
                for (int i = 1; i < 9; ++i) if (sA[i * 9 + i] < sA[im * 9 + im]) im = i;
                for (int i = 0; i < 9; ++i) hn[i] = sV[i * 9 + im];
            }
            MQS_DSTAMP(4);
            if (lane == 0) {
                double h[9], t[9];
                // H = inv(Tb) Hn Ta,  Ta = [s1 0 -s1 c1x; 0 s1 -s1 c1y; 0 0 1],  inv(Tb) = [1/s2 0 c2x; 0 1/s2 c2y; 0 0 1]
                for (int r = 0; r < 3; ++r) {
                    t[3 * r + 0] = hn[3 * r + 0] * s1;
                    t[3 * r + 1] = hn[3 * r + 1] * s1;
                    t[3 * r + 2] = hn[3 * r + 2] - s1 * (hn[3 * r + 0] * c1x + hn[3 * r + 1] * c1y);
                }
                const double is2 = mqs::rcp(s2);
                for (int c = 0; c < 3; ++c) {
                    h[c] = t[c] * is2 + c2x * t[6 + c];
                    h[3 + c] = t[3 + c] * is2 + c2y * t[6 + c];
                    h[6 + c] = t[6 + c];
                }
                const double i8 = mqs::rcp(h[8]);                       // cvConvertScale(&_H0, H, 1. / _H0.data.db[8])
                for (int i = 0; i < 9; ++i) sH[i] = h[i] * i8;
            }
            mqs_wave_lds_sync();
            // ... then the refinement of the transfer error (findHomography's second half), the whole wave
            MQS_DSTAMP(5);
            if (n_acc > 4 && p.homography_refine) homography_refine_wave(sU1, sU2, n_acc, sH, sA, lane);
            MQS_DSTAMP(6);
            if (lane == 0) {
                double h[9];
                for (int i = 0; i < 9; ++i) h[i] = sH[i];
                double gm[9], w[3];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) gm[3 * i + j] = h[i] * h[j] + h[3 + i] * h[3 + j] + h[6 + i] * h[6 + j];
                jacobi3(gm, w);
                const double wmax = fmax(w[0], fmax(w[1], w[2])), wmin = fmin(w[0], fmin(w[1], w[2]));
                sRed[0] = wmin > 0.0 ? sqrt(wmax / wmin) : 1e300;
            }
        }
        __syncthreads();
        ratio = sRed[0];
        MQS_DSTAMP(7);
    }
    return ratio;
}

// The pose's share of the decision: gates (slam2.py:461-468, 493-497) on the refined pose.  Returns the reason (0: accepted so far).
__device__ __forceinline__ int decide_gates(const SlamDev &d, const SlamParams &p, double *sP, double *sI, double *sRed)
{
    const int tid = threadIdx.x;
    if (tid == 0) { d.res[R_NTRACKS] = (double)d.cnt[C_N]; d.res[R_NLAND] = (double)d.cnt[C_NLAND]; }
    if (d.res[R_DECISION] == 0.0) return -1;                       // rejected by the filter: the state is untouched
    const int n_tri = d.cnt[C_NTRI];
    const int best = d.sel[0], ninl = d.sel[1];
    if (tid < 9) sI[tid] = d.intr[tid];
    if (tid < 12) sP[tid] = d.pose_r[tid];
    __syncthreads();
    const double outlier = n_tri > 0 ? (double)(n_tri - ninl) / (double)n_tri : 1.0;
    int reason = 0;
    if (best < 0) reason = 3;                                      // no valid RANSAC model
    else if (outlier > p.max_outlier_ratio || ninl < 8) reason = 4;
    // reprojection RMS of the inliers under the refined pose (calibration_tools.py:116-124)
    double e2 = 0.0;
    for (int j = tid; j < n_tri; j += 256)
        if (d.inl_mask[j])
            e2 += mqs::pnp::reproj_sqerr(sP, sI, d.objp_t[3 * j], d.objp_t[3 * j + 1], d.objp_t[3 * j + 2], d.imgp_t[2 * j], d.imgp_t[2 * j + 1]);
    const double sq = block_sum(e2, tid, sRed);
    const double rms = ninl > 0 ? sqrt(sq / (double)ninl) : 0.0;
    if (!reason && !(rms <= p.max_reproj)) reason = 5;
    if (tid == 0) {
        d.res[R_NINL] = (double)ninl; d.res[R_OUTLIER] = outlier; d.res[R_REPROJ] = rms;
        if (reason) { d.res[R_DECISION] = 0.0; d.res[R_REASON] = (double)reason; }
    }
    return reason;
}

// commit (slam2.py:499-522): inlier landmark tracks and the free tracks stay, in order; the keyframe step's two point sets beside them;
// then the decision from the keyframe test's ratio.
__device__ __forceinline__ void decide_commit(const SlamDev &d, const SlamParams &p, const double *sP, double ratio, int n_acc_kf)
{
    __shared__ int sWave[4];
    __shared__ uint8_t sInl[kMaxTracks];
    const int tid = threadIdx.x;
    const int n_tri = d.cnt[C_NTRI], n_keep = d.cnt[C_NKEEP];
    for (int k = tid; k < n_keep; k += 256) sInl[k] = 1;
    __syncthreads();
    for (int j = tid; j < n_tri; j += 256) sInl[d.tri_pos[j]] = d.inl_mask[j];
    __syncthreads();
    int n_acc = 0, n_old = 0, n_new = 0;
    const int nlog0 = d.log_lm ? d.cnt[C_NLOG] : 0;
    for (int b = 0; b < n_keep; b += 256) {
        const int k = b + tid;
        const bool in = k < n_keep && sInl[k] != 0;
        const bool tri = in && d.t_lm[k] >= 0, fre = in && d.t_lm[k] < 0;
        int tot, tot_o, tot_n;
        const int r = n_acc + block_rank(in, tid, sWave, tot);
        const int ro = n_old + block_rank(tri, tid, sWave, tot_o);
        const int rn = n_new + block_rank(fre, tid, sWave, tot_n);
        if (in) {
            const float px = d.t_pts[2 * k], py = d.t_pts[2 * k + 1], bx = d.t_base[2 * k], by = d.t_base[2 * k + 1];
            d.pts[2 * r] = px; d.pts[2 * r + 1] = py;
            d.base[2 * r] = bx; d.base[2 * r + 1] = by;
            d.lm[r] = d.t_lm[k]; d.tid[r] = d.t_tid[k];
            d.spec_map[r] = k;                                    // where live track r sits among this frame's kept tracks (the tracker ahead ran on those)
            if (d.log_lm && nlog0 + r < d.log_cap) {              // slam2.py:519-522 (landmark tracks) and :634-641 (free tracks, resolved later)
                d.log_lm[nlog0 + r] = tri ? d.t_lm[k] : -2 - d.t_tid[k]; d.log_pose[nlog0 + r] = p.pose_index;
                d.log_uv[2 * (nlog0 + r)] = (double)px; d.log_uv[2 * (nlog0 + r) + 1] = (double)py;
            }
            if (tri) {
                const int l = d.t_lm[k];
                d.kf_objp[3 * ro] = d.map[3 * l]; d.kf_objp[3 * ro + 1] = d.map[3 * l + 1]; d.kf_objp[3 * ro + 2] = d.map[3 * l + 2];
                d.kf_imgp[2 * ro] = (double)px; d.kf_imgp[2 * ro + 1] = (double)py;
            } else {
                d.kf_p0[2 * rn] = (double)bx; d.kf_p0[2 * rn + 1] = (double)by;
                d.kf_p1[2 * rn] = (double)px; d.kf_p1[2 * rn + 1] = (double)py;
                d.kf_pos[rn] = r;
            }
        }
        n_acc += tot; n_old += tot_o; n_new += tot_n;
    }
    __syncthreads();
    if (tid < 12) d.pose_prev[tid] = sP[tid];
    if (tid == 0) d.cnt[C_N] = n_acc;
    if (tid == 0 && d.log_lm) { d.cnt[C_NLOG] = min(nlog0 + n_acc, d.log_cap); if (nlog0 + n_acc > d.log_cap) d.cnt[C_LOG_OVERFLOW] = 1; }
    if (d.traj && tid < 12 && p.pose_index < d.traj_cap) d.traj[12 * (size_t)p.pose_index + tid] = sP[tid];
    if (tid == 0) {
        const bool key = n_acc >= 4 && n_acc == n_acc_kf && ratio > p.homography_threshold;
        d.res[R_DECISION] = key ? 2.0 : 1.0;
        d.res[R_REASON] = n_acc == n_acc_kf ? 0.0 : 6.0;            // (6 cannot happen: both workgroups list the same tracks)
        d.res[R_NTRACKS] = (double)n_acc; d.res[R_NOLD] = (double)n_old; d.res[R_NNEW] = (double)n_new;
        d.res[R_HOMOGRAPHY] = ratio;
        d.cnt[C_KF_PENDING] = key ? 1 : 0;
    }
    if (tid < 12) d.res[R_POSE + tid] = sP[tid];
}

// The decision, then the frame's result block straight into the caller's pinned host memory: the previous keyframe's report rides along
// and its flag is cleared for the next one -- what a device-to-host copy and a fill launch per frame did before (5 + 4 us of the frame's
// ~230).  TWO workgroups (round 6): workgroup 0 finishes solvePnPRansac (best hypothesis, refinement on its inliers: 30 us), applies the
// gates and commits; workgroup 1 computes the keyframe test's ratio meanwhile (22 us) and hands it over through a flag in device memory
// (agent-scope release / acquire; both workgroups are resident -- a launch of two).  One workgroup (p.decide_split == 0, A/B): the same
// functions one behind the other.
__global__ __launch_bounds__(256) void frame_decide_kernel(SlamDev d, SlamParams p)
{
    const int tid = threadIdx.x;
    // the slot of the pinned ring this launch reports into; its ticket goes behind the block with a system-scope release, the host polls for it
    // (no stream synchronisation per frame: a frame enqueued ahead may already be running behind this launch)
    double *out = d.res_out + (size_t)(p.ticket & (kResSlots - 1)) * kResStride;
    if (p.gated && d.cnt[C_LAST_DECISION] != 1) {                 // (uniform; see frame_hypothesis_kernel) -- nobody waits for this ticket
        if (tid == 0 && blockIdx.x == 0) {
            out[R_DECISION] = -2.0;
            __hip_atomic_store((unsigned long long *)(out + kRes), (unsigned long long)p.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int frame_no = d.cnt[C_FRAME] + 1;                       // (workgroup 0 writes the counter behind the hand-over)
    unsigned long long *kf_word = reinterpret_cast<unsigned long long *>(d.kf_hand);
    if (blockIdx.x == 1) {
        int n_acc;
        const double ratio = decide_keyframe_ratio(d, p, frame_no, n_acc);
        if (tid == 0) {
            __hip_atomic_store(d.kf_hand + 1, ratio, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.kf_hand + 2, (double)n_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(kf_word, (unsigned long long)p.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // the end of solvePnPRansac first (pnp_block.h: best hypothesis, its inliers into LDS, refinement by four wavefronts) -- a launch
    // of its own before, with a launch gap on either side
    extern __shared__ __attribute__((aligned(16))) double decide_lds[];
    __shared__ double sI[9], sP[12], sRed[4], sHand[2];
    mqs::pnpblk::select_refine_block(d.objp_t, d.imgp_t, kMaxTracks, d.cnt + C_NTRI, d.intr, d.pnp_poses, d.pnp_counts, kHyp, p.max_reproj * p.max_reproj,
                                     kPnpIters, kPnpEps, 1, d.pose_r, d.sel, d.pnp_inl, d.inl_mask, d.pnp_info, decide_lds);
    __threadfence_block();
    __syncthreads();
    const int reason = decide_gates(d, p, sP, sI, sRed);
    double ratio = 1.0;
    int n_acc_kf = 0;
    bool timed_out = false;
    if (gridDim.x == 1) {
        if (reason == 0) ratio = decide_keyframe_ratio(d, p, frame_no, n_acc_kf);
    } else {
        // the other workgroup's ratio (always waited for: the frame counter below is what it seeds its sample with)
        // (bounded like every device-side wait of this library: 0.5 s of the 100 MHz clock, looked at every 64th poll; the two workgroups are
        // one launch of two, so the other one is resident or about to be)
        if (tid == 0) {
            bool there = true;
            long long t0 = 0;
            for (unsigned spins = 0; __hip_atomic_load(kf_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)p.ticket; ++spins) {
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 63u) == 63u) {
                    if (t0 == 0) t0 = wall_clock64();
                    if (wall_clock64() - t0 > 50000000ll) { there = false; break; }
                }
            }
            sHand[0] = there ? __hip_atomic_load(d.kf_hand + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1.0;
            sHand[1] = there ? __hip_atomic_load(d.kf_hand + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.0;
        }
        __syncthreads();
        ratio = sHand[0]; n_acc_kf = (int)sHand[1];
        if (n_acc_kf < 0) {                                        // the wait gave up: nothing is committed, the call returns MQS_E_TIMEOUT
            if (tid == 0) { d.res[R_DECISION] = -3.0; d.res[R_REASON] = 7.0; }
            timed_out = true;
        }
    }
    if (reason == 0 && !timed_out) decide_commit(d, p, sP, ratio, n_acc_kf);
    __syncthreads();
    if (tid == 0) d.cnt[C_FRAME] = frame_no;
    if (tid < kRes) {
        out[tid] = d.res[tid];
        if (tid == R_KF_VALID) d.res[R_KF_VALID] = 0.0;
    }
    if (tid == 0) d.cnt[C_LAST_DECISION] = (int)d.res[R_DECISION];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store((unsigned long long *)(out + kRes), (unsigned long long)p.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------------
// keyframe: the new landmarks into the map, the tracks that did not triangulate dropped (slam2.py:589-612)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void keyframe_commit_kernel(SlamDev d, SlamParams p, int n_new)
{
    __shared__ int sWave[4];
    const int tid = threadIdx.x;
    const int n = d.cnt[C_N], nl = d.cnt[C_NLAND];
    const int nlog0 = d.log_lm ? d.cnt[C_NLOG] : 0;
    int n_good = 0;
    for (int b = 0; b < n_new; b += 256) {
        const int j = b + tid;
        const bool good = j < n_new && d.kf_status[j] >= 0;
        int tot;
        const int r = n_good + block_rank(good, tid, sWave, tot);
        if (good && nl + r < p.max_landmarks) {
            const int id = nl + r;
            d.map[3 * id] = (double)(float)d.kf_x[3 * j]; d.map[3 * id + 1] = (double)(float)d.kf_x[3 * j + 1];
            d.map[3 * id + 2] = (double)(float)d.kf_x[3 * j + 2];
            d.lm[d.kf_pos[j]] = id;
            // slam2.py:634-641: a new landmark brings its image points along: the frames since the base keyframe are in the log
            // under the track's id (resolved through tid2lm), the base keyframe's point -- the track was born behind that frame's
            // own log entries -- is added here
            const int e = nlog0 + r;
            if (d.log_lm && e < d.log_cap) {
                d.log_lm[e] = id; d.log_pose[e] = p.base_pose_index; d.log_uv[2 * e] = d.kf_p0[2 * j]; d.log_uv[2 * e + 1] = d.kf_p0[2 * j + 1];
            }
            const int t = d.tid[d.kf_pos[j]];
            if (d.tid2lm && t >= 0 && t < d.tid_cap) d.tid2lm[t] = id;
        }
        n_good += tot;
    }
    if (nl + n_good > p.max_landmarks) n_good = p.max_landmarks - nl;
    if (tid == 0 && d.log_lm) { d.cnt[C_NLOG] = min(nlog0 + n_good, d.log_cap); if (nlog0 + n_good > d.log_cap) d.cnt[C_LOG_OVERFLOW] = 1; }
    __threadfence_block();
    __syncthreads();
    // drop the free tracks that did not become landmarks: through the temporaries, back in order
    for (int k = tid; k < n; k += 256) {
        d.t_pts[2 * k] = d.pts[2 * k]; d.t_pts[2 * k + 1] = d.pts[2 * k + 1];
        d.t_lm[k] = d.lm[k]; d.t_tid[k] = d.tid[k];
    }
    __threadfence_block();
    __syncthreads();
    int n2 = 0;
    for (int b = 0; b < n; b += 256) {
        const int k = b + tid;
        const bool in = k < n && d.t_lm[k] >= 0;
        int tot;
        const int r = n2 + block_rank(in, tid, sWave, tot);
        if (in) { d.pts[2 * r] = d.t_pts[2 * k]; d.pts[2 * r + 1] = d.t_pts[2 * k + 1]; d.lm[r] = d.t_lm[k]; d.tid[r] = d.t_tid[k]; }
        n2 += tot;
    }
    // the refined pose of the keyframe step is the frame's pose and the new base keyframe's
    const double *pf = n_new > 0 ? d.kf_pose + 12 : d.pose_prev;
    double v = tid < 12 ? pf[tid] : 0.0;
    __syncthreads();
    if (tid < 12) {
        d.pose_prev[tid] = v; d.pose_key[tid] = v; d.res[R_KF_POSE + tid] = v;
        if (d.traj && p.pose_index < d.traj_cap) d.traj[12 * (size_t)p.pose_index + tid] = v;
    }
    if (tid == 0) {
        d.cnt[C_N] = n2; d.cnt[C_NLAND] = nl + n_good;
        d.res[R_KF_VALID] = 1.0; d.res[R_KF_NGOOD] = (double)n_good; d.res[R_KF_NLAND] = (double)(nl + n_good);
    }
}

// coverage mask (slam2.py:29-40): the mask was set to 1; a filled disc of zeros around every live track
__global__ __launch_bounds__(64) void coverage_disc_kernel(SlamDev d, SlamParams p)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= d.cnt[C_N]) return;
    // the centre as cv2.circle gets it from the reference's `tuple(p)` of float32 (cv2_helpers.py:26-27): the Python 2 binding parses
    // a Point with "ii", which takes a float through __int__ -- truncation, not rounding
    const int cx = (int)d.pts[2 * k], cy = (int)d.pts[2 * k + 1];
    const int r = (int)p.radius, side = 2 * r + 1;
    for (int e = lane; e < side * side; e += 64) {
        const int dy = e / side - r, dx = e % side - r;
        const int x = cx + dx, y = cy + dy;
        if (dx * dx + dy * dy <= r * r && x >= 0 && x < p.W && y >= 0 && y < p.H) d.mask[(size_t)y * p.W + x] = 0;
    }
}

// top-up (slam2.py:657-672) and rebase (:673-674): the strongest new corners up to the target, every track's base = its position
__global__ __launch_bounds__(256) void append_tracks_kernel(SlamDev d, SlamParams p)
{
    const int tid = threadIdx.x;
    const int n = d.cnt[C_N], found = d.gf_n[0], next = d.cnt[C_NEXT_TID];
    int add = p.target - n;
    if (add > found) add = found;
    if (add > kMaxTracks - n) add = kMaxTracks - n;
    if (add < 0) add = 0;
    for (int i = tid; i < add; i += 256) {
        d.pts[2 * (n + i)] = d.gf_xy[2 * i]; d.pts[2 * (n + i) + 1] = d.gf_xy[2 * i + 1];
        d.lm[n + i] = -1; d.tid[n + i] = next + i;
    }
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < n + add; k += 256) { d.base[2 * k] = d.pts[2 * k]; d.base[2 * k + 1] = d.pts[2 * k + 1]; }
    if (tid == 0) { d.cnt[C_N] = n + add; d.cnt[C_NEXT_TID] = next + add; d.cnt[C_KF_PENDING] = 0; d.res[R_KF_NTRACKS] = (double)(n + add); }
}

// first frame: the given correspondences become the first tracks and landmarks (slam2.py:1136-1180)
__global__ __launch_bounds__(256) void start_kernel(SlamDev d, const float *objp0, const float *imgp0, int n0)
{
    const int tid = threadIdx.x;
    for (int k = tid; k < n0; k += 256) {
        d.pts[2 * k] = imgp0[2 * k]; d.pts[2 * k + 1] = imgp0[2 * k + 1];
        d.lm[k] = k; d.tid[k] = k;
        d.map[3 * k] = (double)objp0[3 * k]; d.map[3 * k + 1] = (double)objp0[3 * k + 1]; d.map[3 * k + 2] = (double)objp0[3 * k + 2];
        d.kf_objp[3 * k] = (double)objp0[3 * k]; d.kf_objp[3 * k + 1] = (double)objp0[3 * k + 1]; d.kf_objp[3 * k + 2] = (double)objp0[3 * k + 2];
        d.kf_imgp[2 * k] = (double)imgp0[2 * k]; d.kf_imgp[2 * k + 1] = (double)imgp0[2 * k + 1];
        if (d.log_lm && k < d.log_cap) {                          // slam2.py:1167-1169: the first frame's associations
            d.log_lm[k] = k; d.log_pose[k] = 0; d.log_uv[2 * k] = (double)imgp0[2 * k]; d.log_uv[2 * k + 1] = (double)imgp0[2 * k + 1];
        }
    }
    if (tid == 0) {
        d.cnt[C_N] = n0; d.cnt[C_NLAND] = n0; d.cnt[C_NEXT_TID] = n0; d.cnt[C_FRAME] = 1; d.cnt[C_NTRI] = 0; d.cnt[C_NKEEP] = 0;
        d.cnt[C_KF_PENDING] = 0;
        d.cnt[C_NLOG] = d.log_lm ? min(n0, d.log_cap) : 0;
        d.cnt[C_LOG_OVERFLOW] = (d.log_lm && n0 > d.log_cap) ? 1 : 0;
    }
}

__global__ void start_pose_kernel(SlamDev d)
{
    const int tid = threadIdx.x;
    if (tid < 12) { const double v = d.pose_r[tid]; d.pose_prev[tid] = v; d.pose_key[tid] = v; d.res[R_POSE + tid] = v; if (d.traj) d.traj[tid] = v; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Re-association behind a keyframe's top-up: the reference's match_OF_based (slam.py:81-127) -- predicted positions matched to
// freshly detected keypoints by BFMatcher.radiusMatch on PIXEL coordinates (cv2_helpers.py:296-339), ratio test, one match per
// keypoint -- with the projections of the landmarks that are in the map but no longer tracked as the predictions and the
// top-up's new corners as the keypoints.  A matched corner takes its landmark up again instead of starting a new one.
// ---------------------------------------------------------------------------------------------------------------------
constexpr float kFar = 1.0e9f;                   // coordinates of a row that must match nothing

// queries: row l = the projection of landmark l under the frame's pose, or (far, far) when the landmark is tracked, behind the
// camera or outside the image; trains: row k = track k's position, or (-far, -far) beyond the live tracks
__global__ __launch_bounds__(256) void reassoc_rows_kernel(SlamDev d, SlamParams p, float *q_xy, int nq, float *t_xy, uint8_t *tracked)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = d.cnt[C_N], nl = d.cnt[C_NLAND];
    if (g < kMaxTracks) {
        t_xy[2 * g] = g < n ? d.pts[2 * g] : -kFar;
        t_xy[2 * g + 1] = g < n ? d.pts[2 * g + 1] : -kFar;
    }
    if (g < nq) {
        float u = kFar, v = kFar;
        if (g < nl && !tracked[g]) {
            double pu, pv;
            const double Z = mqs::cam::project(d.pose_prev, d.intr, d.map[3 * g], d.map[3 * g + 1], d.map[3 * g + 2], pu, pv);
            if (Z > 0.0 && pu >= 0.0 && pv >= 0.0 && pu <= (double)(p.W - 1) && pv <= (double)(p.H - 1)) { u = (float)pu; v = (float)pv; }
        }
        q_xy[2 * g] = u; q_xy[2 * g + 1] = v;
    }
}

__global__ __launch_bounds__(256) void reassoc_mark_kernel(SlamDev d, uint8_t *tracked)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < d.cnt[C_N] && d.lm[k] >= 0) tracked[d.lm[k]] = 1;
}

// a free track whose corner matched a lost landmark's projection becomes that landmark's track (and its observation is logged)
__global__ __launch_bounds__(256) void reassoc_apply_kernel(SlamDev d, SlamParams p, const int32_t *query_of_train, int32_t *n_out)
{
    __shared__ int sWave[4];
    const int tid = threadIdx.x;
    const int n = d.cnt[C_N];
    const int nlog0 = d.log_lm ? d.cnt[C_NLOG] : 0;
    int done = 0;
    for (int b = 0; b < n; b += 256) {
        const int k = b + tid;
        const bool hit = k < n && d.lm[k] < 0 && query_of_train[k] >= 0;
        int tot;
        const int r = done + block_rank(hit, tid, sWave, tot);
        if (hit) {
            const int l = query_of_train[k];
            d.lm[k] = l;
            const int e = nlog0 + r;
            if (d.log_lm && e < d.log_cap) {
                d.log_lm[e] = l; d.log_pose[e] = p.pose_index;
                d.log_uv[2 * e] = (double)d.pts[2 * k]; d.log_uv[2 * e + 1] = (double)d.pts[2 * k + 1];
            }
        }
        done += tot;
    }
    if (tid == 0) {
        n_out[0] = done;
        if (d.log_lm) { d.cnt[C_NLOG] = min(nlog0 + done, d.log_cap); if (nlog0 + done > d.log_cap) d.cnt[C_LOG_OVERFLOW] = 1; }
    }
}

}  // namespace


namespace {

int topup(mqs_slam *s, const uint8_t *img)
{
    { const int rcw = mqs_slam_ingest_main_wait(s, img); if (rcw != MQS_OK) return rcw; }      // (the corner detector reads the image on the loop's stream)
    MQS_HIP_CHECK(hipMemsetAsync(s->d.mask, 1, (size_t)s->p.W * s->p.H, s->stream));
    hipLaunchKernelGGL(coverage_disc_kernel, dim3(kMaxTracks), dim3(64), 0, s->stream, s->d, s->p);
    int rc = mqs_good_features_to_track_dev(img, s->p.W, s->p.H, s->p.target, s->p.quality, s->p.radius, s->d.mask, s->d.gf_xy,
                                            kMaxTracks, s->d.gf_n, s->ws_gftt, s->ws_gftt_bytes, s->stream);
    if (rc != MQS_OK) return rc;
    hipLaunchKernelGGL(append_tracks_kernel, dim3(1), dim3(256), 0, s->stream, s->d, s->p);
    MQS_HIP_CHECK(hipGetLastError());
    return MQS_OK;
}

}  // namespace

extern "C" {

int mqs_slam_create(int device, int W, int H, const double *intr, int target_keypoints, double coverage_radius,
                    double quality_level, int max_landmarks, uint64_t seed, mqs_slam **out)
{
    MQS_ARG_CHECK(out != nullptr && intr != nullptr, "out, intr must not be null");
    *out = nullptr;
    MQS_ARG_CHECK(W >= 16 && H >= 16 && W < 65536 && H < 65536, "16 <= W, H < 65536");
    MQS_ARG_CHECK(target_keypoints >= 8 && target_keypoints <= kMaxTracks, "8 <= target_keypoints <= 512");
    MQS_ARG_CHECK(coverage_radius >= 1.0 && coverage_radius <= 64.0 && quality_level > 0.0 && max_landmarks >= 64, "parameter ranges");
    MQS_HIP_CHECK(hipSetDevice(device));
    mqs_slam *s = new (std::nothrow) mqs_slam();
    if (!s) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
    s->device = device;
    s->started = false;
    s->accepted = 0; s->base_pose = 0; s->log_arena = nullptr; s->re_arena = nullptr; s->ba = nullptr; s->land_ub = 0; s->key_pose = 0; s->ingest = nullptr;
    s->ws_lk2 = nullptr; s->pyr_stream = nullptr; s->ahead.set = false; s->spec.valid = false; s->last_decision = 0;
    s->spec_enabled = true;
    if (const char *e = getenv("MQS_SLAM_TRACK_AHEAD")) s->spec_enabled = e[0] != '0';      // A/B: 0 = the tracker inside its frame's call
    s->pipeline = false; s->pre.valid = false; s->ticket = 0;
    s->decide_grid = 2;
    if (const char *e = getenv("MQS_SLAM_DECIDE_SPLIT")) s->decide_grid = e[0] != '0' ? 2 : 1;   // A/B: 0 = one workgroup, the keyframe test behind the pose
    for (int k = 0; k < 2; ++k) { s->prep[k].valid = false; s->prep[k].prev = nullptr; s->prep[k].next = nullptr; s->prep[k].has_event = false; }
    s->p = SlamParams{W, H, target_keypoints, max_landmarks, coverage_radius, quality_level,
                      12.0, 0.5, 2.0, 0.33, 1.04, 0.0, (unsigned long long)seed, 1, 0, 0, 0, 0, 0, 0, 0u};  // slam2.py:1070-1098; keyframe test on ALL tracks
    if (const char *e = getenv("MQS_SLAM_HOMOGRAPHY_REFINE")) s->p.homography_refine = e[0] != '0';        // A/B: 0 = the DLT alone
    if (const char *e = getenv("MQS_SLAM_NULL_VECTOR_JACOBI")) s->p.null_vector_jacobi = e[0] != '0';      // A/B: 1 = the Jacobi sweeps always
    s->fused_filter = true;
    if (const char *e = getenv("MQS_SLAM_FUSED_FILTER")) s->fused_filter = e[0] != '0';                    // A/B: 0 = filter and hypotheses as two launches
    s->ws_lk_bytes = mqs_lk_workspace_bytes(W, H, 3);
    s->ws_gftt_bytes = mqs_gftt_workspace_bytes(W, H);
    const int64_t ws_pnp_bytes = mqs_pnp_workspace_bytes(kMaxTracks, kHyp);
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = up(off + bytes); return o; };
    const size_t T = kMaxTracks;
    const size_t o_cnt = take(C_COUNT * 4), o_pts = take(T * 8), o_base = take(T * 8), o_lm = take(T * 4), o_tid = take(T * 4),
                 o_map = take((size_t)max_landmarks * 24), o_pk = take(96), o_pp = take(96), o_intr = take(72),
                 o_lkp = take(T * 8), o_lke = take(T * 4), o_lks = take(T), o_lkpb = take(T * 8), o_lkeb = take(T * 4), o_lksb = take(T), o_smap = take(T * 4), o_tp = take(T * 8), o_tb = take(T * 8), o_tl = take(T * 4),
                 o_tt = take(T * 4), o_ot = take(T * 24), o_it = take(T * 16), o_tpos = take(T * 4), o_smp = take((size_t)kHyp * kSample * 4),
                 o_pr = take(96), o_sel = take(8), o_im = take(T), o_pi = take(32), o_ko = take(T * 24), o_ki = take(T * 16),
                 o_k0 = take(T * 16), o_k1 = take(T * 16), o_kp = take(T * 4), o_ks = take(T * 80 + 64), o_kpose = take(192),
                 o_kx = take(T * 24), o_kinfo = take(64), o_kst = take(T * 4), o_mask = take((size_t)W * H), o_gxy = take(T * 8),
                 o_gn = take(4), o_res = take(kRes * 8), o_hand = take(32), o_wl = take((size_t)s->ws_lk_bytes), o_wg = take((size_t)s->ws_gftt_bytes),
                 o_wp = take((size_t)ws_pnp_bytes);
    hipError_t e = hipMalloc((void **)&s->arena, off);
    if (e != hipSuccess) { delete s; mqs_set_error("hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); return MQS_E_NOMEM; }
    char *a = s->arena;
    SlamDev &d = s->d;
    d.cnt = (int32_t *)(a + o_cnt); d.pts = (float *)(a + o_pts); d.base = (float *)(a + o_base); d.lm = (int32_t *)(a + o_lm);
    d.tid = (int32_t *)(a + o_tid); d.map = (double *)(a + o_map); d.pose_key = (double *)(a + o_pk); d.pose_prev = (double *)(a + o_pp);
    d.intr = (double *)(a + o_intr); d.lk_pts = (float *)(a + o_lkp); d.lk_err = (float *)(a + o_lke); d.lk_st = (uint8_t *)(a + o_lks);
    d.lk_pts_b = (float *)(a + o_lkpb); d.lk_err_b = (float *)(a + o_lkeb); d.lk_st_b = (uint8_t *)(a + o_lksb); d.spec_map = (int32_t *)(a + o_smap);
    d.t_pts = (float *)(a + o_tp); d.t_base = (float *)(a + o_tb); d.t_lm = (int32_t *)(a + o_tl); d.t_tid = (int32_t *)(a + o_tt);
    d.objp_t = (double *)(a + o_ot); d.imgp_t = (double *)(a + o_it); d.tri_pos = (int32_t *)(a + o_tpos); d.samples = (int32_t *)(a + o_smp);
    d.pose_r = (double *)(a + o_pr); d.sel = (int32_t *)(a + o_sel); d.inl_mask = (uint8_t *)(a + o_im); d.pnp_info = (double *)(a + o_pi);
    d.kf_objp = (double *)(a + o_ko); d.kf_imgp = (double *)(a + o_ki); d.kf_p0 = (double *)(a + o_k0); d.kf_p1 = (double *)(a + o_k1);
    d.kf_pos = (int32_t *)(a + o_kp); d.kf_scratch = (double *)(a + o_ks); d.kf_pose = (double *)(a + o_kpose); d.kf_x = (double *)(a + o_kx);
    d.kf_info = (double *)(a + o_kinfo); d.kf_status = (int32_t *)(a + o_kst); d.mask = (uint8_t *)(a + o_mask); d.gf_xy = (float *)(a + o_gxy);
    d.gf_n = (int32_t *)(a + o_gn); d.res = (double *)(a + o_res); d.kf_hand = (double *)(a + o_hand);
    s->ws_lk = a + o_wl; s->ws_gftt = a + o_wg; s->ws_pnp = a + o_wp;
    mqs_pnp_workspace_layout(s->ws_pnp, kHyp, &d.pnp_poses, &d.pnp_counts, &d.pnp_inl);
    // (the highest stream priority, like the side stream of mqs_slam_set_next: their hardware queues then come from a pool nothing else in the
    // process draws from -- a default-priority stream, the ingest's upload stream for one, cannot land on the loop's queue and put its
    // copy / event packets between the loop's kernels; measured inside bench.py: with-upload / resident 0.87-0.89 -> see DESIGN.md section 0)
    {
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        s->stream_priority = greatest;
        if (const char *pe = getenv("MQS_SLAM_STREAM_PRIORITY")) s->stream_priority = (pe[0] == 'n' || pe[0] == '0') ? 0 : greatest;      // A/B: "normal"
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, s->stream_priority);
    }
    // (coherent: the decision kernel's block and ticket must be visible to the polling host while later launches are still running)
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->res_host, (size_t)kResSlots * kResStride * 8, hipHostMallocCoherent);
    if (e == hipSuccess) memset(s->res_host, 0xff, (size_t)kResSlots * kResStride * 8);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&d.res_out, s->res_host, 0);
    if (e == hipSuccess) e = hipMemsetAsync(s->arena, 0, o_wl, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d.intr, intr, 72, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        mqs_set_error("mqs_slam_create: %s", hipGetErrorString(e));
        (void)hipFree(s->arena);
        delete s;
        return MQS_E_HIP;
    }
    *out = s;
    return MQS_OK;
}

void mqs_slam_destroy(mqs_slam *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->pyr_stream) {
        (void)hipStreamSynchronize(s->pyr_stream);
        (void)hipStreamSynchronize(s->stream);                     // (a frame enqueued ahead waits for the side stream's events)
        for (int k = 0; k < 2; ++k)
            if (s->prep[k].has_event) (void)hipEventDestroy(s->prep[k].done);
        if (s->prep[0].has_event) { (void)hipEventDestroy(s->spec.done); (void)hipEventDestroy(s->hyp_done); }
        (void)hipStreamDestroy(s->pyr_stream);
    }
    if (s->ws_lk2) (void)hipFree(s->ws_lk2);
    mqs_slam_ingest_release(s);
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(s->arena);
    mqs_slam_ba_release(s);
    if (s->log_arena) (void)hipFree(s->log_arena);
    if (s->re_arena) (void)hipFree(s->re_arena);
    if (s->res_host) (void)hipHostFree(s->res_host);
    (void)hipStreamDestroy(s->stream);
    delete s;
}

// The observation log (off by default): capacity in observations; before mqs_slam_start.
int mqs_slam_log_enable(mqs_slam *s, int64_t capacity)
{
    MQS_ARG_CHECK(s != nullptr && capacity >= 64 && capacity < (1ll << 30), "handle; 64 <= capacity < 2^30");
    MQS_ARG_CHECK(!s->started && s->log_arena == nullptr, "before mqs_slam_start, once");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    const int traj_cap = 65536;                     // accepted frames whose pose is kept for the in-loop adjuster (96 bytes each)
    const size_t traj_off = ((size_t)capacity * (4 + 4 + 16 + 4) + 255) & ~size_t(255);
    const size_t bytes = traj_off + (size_t)traj_cap * 96;
    hipError_t e = hipMalloc((void **)&s->log_arena, bytes);
    if (e != hipSuccess) { s->log_arena = nullptr; mqs_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return MQS_E_NOMEM; }
    s->d.log_uv = reinterpret_cast<double *>(s->log_arena);
    s->d.log_lm = reinterpret_cast<int32_t *>(s->log_arena + (size_t)capacity * 16);
    s->d.log_pose = s->d.log_lm + capacity;
    s->d.tid2lm = s->d.log_pose + capacity;         // track ids never outnumber the observations
    s->d.log_cap = (int)capacity;
    s->d.tid_cap = (int)capacity;
    s->d.traj = reinterpret_cast<double *>(s->log_arena + traj_off);
    s->d.traj_cap = traj_cap;
    MQS_HIP_CHECK(hipMemsetAsync(s->d.tid2lm, 0xff, (size_t)capacity * 4, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    return MQS_OK;
}

// the log so far (synchronises the handle's stream): lm / pose [n] int32, uv [n][2] float64 into host arrays of capacity `cap`
int mqs_slam_read_log(mqs_slam *s, int32_t *lm, int32_t *pose, double *uv, int64_t cap, int64_t *n)
{
    MQS_ARG_CHECK(s != nullptr && n != nullptr && cap >= 0, "handle, n");
    MQS_ARG_CHECK(s->log_arena != nullptr, "mqs_slam_log_enable first");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int32_t cnt[C_COUNT];
    MQS_HIP_CHECK(hipMemcpyAsync(cnt, s->d.cnt, sizeof(cnt), hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n = cnt[C_NLOG];
    if (cnt[C_LOG_OVERFLOW]) {
        mqs_set_error("the observation log is full (%d entries, mqs_slam_log_enable's capacity): observations have been dropped", s->d.log_cap);
        return MQS_E_ARG;
    }
    const int64_t m = cnt[C_NLOG] < cap ? cnt[C_NLOG] : cap;
    if (m > 0) {
        if (lm) MQS_HIP_CHECK(hipMemcpyAsync(lm, s->d.log_lm, (size_t)m * 4, hipMemcpyDeviceToHost, s->stream));
        if (pose) MQS_HIP_CHECK(hipMemcpyAsync(pose, s->d.log_pose, (size_t)m * 4, hipMemcpyDeviceToHost, s->stream));
        if (uv) MQS_HIP_CHECK(hipMemcpyAsync(uv, s->d.log_uv, (size_t)m * 16, hipMemcpyDeviceToHost, s->stream));
        MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
        if (lm) {
            // free-track entries (-2 - track id): the landmark the track became, or -1 while / if it has not
            const int nt = cnt[C_NEXT_TID] < s->d.tid_cap ? cnt[C_NEXT_TID] : s->d.tid_cap;
            int32_t *map = new (std::nothrow) int32_t[nt > 0 ? nt : 1];
            if (!map) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
            hipError_t e = nt > 0 ? hipMemcpyAsync(map, s->d.tid2lm, (size_t)nt * 4, hipMemcpyDeviceToHost, s->stream) : hipSuccess;
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            if (e == hipSuccess)
                for (int64_t k = 0; k < m; ++k)
                    if (lm[k] <= -2) { const int t = -2 - lm[k]; lm[k] = t < nt ? map[t] : -1; }
            delete[] map;
            MQS_HIP_CHECK(e);
        }
    }
    return MQS_OK;
}

// What a bundle adjustment hands back to the loop: the first n landmarks of the map (float64 host array, stored as float32
// values like every landmark, slam2.py:19), the pose of the last accepted frame and of the base keyframe ([R | t] 3x4 world ->
// camera, host; NULL: unchanged).  Waits for the handle's stream first (a keyframe branch may still be writing the map).
int mqs_slam_write_back(mqs_slam *s, const double *map, int n, const double *pose_prev, const double *pose_key)
{
    MQS_ARG_CHECK(s != nullptr && s->started && n >= 0 && (n == 0 || map != nullptr), "handle (started), map");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int32_t cnt[C_COUNT];
    MQS_HIP_CHECK(hipMemcpyAsync(cnt, s->d.cnt, sizeof(cnt), hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    MQS_ARG_CHECK(n <= cnt[C_NLAND], "n <= landmarks in the map");
    if (n > 0) {
        double *tmp = new (std::nothrow) double[(size_t)n * 3];
        if (!tmp) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
        for (size_t k = 0; k < (size_t)n * 3; ++k) tmp[k] = (double)(float)map[k];
        hipError_t e = hipMemcpyAsync(s->d.map, tmp, (size_t)n * 24, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        delete[] tmp;
        MQS_HIP_CHECK(e);
    }
    if (pose_prev) MQS_HIP_CHECK(hipMemcpyAsync(s->d.pose_prev, pose_prev, 96, hipMemcpyHostToDevice, s->stream));
    if (pose_key) MQS_HIP_CHECK(hipMemcpyAsync(s->d.pose_key, pose_key, 96, hipMemcpyHostToDevice, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    return MQS_OK;
}

int mqs_slam_set_thresholds(mqs_slam *s, double max_of_error, double max_lost_tracks_ratio, double max_reproj_error,
                            double max_outlier_ratio, double homography_condition_threshold, int max_homography_points)
{
    MQS_ARG_CHECK(s != nullptr, "handle must not be null");
    MQS_ARG_CHECK(max_homography_points == 0 || max_homography_points >= 4, "max_homography_points: 0 (all tracks) or >= 4");
    s->p.max_homography_points = max_homography_points;
    s->p.max_of_error = max_of_error; s->p.max_lost_ratio = max_lost_tracks_ratio; s->p.max_reproj = max_reproj_error;
    s->p.max_outlier_ratio = max_outlier_ratio; s->p.homography_threshold = homography_condition_threshold;
    return MQS_OK;
}

int mqs_slam_set_second_pass_screen(mqs_slam *s, double max_reproj_error_px)
{
    MQS_ARG_CHECK(s != nullptr && max_reproj_error_px >= 0.0, "handle must not be null, bound >= 0 (0: off)");
    s->p.second_pass_screen_px = max_reproj_error_px;
    return MQS_OK;
}

// first frame (slam2.py:1136-1180): pose from n0 known 3-D points, those points become the first landmarks and tracks, the rest
// of the tracks come from goodFeaturesToTrack under their coverage mask.  objp0 [n0][3], imgp0 [n0][2]: HOST float32.
int mqs_slam_start(mqs_slam *s, const uint8_t *img_dev, const float *objp0, const float *imgp0, int n0, double *pose_out)
{
    MQS_ARG_CHECK(s != nullptr && img_dev && objp0 && imgp0 && pose_out, "pointers must not be null");
    MQS_ARG_CHECK(n0 >= 6 && n0 <= s->p.target, "6 <= n0 <= target_keypoints");
    MQS_ARG_CHECK(n0 <= s->p.max_landmarks, "n0 <= max_landmarks (the start-up landmarks go into the map)");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    // staged through the keyframe-step output arrays (free at this point)
    float *o_dev = reinterpret_cast<float *>(s->d.kf_x), *i_dev = reinterpret_cast<float *>(s->d.kf_scratch);
    MQS_HIP_CHECK(hipMemcpyAsync(o_dev, objp0, (size_t)n0 * 12, hipMemcpyHostToDevice, s->stream));
    MQS_HIP_CHECK(hipMemcpyAsync(i_dev, imgp0, (size_t)n0 * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(start_kernel, dim3(1), dim3(256), 0, s->stream, s->d, o_dev, i_dev, n0);
    int rc = mqs_pnp_refine_dev(s->d.kf_objp, s->d.kf_imgp, n0, nullptr, nullptr, 1, s->d.intr, nullptr, 0, kPnpIters, kPnpEps, s->d.pose_r,
                                s->d.pnp_info, s->stream);
    if (rc != MQS_OK) return rc;
    hipLaunchKernelGGL(start_pose_kernel, dim3(1), dim3(64), 0, s->stream, s->d);
    rc = topup(s, img_dev);
    if (rc != MQS_OK) return rc;
    rc = mqs_slam_ba_anchor(s, n0);                    // with the log on: where the in-loop adjuster's gauge sits
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipMemcpyAsync(s->res_host, s->d.res, kRes * 8, hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    memcpy(pose_out, s->res_host + R_POSE, 96);
    s->started = true;
    s->pre.valid = false; s->spec.valid = false; s->last_decision = 0;
    s->accepted = 1;                  // the first frame is pose 0 and the first base keyframe
    s->base_pose = 0;
    s->land_ub = n0;
    s->key_pose = 0;
    return MQS_OK;
}

// One frame.  result [40] doubles: decision (0 rejected, 1 frame, 2 keyframe), reason of a rejection (1 lost tracks, 2 fewer
// than 8 landmark tracks, 3 no RANSAC model, 4 outlier ratio, 5 reprojection error), tracks kept, landmark tracks, inliers,
// |old|, |new| point sets of the keyframe step, lost ratio, outlier ratio, reprojection RMS, homography w0 / w2, landmarks,
// pose [12] (3x4 world -> camera; of a keyframe: before its refinement); from [24]: what the PREVIOUS call's keyframe branch
// left -- valid flag, landmarks added, tracks after the top-up, landmarks, refined pose [12] -- because that branch runs
// behind the call that started it.  mqs_slam_flush returns the same block once the stream has drained.
static int prepare_next_into(mqs_slam *s, const uint8_t *prev_img_dev, int prev_slot, const uint8_t *next_img_dev, int next_slot, int ws_target);

// The host's wait for a frame's result: the ticket word behind the block in the pinned ring (written by the decision kernel with a
// system-scope release).  The stream is looked at every few thousand polls -- a launch that failed never writes its ticket -- and the
// wait gives up after ten seconds.
static int wait_result(mqs_slam *s, unsigned ticket)
{
    const unsigned long long *word = (const unsigned long long *)(s->res_host + (size_t)(ticket & (kResSlots - 1)) * kResStride + kRes);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == (unsigned long long)ticket) return MQS_OK;
        if ((spins & 4095u) == 0) {
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) {                                   // drained: the ticket is there, or the launch never ran
                if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == (unsigned long long)ticket) return MQS_OK;
                mqs_set_error("mqs_slam_track: the stream drained without the frame's result (ticket %u)", ticket);
                return MQS_E_HIP;
            }
            if (q != hipErrorNotReady) { mqs_set_error("mqs_slam_track: %s", hipGetErrorString(q)); return MQS_E_HIP; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
                mqs_set_error("mqs_slam_track: no result after 10 s (ticket %u)", ticket);
                return MQS_E_TIMEOUT;
            }
        }
        __builtin_ia32_pause();
    }
}

int mqs_slam_track(mqs_slam *s, const uint8_t *prev_img_dev, const uint8_t *img_dev, double *result)
{
    MQS_ARG_CHECK(s != nullptr && prev_img_dev && img_dev && result, "pointers must not be null");
    MQS_ARG_CHECK(s->started, "mqs_slam_start first");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    SlamDev &d = s->d;
    s->p.pose_index = s->accepted;
    s->p.base_pose_index = s->base_pose;
    // The tracker's share of this frame:
    //  * it ran AHEAD (below, behind the frame before's hypotheses, under its decision kernel: 46 of a frame's ~150 us): valid when that
    //    frame was accepted as a plain frame -- a keyframe's branch rewrites the tracks, a rejected frame's successor is tracked from the
    //    frame before it -- and this is the pair it ran on.  The loop's stream waits for it; the filter reads its results through the
    //    commit's map (SlamParams::spec);
    //  * else the pair's pyramid may have been prepared ahead on the side stream (mqs_slam_prepare_next): the loop's stream waits for
    //    that launch and runs the tracker alone;
    //  * else pyramid and tracker here.  A workspace that holds a pyramid prepared for the FOLLOWING pair is left alone.
    int ws = -1, phases = 3, rc = MQS_OK;
    // Was this frame ENQUEUED by the call before (mqs_slam_pipeline)?  Its kernels ran -- or are running -- if the frame in front was an
    // ordinary one; else they looked at that frame's decision and did nothing, and the frame is issued here as usual.
    bool launched = false;
    unsigned ticket = 0;
    if (s->pre.valid) {
        s->pre.valid = false;
        if (s->last_decision == 1) {
            if (s->pre.prev != prev_img_dev || s->pre.img != img_dev) {
                mqs_set_error("mqs_slam_track: with mqs_slam_pipeline on, the frame named by mqs_slam_set_next has been enqueued behind an ordinary frame -- it must be the next one tracked");
                return MQS_E_ARG;
            }
            launched = true; ws = s->pre.ws; ticket = s->pre.ticket;
            s->spec.valid = false;
            s->prep[ws].valid = false;                                // (its pyramid has been used: the pair was tracked on the side stream)
        }
    }
    if (!launched) {
        const bool use_spec = s->spec.valid && s->spec_enabled && s->last_decision == 1 && s->spec.prev == prev_img_dev && s->spec.next == img_dev;
        s->spec.valid = false;
        s->p.spec = use_spec ? 1 : 0;
        s->p.gated = 0;
        s->p.ticket = ticket = ++s->ticket;
        if (use_spec) {
            ws = s->spec.ws;
            MQS_HIP_CHECK(hipStreamWaitEvent(s->stream, s->spec.done, 0));
            s->prep[ws].valid = false;
        } else {
            for (int k = 0; k < 2; ++k)
                if (s->prep[k].valid && s->prep[k].prev == prev_img_dev && s->prep[k].next == img_dev) { ws = k; phases = 2; }
            if (ws < 0) ws = (s->prep[0].valid && s->prep[0].prev == img_dev) ? 1 : 0;
            if (ws == 1 && !s->ws_lk2) ws = 0;
            if (s->prep[ws].has_event && (s->prep[ws].valid || phases == 2)) MQS_HIP_CHECK(hipStreamWaitEvent(s->stream, s->prep[ws].done, 0));
            s->prep[ws].valid = false;
            rc = mqs_slam_ingest_main_wait(s, prev_img_dev);       // (the tracker inside the call reads both images on the loop's stream)
            if (rc == MQS_OK) rc = mqs_slam_ingest_main_wait(s, img_dev);
            if (rc != MQS_OK) return rc;
            rc = mqs_lk_launch(prev_img_dev, img_dev, s->p.W, s->p.H, d.pts, kMaxTracks, d.cnt + C_N, 21, 21, 3, 30, 0.01, 1e-4, d.lk_pts,
                               d.lk_st, d.lk_err, ws == 0 ? s->ws_lk : s->ws_lk2, s->ws_lk_bytes, s->stream, phases);
            if (rc != MQS_OK) return rc;
        }
        if (s->fused_filter) {
            hipLaunchKernelGGL(frame_hypothesis_kernel, dim3(kHyp), dim3(64), 0, s->stream, d, s->p);
        } else {
            hipLaunchKernelGGL(frame_filter_kernel, dim3(1), dim3(256), 0, s->stream, d, s->p);
            rc = mqs_pnp_ransac_launch(d.objp_t, d.imgp_t, kMaxTracks, d.cnt + C_NTRI, d.intr, d.samples, kHyp, kSample, s->p.max_reproj,
                                       kSampleIters, kPnpIters, kPnpEps, d.pose_r, d.sel, d.inl_mask, d.pnp_info, s->ws_pnp, s->stream, 1);
            if (rc != MQS_OK) return rc;
        }
        if (s->pyr_stream) MQS_HIP_CHECK(hipEventRecord(s->hyp_done, s->stream));      // this frame's kept tracks (t_pts, C_NKEEP) are written
        hipLaunchKernelGGL(frame_decide_kernel, dim3(s->decide_grid), dim3(256), (size_t)kMaxTracks * 40, s->stream, d, s->p);
        MQS_HIP_CHECK(hipGetLastError());
    }
    // the NEXT pair's pyramid (mqs_slam_set_next): enqueued on the side stream now, behind this frame's launches -- the host's work for it
    // runs while the device is busy with them, the kernel runs under the pose kernels
    if (s->ahead.set) {
        s->ahead.set = false;
        const bool had_stream = s->pyr_stream != nullptr;
        const int wn = 1 - ws;                                       // (not the workspace this frame's tracker is reading)
        rc = prepare_next_into(s, img_dev, s->ahead.prev_slot, s->ahead.next, s->ahead.next_slot, wn);
        if (rc != MQS_OK) return rc;
        // ... and the NEXT frame's tracker on that pyramid, from this frame's kept tracks: behind this frame's hypotheses (they wrote the kept
        // tracks), beside its decision kernel.  Which of these tracks the commit keeps, the next frame's filter learns from spec_map.
        if (s->spec_enabled && had_stream) {
            MQS_HIP_CHECK(hipStreamWaitEvent(s->pyr_stream, s->hyp_done, 0));
            rc = mqs_lk_launch(img_dev, s->prep[wn].next, s->p.W, s->p.H, d.t_pts, kMaxTracks, d.cnt + C_NKEEP, 21, 21, 3, 30, 0.01, 1e-4, d.lk_pts_b,
                               d.lk_st_b, d.lk_err_b, wn == 0 ? s->ws_lk : s->ws_lk2, s->ws_lk_bytes, s->pyr_stream, 2);
            if (rc != MQS_OK) return rc;
            MQS_HIP_CHECK(hipEventRecord(s->spec.done, s->pyr_stream));
            s->spec.valid = true; s->spec.prev = img_dev; s->spec.next = s->prep[wn].next; s->spec.ws = wn;
            // ... and (mqs_slam_pipeline) the next frame's pose kernels themselves, behind this frame's decision on the loop's stream: they read
            // that decision on the device and run only behind an ordinary frame.  What the host would pass them then: one more accepted
            // frame, the same base keyframe, the tracker's results from the side stream.
            if (s->pipeline && s->fused_filter) {
                SlamParams pn = s->p;
                pn.pose_index = s->accepted + 1;
                pn.spec = 1; pn.gated = 1; pn.ticket = ++s->ticket;
                MQS_HIP_CHECK(hipStreamWaitEvent(s->stream, s->spec.done, 0));
                hipLaunchKernelGGL(frame_hypothesis_kernel, dim3(kHyp), dim3(64), 0, s->stream, d, pn);
                MQS_HIP_CHECK(hipEventRecord(s->hyp_done, s->stream));
                hipLaunchKernelGGL(frame_decide_kernel, dim3(s->decide_grid), dim3(256), (size_t)kMaxTracks * 40, s->stream, d, pn);
                MQS_HIP_CHECK(hipGetLastError());
                s->pre.valid = true; s->pre.prev = img_dev; s->pre.img = s->prep[wn].next; s->pre.ws = wn; s->pre.ticket = pn.ticket;
            }
        }
    }
    // The decision kernel writes the result block into its slot of the pinned ring and the ticket behind it; the host polls for the ticket
    // (a stream synchronisation would also wait for the frame enqueued ahead, and wakes up 10-20 us late).
    rc = wait_result(s, ticket);
    if (rc != MQS_OK) return rc;
    memcpy(result, s->res_host + (size_t)(ticket & (kResSlots - 1)) * kResStride, kRes * 8);
    if (result[R_DECISION] == -3.0) {
        s->last_decision = 0; s->pre.valid = false; s->spec.valid = false;
        mqs_set_error("mqs_slam_track: the decision kernel's keyframe-test workgroup did not report within 0.5 s (nothing was committed)");
        return MQS_E_TIMEOUT;
    }
    s->last_decision = (int)result[R_DECISION];
    if (result[R_DECISION] >= 1.0) s->accepted += 1;
    if (result[R_DECISION] == 2.0) {
        s->base_pose = s->p.pose_index;                 // (the kernels below still get the OLD base through s->p.base_pose_index)
        s->key_pose = s->p.pose_index;
        const int n_old = (int)result[R_NOLD], n_new = (int)result[R_NNEW];
        s->land_ub = (int)result[R_NLAND] + n_new;      // the keyframe step adds at most its free tracks
        if (s->land_ub > s->p.max_landmarks) s->land_ub = s->p.max_landmarks;
        if (n_new > 0) {
            rc = mqs_keyframe_step_launch(d.kf_objp, d.kf_imgp, n_old, d.kf_p0, d.kf_p1, n_new, d.intr, d.pose_prev, d.pose_key, 3.e-5,
                                          kPnpIters, kPnpEps, s->p.second_pass_screen_px, d.kf_scratch, d.kf_pose, d.kf_x, d.kf_status, d.kf_info,
                                          s->stream);
            if (rc != MQS_OK) return rc;
        }
        hipLaunchKernelGGL(keyframe_commit_kernel, dim3(1), dim3(256), 0, s->stream, d, s->p, n_new);
        rc = topup(s, img_dev);
        if (rc != MQS_OK) return rc;
    }
    return MQS_OK;
}

// The pyramid of the NEXT image pair, ahead of its frame: enqueued on a stream of its own, it runs under the current frame's pose kernels
// (RANSAC hypotheses, selection, decision: one to 256 small workgroups for ~90 us, most of the chip idle).  Call it BEFORE the
// mqs_slam_track of the current frame (which blocks for that frame's result); the mqs_slam_track of the next frame finds the pyramid by
// the two image pointers and runs the tracker alone.  A frame that is rejected in between (its predecessor stays the previous image) simply
// does not match: the pyramid is built in the call as before.  prev_slot / next_slot >= 0: ring slots of the frame ingest -- the side stream
// waits for their uploads (next_img_dev may then be NULL); -1: the caller vouches for the image.  Same results either way, bit for bit.
int mqs_slam_prepare_next(mqs_slam *s, const uint8_t *prev_img_dev, int prev_slot, const uint8_t *next_img_dev, int next_slot)
{
    return prepare_next_into(s, prev_img_dev, prev_slot, next_img_dev, next_slot, -1);
}

// ws_target: the workspace to build into, or -1: the one that is not reserved for the frame about to be tracked
static int prepare_next_into(mqs_slam *s, const uint8_t *prev_img_dev, int prev_slot, const uint8_t *next_img_dev, int next_slot, int ws_target)
{
    MQS_ARG_CHECK(s != nullptr && s->started && prev_img_dev != nullptr, "handle (started), previous image");
    MQS_ARG_CHECK(next_img_dev != nullptr || next_slot >= 0, "the next image: a device pointer or a ring slot");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    if (!s->pyr_stream) {
        // A hardware queue of its own: the runtime deals its few hardware queues (four by default) to the streams of a process as they
        // come and go, and in a process that has created and destroyed many streams this one can land on the SAME queue as the loop's
        // stream -- everything enqueued "beside" the pose kernels then runs behind them (measured inside bench.py: the loop at the
        // rate of the form without any of it, results the same).  Queues of different priority are different queues, and the priority is
        // the right one anyway: the tracker ahead is what the next frame's hypotheses wait for.
        {
            int least = 0, greatest = 0;
            MQS_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            MQS_HIP_CHECK(hipStreamCreateWithPriority(&s->pyr_stream, hipStreamNonBlocking, s->stream_priority == 0 ? 0 : greatest));
        }
        hipError_t e = hipMalloc(&s->ws_lk2, (size_t)s->ws_lk_bytes);
        if (e != hipSuccess) { s->ws_lk2 = nullptr; mqs_set_error("hipMalloc(%lld) failed: %s", (long long)s->ws_lk_bytes, hipGetErrorString(e)); return MQS_E_NOMEM; }
        for (int k = 0; k < 2; ++k) {
            MQS_HIP_CHECK(hipEventCreateWithFlags(&s->prep[k].done, hipEventDisableTiming));
            s->prep[k].has_event = true;
        }
        MQS_HIP_CHECK(hipEventCreateWithFlags(&s->spec.done, hipEventDisableTiming));
        MQS_HIP_CHECK(hipEventCreateWithFlags(&s->hyp_done, hipEventDisableTiming));
    }
    hipEvent_t up;
    if (prev_slot >= 0) {
        const uint8_t *p = nullptr;
        MQS_ARG_CHECK(mqs_slam_ingest_slot(s, prev_slot, &p, &up) && p == prev_img_dev, "prev_slot does not hold the previous image");
        MQS_HIP_CHECK(hipStreamWaitEvent(s->pyr_stream, up, 0));
    }
    if (next_slot >= 0) {
        MQS_ARG_CHECK(mqs_slam_ingest_slot(s, next_slot, &next_img_dev, &up), "nothing was uploaded into next_slot");
        MQS_HIP_CHECK(hipStreamWaitEvent(s->pyr_stream, up, 0));
    }
    // the workspace that is NOT reserved for the frame about to be tracked (whose image is this pair's previous one); what the loop's
    // stream last read from it has completed: every mqs_slam_track waits for its frame's result
    const int ws = ws_target >= 0 ? ws_target : ((s->prep[0].valid && s->prep[0].next == prev_img_dev) ? 1 : 0);
    int rc = mqs_lk_launch(prev_img_dev, next_img_dev, s->p.W, s->p.H, nullptr, 0, nullptr, 21, 21, 3, 30, 0.01, 1e-4, nullptr, nullptr, nullptr,
                           ws == 0 ? s->ws_lk : s->ws_lk2, s->ws_lk_bytes, s->pyr_stream, 1);
    if (rc != MQS_OK) return rc;
    MQS_HIP_CHECK(hipEventRecord(s->prep[ws].done, s->pyr_stream));
    s->prep[ws].valid = true; s->prep[ws].prev = prev_img_dev; s->prep[ws].next = next_img_dev;
    return MQS_OK;
}

// on = 1: a frame named by mqs_slam_set_next is not only prepared (pyramid, tracker) but ENQUEUED -- its hypotheses and decision behind
// the current frame's decision on the loop's stream, before the call waits for the current frame's result.  The kernels look at the
// current frame's decision on the device: behind an ordinary frame they run (no host round trip between the two frames: ~30 us of a
// frame's ~125), behind a keyframe or a rejected frame they do nothing and the next mqs_slam_track issues the frame as usual.  The
// contract: after a call that returned an ordinary frame, the next mqs_slam_track MUST be for (this image, the named next image)
// (MQS_E_ARG otherwise), and whatever reads the handle's state between the two calls may already see that frame in it.  Same results.
int mqs_slam_pipeline(mqs_slam *s, int on)
{
    MQS_ARG_CHECK(s != nullptr, "handle");
    MQS_ARG_CHECK(!(s->pre.valid && s->last_decision == 1), "a frame is enqueued: track it first");
    s->pipeline = on != 0;
    return MQS_OK;
}

// The frame behind the one the next mqs_slam_track handles: that call then prepares the pair (its image, this one) behind its own launches.
int mqs_slam_set_next(mqs_slam *s, int this_slot, const uint8_t *next_img_dev, int next_slot)
{
    MQS_ARG_CHECK(s != nullptr && (next_img_dev != nullptr || next_slot >= 0), "handle; the next image: a device pointer or a ring slot");
    s->ahead.set = true; s->ahead.next = next_img_dev; s->ahead.prev_slot = this_slot; s->ahead.next_slot = next_slot;
    return MQS_OK;
}

// Behind a keyframe (its top-up is enqueued on the handle's stream): the new corners matched against the projections of the
// landmarks that are no longer tracked -- mqs_match_knn2_f32_dev on pixel coordinates, then mqs_match_ratio_unique_dev
// (radius, ratio test, one landmark per corner; slam.py:101-125) -- and the matched free tracks given their landmark back.
// *n_matched: how many (synchronises the stream).
int mqs_slam_reassociate(mqs_slam *s, float max_radius, double max_dist_ratio, int32_t *n_matched)
{
    MQS_ARG_CHECK(s != nullptr && s->started && n_matched != nullptr, "handle (started), n_matched");
    MQS_ARG_CHECK(max_radius > 0.f && max_dist_ratio > 0.0, "max_radius, max_dist_ratio > 0");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    const int nq = s->p.max_landmarks, nt = kMaxTracks;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_q = 0, o_t = up((size_t)nq * 8), o_idx = o_t + up((size_t)nt * 8), o_dist = o_idx + up((size_t)nq * 8),
                 o_qot = o_dist + up((size_t)nq * 8), o_trk = o_qot + up((size_t)nt * 4), o_n = o_trk + up((size_t)nq),
                 o_ws = o_n + 256, ws_bytes = (size_t)mqs_match_ratio_unique_workspace_bytes(nt), total = o_ws + up(ws_bytes);
    if (!s->re_arena) {
        hipError_t e = hipMalloc((void **)&s->re_arena, total);
        if (e != hipSuccess) { s->re_arena = nullptr; mqs_set_error("hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); return MQS_E_NOMEM; }
    }
    char *a = s->re_arena;
    float *q_xy = (float *)(a + o_q), *t_xy = (float *)(a + o_t), *dist = (float *)(a + o_dist);
    int32_t *idx = (int32_t *)(a + o_idx), *qot = (int32_t *)(a + o_qot), *n_dev = (int32_t *)(a + o_n);
    uint8_t *tracked = (uint8_t *)(a + o_trk);
    s->p.pose_index = s->accepted - 1;              // the keyframe this call stands behind
    MQS_HIP_CHECK(hipMemsetAsync(tracked, 0, (size_t)nq, s->stream));
    hipLaunchKernelGGL(reassoc_mark_kernel, dim3((kMaxTracks + 255) / 256), dim3(256), 0, s->stream, s->d, tracked);
    hipLaunchKernelGGL(reassoc_rows_kernel, dim3((unsigned)((nq > nt ? nq : nt) + 255) / 256), dim3(256), 0, s->stream, s->d, s->p, q_xy, nq, t_xy, tracked);
    int rc = mqs_match_knn2_f32_dev(q_xy, nq, t_xy, nt, 2, idx, dist, s->stream);
    if (rc != MQS_OK) return rc;
    rc = mqs_match_ratio_unique_dev(idx, dist, nq, nt, max_radius, max_dist_ratio, nullptr, qot, nullptr, a + o_ws, (int64_t)ws_bytes, s->stream);
    if (rc != MQS_OK) return rc;
    hipLaunchKernelGGL(reassoc_apply_kernel, dim3(1), dim3(256), 0, s->stream, s->d, s->p, qot, n_dev);
    MQS_HIP_CHECK(hipGetLastError());
    MQS_HIP_CHECK(hipMemcpyAsync(n_matched, n_dev, 4, hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    return MQS_OK;
}

int mqs_slam_flush(mqs_slam *s, double *result)
{
    MQS_ARG_CHECK(s != nullptr && result, "pointers must not be null");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    MQS_HIP_CHECK(hipMemcpyAsync(s->res_host, s->d.res, kRes * 8, hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipMemsetAsync(s->d.res + R_KF_VALID, 0, 8, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    memcpy(result, s->res_host, kRes * 8);
    return MQS_OK;
}

// the live tracks (host arrays of capacity `cap`): positions, base-keyframe positions, landmark id or -1, track id
int mqs_slam_read_tracks(mqs_slam *s, float *pts, float *base, int32_t *lm, int32_t *tid, int cap, int32_t *n)
{
    MQS_ARG_CHECK(s != nullptr && n != nullptr && cap >= 0, "handle, n");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int32_t cnt[C_COUNT];
    MQS_HIP_CHECK(hipMemcpyAsync(cnt, s->d.cnt, sizeof(cnt), hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    const int m = cnt[C_N] < cap ? cnt[C_N] : cap;
    *n = cnt[C_N];
    if (m > 0) {
        if (pts) MQS_HIP_CHECK(hipMemcpyAsync(pts, s->d.pts, (size_t)m * 8, hipMemcpyDeviceToHost, s->stream));
        if (base) MQS_HIP_CHECK(hipMemcpyAsync(base, s->d.base, (size_t)m * 8, hipMemcpyDeviceToHost, s->stream));
        if (lm) MQS_HIP_CHECK(hipMemcpyAsync(lm, s->d.lm, (size_t)m * 4, hipMemcpyDeviceToHost, s->stream));
        if (tid) MQS_HIP_CHECK(hipMemcpyAsync(tid, s->d.tid, (size_t)m * 4, hipMemcpyDeviceToHost, s->stream));
        MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    }
    return MQS_OK;
}

// the map as the reference keeps it: float32 [n][3]
int mqs_slam_read_map(mqs_slam *s, float *objp, int cap, int32_t *n)
{
    MQS_ARG_CHECK(s != nullptr && n != nullptr && cap >= 0, "handle, n");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    int32_t cnt[C_COUNT];
    MQS_HIP_CHECK(hipMemcpyAsync(cnt, s->d.cnt, sizeof(cnt), hipMemcpyDeviceToHost, s->stream));
    MQS_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n = cnt[C_NLAND];
    const int m = cnt[C_NLAND] < cap ? cnt[C_NLAND] : cap;
    if (m > 0 && objp) {
        double *tmp = new (std::nothrow) double[(size_t)m * 3];
        if (!tmp) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
        hipError_t e = hipMemcpyAsync(tmp, s->d.map, (size_t)m * 24, hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e == hipSuccess)
            for (size_t k = 0; k < (size_t)m * 3; ++k) objp[k] = (float)tmp[k];
        delete[] tmp;
        MQS_HIP_CHECK(e);
    }
    return MQS_OK;
}

}  // extern "C"
