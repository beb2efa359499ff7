/*
 * mqslam.h -- C ABI of libmqslam_hip.so: the MI355X (gfx950) hot path of
 * Multiple-Quadrotor-SLAM behind the reference's own native-extension boundary.
 *
 * Every entry point returns 0 on success or a negative MQS_E_* code; the message of the
 * last failure on the calling thread is returned by mqs_last_error().  Nothing throws or
 * exits across this boundary.  The caller owns every buffer; the library allocates only
 * scratch that lives inside an mqs_ctx.
 *
 * Reference interfaces replaced (paths relative to the reference repository root):
 *   - Work/python_libs/triangulation_c/triangulation.c:64-83   linear_LS_triangulation
 *     (weave signature `linear_LS_triangulation(u1, P1, u2, P2, x)`,
 *      Work/python_libs/triangulation_c/__init__.py:45)
 *   - Work/python_libs/triangulation_c/triangulation.c:103-161 iterative_LS_triangulation
 *     (weave signature `iterative_LS_triangulation(u1, P1, u2, P2, tolerance, x, x_status)`,
 *      Work/python_libs/triangulation_c/__init__.py:84)
 *   - Work/python_libs/triangulation.py:20  cv2.triangulatePoints call of
 *     linear_eigen_triangulation
 *   - Work/python_libs/cv2_helpers.py:300-306  cv2.batchDistance call of BFMatcher.radiusMatch
 *   - Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:289-298,323-324  projection-factor
 *     graph + LevenbergMarquardtOptimizer::optimize (GTSAM 3.2.1): linearise, normal
 *     equations, solve.
 *
 * Array layouts (all row-major, float64 unless stated):
 *   u      [C][N][2]   normalised image coordinates, camera-major
 *   P      [C][3][4]   camera matrices (top three rows; world -> camera)
 *   x      [N][3]      triangulated points
 *   status [N] int32   iterative-LS status code (see mqs_triangulate_iterative_ls)
 *   ok     [N] uint8   linear-eigen finite flag
 * N-view generalisation: SURVEY.md Appendix C; C == 2 reproduces the reference exactly.
 */
#ifndef MQSLAM_H
#define MQSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQS_OK          0
#define MQS_E_ARG      -1   /* bad argument (null pointer, C out of range, misaligned device ptr) */
#define MQS_E_HIP      -2   /* HIP runtime error (message in mqs_last_error) */
#define MQS_E_NOMEM    -3   /* device or host allocation failed */
#define MQS_E_NODEVICE -4   /* no gfx950 device visible */
#define MQS_E_RCCL     -5   /* RCCL missing or a collective failed (message in mqs_last_error) */
#define MQS_E_TIMEOUT  -6   /* a bounded device-side wait gave up (a peer's row, or the finalizer pieces of an iteration): results invalid */
#define MQS_E_CAPACITY -7   /* a resident working set does not hold the problem (mqs_slam_bundle_adjust: a list or table is full); nothing was written, the caller may take another path */

#define MQS_MAX_CAMS    8
#define MQS_TRI_MAX_ITER_DEFAULT 10        /* triangulation.c:125 */
#define MQS_TRI_TOL_DEFAULT      3.e-5     /* triangulation_c/__init__.py:51 */
#define MQS_TRI_MAX_COORD_DEFAULT 1.e16    /* triangulation.py:6 */

typedef struct mqs_ctx mqs_ctx;            /* one ctx <-> one device <-> one stream; one ctx per thread */

const char *mqs_last_error(void);
const char *mqs_version(void);
int  mqs_device_count(void);

/* Context: owns device staging buffers used by the host-pointer entry points. */
int  mqs_create(int device_id, mqs_ctx **out);
void mqs_destroy(mqs_ctx *ctx);
int  mqs_synchronize(mqs_ctx *ctx);

/* ---------------------------------------------------------------------------------------
 * Triangulation, host pointers (H2D copy, kernel, D2H copy; synchronous).
 * ------------------------------------------------------------------------------------- */

/* T1: x = argmin |A x - b| of the 2C x 3 inhomogeneous DLT system (triangulation.c:65-83). */
int mqs_triangulate_linear_ls(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N,
                              double *x);

/* T2: Hartley-Sturm iterative re-weighted LS exactly as triangulation.c:104-161:
 * status = (iters < max_iter && all d_c > 0); for each c with d_c <= 0: status -= 2^c. */
int mqs_triangulate_iterative_ls(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N,
                                 double tolerance, int max_iter, double *x, int32_t *status);

/* T3: homogeneous 3C x 4 DLT of cv2.triangulatePoints, x = X[0:3]/X[3],
 * ok = all |x_i| <= max_coord (triangulation.py:20-23). */
int mqs_triangulate_linear_eigen(mqs_ctx *ctx, const double *u, const double *P, int C, int64_t N,
                                 double max_coord, double *x, uint8_t *ok);

/* Two-view forms with the reference extension's exact argument order
 * (triangulation_c/__init__.py:45,84).  P1/P2 may be 3x4 or 4x4 row-major: only the first
 * 12 doubles are read, as in triangulation.c:24-25. */
int mqs_linear_LS_triangulation(mqs_ctx *ctx, const double *u1, const double *P1, const double *u2,
                                const double *P2, int64_t N, double *x);
int mqs_iterative_LS_triangulation(mqs_ctx *ctx, const double *u1, const double *P1, const double *u2,
                                   const double *P2, int64_t N, double tolerance, double *x,
                                   int32_t *x_status);

/* ---------------------------------------------------------------------------------------
 * Triangulation, device pointers (asynchronous on `stream`, a hipStream_t passed as void*;
 * NULL = the null stream).  All device pointers must be 16-byte aligned.  P is a DEVICE
 * pointer to [C][3][4].
 * ------------------------------------------------------------------------------------- */
int mqs_triangulate_linear_ls_dev(const double *u, const double *P, int C, int64_t N, double *x,
                                  void *stream);
int mqs_triangulate_iterative_ls_dev(const double *u, const double *P, int C, int64_t N,
                                     double tolerance, int max_iter, double *x, int32_t *status,
                                     void *stream);
int mqs_triangulate_linear_eigen_dev(const double *u, const double *P, int C, int64_t N,
                                     double max_coord, double *x, uint8_t *ok, void *stream);
/* float32 observations u [C][N][2] (slam2.py works in float32, slam2.py:19; the reference's wrapper widens them on the
 * host, triangulation_c/__init__.py:32-33): widened on load, same results as the float64 entry points on the widened
 * array, half the input traffic.  kind: 0 linear_ls, 1 iterative_ls, 2 linear_eigen; status / ok as for those. */
int mqs_triangulate_f32_dev(int kind, const float *u, const double *P, int C, int64_t N, double tolerance, int max_iter,
                            double max_coord, double *x, int32_t *status, uint8_t *ok, void *stream);
/* The same through the host-pointer boundary: packed u [C][N][2] float32, and the reference's 2-view argument order. */
int mqs_triangulate_f32(mqs_ctx *ctx, int kind, const float *u, const double *P, int C, int64_t N, double tolerance,
                        int max_iter, double max_coord, double *x, int32_t *status, uint8_t *ok);
int mqs_triangulation_2view_f32(mqs_ctx *ctx, int kind, const float *u1, const double *P1, const float *u2, const double *P2,
                                int64_t N, double tolerance, double max_coord, double *x, int32_t *status, uint8_t *ok);
/* linear_LS AND iterative_LS of the same observations in one pass (the reference's harness calls every method on the
 * same inputs, triangulation_comparison.py:590): the first solve of the iteration, with unit weights, IS the linear-LS
 * system (triangulation.c:65-83 vs :104-130), so x_ls costs one extra refinement step instead of a second read of the
 * observations.  x_it / status as mqs_triangulate_iterative_ls_dev; x_ls equals mqs_triangulate_linear_ls_dev to rounding
 * (the Gram matrix is summed camera by camera here).  max_iter >= 1. */
int mqs_triangulate_ls_and_iterative_dev(const double *u, const double *P, int C, int64_t N, double tolerance,
                                         int max_iter, double *x_ls, double *x_it, int32_t *status, void *stream);

/* Fused form: observations given in PIXELS [C][N][2] with per-camera intrinsics intr [C][9]
 * (fx fy cx cy k1 k2 p1 p2 k3); undistort + normalise (cv2.undistortPoints, slam2.py:551-552) happens on
 * load inside the triangulation kernel.  kind: 0 linear_ls, 1 iterative_ls, 2 linear_eigen; status / ok
 * may be NULL when the kind does not produce them.  Bitwise equal to mqs_undistort_points_dev followed by
 * the matching mqs_triangulate_*_dev. */
int mqs_triangulate_pixels_dev(int kind, const double *pixels, const double *intr, const double *P, int C,
                               int64_t N, double tolerance, int max_iter, double max_coord, double *x,
                               int32_t *status, uint8_t *ok, void *stream);

/* ---------------------------------------------------------------------------------------
 * Brute-force descriptor matching (replaces cv2.batchDistance + the per-query loop of
 * cv2_helpers.py:296-339): for every query row the two nearest train rows under L2, ties
 * broken towards the LOWER train index.  idx [Nq][2] int32 (-1 when Nt < 2),
 * dist [Nq][2] float32 (L2 distance, not squared; +inf where idx == -1).
 * ------------------------------------------------------------------------------------- */

/* Exact float32 path, any D >= 1 (the reference's use: D == 2 pixel coordinates).
 * dist^2 accumulated in float32 in dimension order without fused multiply-add, then sqrtf. */
int mqs_match_knn2_f32(mqs_ctx *ctx, const float *query, int64_t Nq, const float *train, int64_t Nt,
                       int D, int32_t *idx, float *dist);
int mqs_match_knn2_f32_dev(const float *query, int64_t Nq, const float *train, int64_t Nt, int D,
                           int32_t *idx, float *dist, void *stream);

/* MFMA path for binary descriptors expanded to {0,1} fp16 (IEEE half, passed as uint16_t),
 * D in {32, 64, 128, 256, 512}, Nt < 2^31: |q-t|^2 = |q|^2 + |t|^2 - 2 q.t with the contraction on
 * v_mfma_f32_32x32x16_f16; exact because every partial sum is a small integer (squared distances
 * are handled as integers < 2048: inputs must be {0,1}-valued; use the f32 path otherwise). */
int mqs_match_knn2_f16(mqs_ctx *ctx, const uint16_t *query, int64_t Nq, const uint16_t *train,
                       int64_t Nt, int D, int32_t *idx, float *dist);
int mqs_match_knn2_f16_dev(const uint16_t *query, int64_t Nq, const uint16_t *train, int64_t Nt, int D,
                           int32_t *idx, float *dist, void *workspace, int64_t workspace_bytes,
                           void *stream);
int64_t mqs_match_knn2_f16_workspace_bytes(int64_t Nq, int64_t Nt);

/* Packed binary descriptors as ORB / BRIEF produce them (D bits per row = D / 8 bytes, bit k of a descriptor = bit
 * (k & 7) of byte k >> 3; D in {128, 256, 512}): Hamming distance = |q - t|^2 of the {0,1} vectors, contracted on the
 * matrix pipe as FP4 operands (v_mfma_f32_32x32x64_f8f6f4 with E2M1 nibbles 0 / 1 and 0 / -2: every partial sum is a small
 * integer, exact in the fp32 accumulator; round 1 and the first half of round 2 used int8, v_mfma_i32_32x32x32_i8, which
 * csrc/match.hip still carries behind MQS_MATCH_BITS_FP4=0).  Same outputs as the fp16 path: dist = sqrt(Hamming distance),
 * ties towards the lower train index; bit-exact against it on the same descriptors. */
int mqs_match_knn2_bits(mqs_ctx *ctx, const uint8_t *query_bits, int64_t Nq, const uint8_t *train_bits, int64_t Nt, int D,
                        int32_t *idx, float *dist);
int mqs_match_knn2_bits_dev(const uint8_t *query_bits, int64_t Nq, const uint8_t *train_bits, int64_t Nt, int D, int32_t *idx,
                            float *dist, void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_match_knn2_bits_workspace_bytes(int64_t Nq, int64_t Nt, int D);

/* Caller-side filter of the reference's matcher use (Work/SLAM/application/own/slam.py:108-125) on the output of any
 * mqs_match_knn2_*: per query keep the nearest train row when it lies within max_radius (the radius filter of
 * cv2_helpers.py:311-331, compared in float32) and passes the ratio test (it is the only one within the radius, or
 * dist0 / dist1 < max_dist_ratio with the division in double as Python does it; a 0 / 0 ratio fails where the reference
 * raises ZeroDivisionError); then one match per train row: the query with the smallest priority[q] wins (the reference's
 * err_OF; NULL: the match distance), the EARLIER query on equal priority (the reference replaces on strict `<` only).
 * query_of_train [Nt] int32 (-1: unmatched), dist_of_train [Nt] float32 or NULL (+inf where unmatched).
 * Deterministic (64-bit atomicMin on priority|query keys).  workspace: 8-byte aligned. */
int mqs_match_ratio_unique_dev(const int32_t *idx, const float *dist, int64_t Nq, int64_t Nt, float max_radius,
                               double max_dist_ratio, const float *priority, int32_t *query_of_train,
                               float *dist_of_train, void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_match_ratio_unique_workspace_bytes(int64_t Nt);
/* radiusMatch + ratio test + de-duplication in one host-pointer call (slam.py:101-125), exact float32 distances. */
int mqs_match_radius_ratio_unique(mqs_ctx *ctx, const float *query, int64_t Nq, const float *train, int64_t Nt, int D,
                                  float max_radius, double max_dist_ratio, const float *priority, int32_t *query_of_train,
                                  float *dist_of_train);

/* ---------------------------------------------------------------------------------------
 * Bundle adjustment: projection-factor linearisation + landmark Schur complement, the reduced
 * camera solve, and the landmark back-substitution (replaces the GTSAM work behind
 * bundle_adjust.cpp:268-298,323-324).
 *
 *   poses  [C][12]  camera-to-world pose: R (row-major 3x3) then t (3)   (IO.hpp:221-227)
 *   calib  [C][9]   fx fy s u0 v0 k1 k2 p1 p2  (Cal3DS2 order, IO.hpp:230-236)
 *   sigma  [C]      isotropic pixel sigma of camera c's projection factors
 *   points [N][3]   landmarks (world)
 *   obs    [C][N][2] pixel measurements;  mask [C][N] uint8 (NULL = every landmark seen by
 *                   every camera)
 *   prior_w [N]     PriorFactor<Point3> weight 1/sigma^2 per landmark, 0 = none (NULL = none);
 *   prior_xyz [N][3] prior positions (read only where prior_w > 0)     (bundle_adjust.cpp:277-281)
 *
 * Pose tangent order [omega(3), v(3)], right perturbation T*Exp(xi) (GTSAM Pose3).  A factor whose
 * landmark is not in front of its camera contributes the constant residual 2*fx/sigma*(1,1) and
 * zero Jacobians (GenericProjectionFactor, throwCheirality = false).
 * linearize writes out[(6C)*(6C) + 6C + 2] = { S row-major (reduced camera matrix, J^T J form),
 * g (S * dpose = g), cost = 0.5*sum|r/sigma|^2 (+ point priors), number of valid factors }.
 * `out` is overwritten.  lambda ("damping", the same convention in every BA entry point below): 0 = Gauss-Newton;
 * lambda > 0 = Marquardt scaling, lambda*diag(Hll) on the landmark blocks before elimination and lambda*diag(S) in the
 * solve; lambda < 0 = Levenberg damping |lambda|*I on both, which is what GTSAM 3.2.1's default LevenbergMarquardtParams
 * (diagonalDamping = false, bundle_adjust.cpp:323) adds to every variable.  The result is bitwise reproducible.
 * All pointers are DEVICE pointers, 16-byte aligned; asynchronous on `stream`.
 * ------------------------------------------------------------------------------------- */
int mqs_ba_linearize_dev(const double *poses, const double *calib, const double *sigma, int C,
                         const double *points, const double *obs, const uint8_t *mask,
                         const double *prior_w, const double *prior_xyz, int64_t N, double lambda,
                         double *out, void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_ba_workspace_bytes(int C, int64_t N);

/* Reduced camera system: adds PriorFactor<Pose3> terms (prior_mask [C] uint8 or NULL,
 * prior_poses [C][12], prior_sigmas [C][6] = rot x3, trans x3; e ~ (Log(R0^T R), R0^T (t - t0)),
 * Jacobian ~ I) and lambda*diag(S) damping to lin = linearize's `out`, solves S dpose = g by
 * Cholesky, writes dpose [6C], the retracted poses R <- R Exp(omega), t <- t + R v into
 * poses_out [C][12] (NULL = skip) and info[2] = { pose-prior cost, 1.0 if S was not positive
 * definite } (NULL = skip). */
int mqs_ba_solve_dev(const double *lin, int C, const double *poses, const double *prior_poses,
                     const double *prior_sigmas, const uint8_t *prior_mask, double lambda,
                     double *dpose, double *poses_out, double *info, void *stream);

/* points_out = points + dpoint, dpoint = Hll^-1 (gl - Hpl^T dpose) at the SAME linearisation
 * point as the preceding linearize call (recomputed, not stored).  dpose [6C] device ptr.
 * points_out may alias points. */
int mqs_ba_backsub_dev(const double *poses, const double *calib, const double *sigma, int C,
                       const double *points, const double *obs, const uint8_t *mask,
                       const double *prior_w, const double *prior_xyz, int64_t N, double lambda,
                       const double *dpose, double *points_out, void *stream);
/* The tail of an iteration in ONE launch (C <= 4; two launches beyond): reduced solve + pose retraction (as mqs_ba_solve_dev)
 * and the landmark back-substitution with the dpose it has just produced (as mqs_ba_backsub_dev), bit-identical to the two
 * calls.  Every workgroup solves the 6C x 6C system itself, so no launch boundary separates the solve from its consumers. */
int mqs_ba_solve_backsub_dev(const double *lin, int C, const double *poses, const double *calib, const double *sigma,
                             const double *points, const double *obs, const uint8_t *mask, const double *prior_w,
                             const double *prior_xyz, int64_t N, double lambda, const double *prior_poses,
                             const double *prior_sigmas, const uint8_t *prior_mask, double *dpose, double *poses_out,
                             double *info, double *points_out, void *stream);

/* cost only: out[2] = { 0.5*sum|r/sigma|^2 (+ point priors), valid-factor count }.
 * workspace: at least 64 KiB. */
int mqs_ba_cost_dev(const double *poses, const double *calib, const double *sigma, int C,
                    const double *points, const double *obs, const uint8_t *mask,
                    const double *prior_w, const double *prior_xyz, int64_t N, double *out,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU transport: one process per GPU, one RCCL communicator per mqs_ctx.  The only collective
 * of the hot path is the sum all-reduce of linearize's `out` ((6C)^2 + 6C + 2 doubles) once per optimiser iteration --
 * the GTSAM counterpart has none: bundle_adjust.cpp:323-324 optimises one graph in one process.
 * RCCL is bound at run time (the library loads without it).  Rank 0 calls mqs_comm_unique_id and the host program hands
 * the 128 bytes to every rank (torch.distributed store, MPI, a file); every rank then calls mqs_comm_init_rank
 * (collective).  mqs_comm_world_size: ranks of the context's communicator, 0 without one.
 * ------------------------------------------------------------------------------------- */
int mqs_comm_unique_id(uint8_t *id128);
int mqs_comm_init_rank(mqs_ctx *ctx, const uint8_t *id128, int rank, int world);
int mqs_comm_world_size(const mqs_ctx *ctx);
int mqs_comm_destroy(mqs_ctx *ctx);
/* in-place sum over the ranks of buf[n] (device pointer), asynchronous on `stream`.  With a peer transport (below) and
 * n <= 5120 the sum is the peers' rows added in rank order by one small kernel; otherwise ncclAllReduce. */
int mqs_comm_all_reduce_sum_f64_dev(mqs_ctx *ctx, double *buf, int64_t n, void *stream);

/* Peer transport: the same collective as plain stores over xGMI into receive buffers the ranks map from each other
 * (hipIpc), for a message this small (4.8 KB) a launch-free alternative to RCCL: inside mqs_ba_gn_iteration_dev the
 * lineariser's finalize kernel stores the rank's row into every peer and the fused solve/back-substitution kernel waits for
 * the rows and adds them in rank order (bit-identical on every rank).  Set-up mirrors the unique id: every rank calls
 * mqs_comm_peer_export (allocates its receive buffer, returns an opaque MQS_PEER_HANDLE_BYTES blob), the host program
 * gathers the blobs of all ranks in rank order, every rank calls mqs_comm_peer_open.  Independent of the RCCL communicator:
 * a context may hold either or both (RCCL then serves buffers that do not fit a row).
 * mqs_comm_peer_state: 0 none, 1 open (a peer shares this GPU: consumers wait in a kernel of their own), 2 open (fused wait).
 * mqs_comm_peer_timed_out: *timed_out = 1 when a consumer gave up waiting for a row (2 s); synchronises `stream`. */
#define MQS_PEER_MAX_WORLD 8
#define MQS_PEER_HANDLE_BYTES 128
int mqs_comm_peer_export(mqs_ctx *ctx, int rank, int world, uint8_t *handle);
int mqs_comm_peer_open(mqs_ctx *ctx, const uint8_t *handles);
int mqs_comm_peer_close(mqs_ctx *ctx);
int mqs_comm_peer_state(const mqs_ctx *ctx);
int mqs_comm_peer_timed_out(mqs_ctx *ctx, void *stream, int *timed_out);

/* ---------------------------------------------------------------------------------------
 * One optimiser iteration behind one call (the loop body of LevenbergMarquardtOptimizer::optimize,
 * bundle_adjust.cpp:323-324, for the graph of :268-298 with every landmark visible in every camera).
 * A problem binds the caller's DEVICE buffers (nothing is copied or owned): two pose buffers [C][12] and two landmark
 * buffers [N][3] that alternate as current / next estimate, the constant inputs of mqs_ba_linearize_dev /
 * mqs_ba_solve_dev, lin [(6C)^2 + 6C + 2], dpose [6C], info [2] (may be NULL) and a workspace of
 * mqs_ba_workspace_bytes(C, N).  ctx: the context whose communicator sums `lin` over the ranks (NULL or a context
 * without communicator: single GPU).  The estimate starts in poses_a / points_a (current = 0).
 *   mqs_ba_gn_iteration_dev   linearise + Schur -> [all-reduce] -> solve + retract -> back-substitute, all enqueued on
 *                             `stream` without a host round trip; the new estimate becomes current.
 *   mqs_ba_gn_begin_dev / mqs_ba_gn_finish_dev   the same in two halves for a caller that sums `lin` itself between
 *                             them (another transport) or decides acceptance (accept = 0: the step stays a trial in the
 *                             other buffers -- Levenberg-Marquardt).
 *   mqs_ba_gn_iterations_dev  `iters` iterations back to back.
 *   mqs_ba_problem_status     the iteration's kernels hand data over between workgroups (the finalize inside the tail) and
 *                             between GPUs (peer transport) behind flags, with BOUNDED waits (2 s): a GPU must never hang on a
 *                             rank that does not arrive.  A wait that gives up is an error, not a wrong answer: the waiting
 *                             workgroups read nothing of what they waited for and publish nothing (no poses, no landmarks;
 *                             info[1] = 2), and a sticky status word of the problem is raised.  mqs_ba_problem_status
 *                             synchronises `stream` and returns MQS_OK or MQS_E_TIMEOUT; mqs_ba_gn_iteration(s)_dev return
 *                             MQS_E_TIMEOUT themselves (without synchronising) once an earlier iteration's failure has become
 *                             visible.  Call it wherever the host consumes the estimate.
 * ------------------------------------------------------------------------------------- */
typedef struct mqs_ba_problem mqs_ba_problem;
int mqs_ba_problem_create(mqs_ctx *ctx, int C, int64_t N, double *poses_a, double *poses_b, const double *calib,
                          const double *sigma, double *points_a, double *points_b, const double *obs, const uint8_t *mask,
                          const double *prior_w, const double *prior_xyz, const double *prior_poses,
                          const double *prior_sigmas, const uint8_t *prior_mask, double *lin, double *dpose, double *info,
                          void *workspace, int64_t workspace_bytes, mqs_ba_problem **out);
void mqs_ba_problem_destroy(mqs_ba_problem *p);
int mqs_ba_problem_current(const mqs_ba_problem *p);              /* 0 / 1: which buffer pair holds the estimate */
int mqs_ba_problem_set_current(mqs_ba_problem *p, int which);
int mqs_ba_gn_begin_dev(mqs_ba_problem *p, double lambda, void *stream);
int mqs_ba_gn_finish_dev(mqs_ba_problem *p, double lambda, int accept, void *stream);
int mqs_ba_gn_iteration_dev(mqs_ba_problem *p, double lambda, void *stream);
int mqs_ba_gn_iterations_dev(mqs_ba_problem *p, int iters, double lambda, void *stream);
int mqs_ba_problem_status(mqs_ba_problem *p, void *stream);
/* Test hook: the finalizer piece (0 .. 47) of the fused tail that does not raise its flag in launches the CALLING THREAD issues from
 * now on, so that tests can reach the time-out path (-1 = none, the default; other threads' problems never see it).  Never set in
 * production code. */
int mqs_debug_ba_withhold_flag(int piece);

/* Timing helper used by bench.py: average duration (ms) of `reps` back-to-back launches of ONE kernel of the iteration,
 * hipEvents on `stream` (what: 0 = the lineariser kernel alone, 1 = its finalize kernel alone, 2 = solve + retract,
 * 3 = back-substitution).  Buffers as for the calls above; lin must hold a linearisation for what = 2. */
int mqs_ba_time_dev(int what, const double *poses, const double *calib, const double *sigma, int C, const double *points,
                    const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                    double lambda, double *lin, double *dpose, double *poses_out, double *points_out, void *workspace,
                    int64_t workspace_bytes, int reps, void *stream, float *avg_ms);

/* Host-pointer convenience wrappers (copy in, run, copy out; synchronous). */
int mqs_ba_linearize(mqs_ctx *ctx, const double *poses, const double *calib, const double *sigma, int C,
                     const double *points, const double *obs, const uint8_t *mask,
                     const double *prior_w, const double *prior_xyz, int64_t N, double lambda,
                     double *out);
int mqs_ba_backsub(mqs_ctx *ctx, const double *poses, const double *calib, const double *sigma, int C,
                   const double *points, const double *obs, const uint8_t *mask,
                   const double *prior_w, const double *prior_xyz, int64_t N, double lambda,
                   const double *dpose, double *points_out);

/* ---------------------------------------------------------------------------------------
 * General sparse-visibility bundle adjustment: P poses (pose_cam [P] = camera id of each pose),
 * N landmarks, M observations in CSR by landmark (obs_ptr [N+1] int64, obs_pose [M] int32 sorted by
 * pose within a landmark, obs_uv [M][2]), and the list of Q observation pairs (pair_a <= pair_b,
 * global observation indices of one landmark, including a == b) that produce reduced-system blocks.
 * This is the graph bundle_adjust.cpp:245-298 builds from the reference's file set (IO.hpp:366-406).
 * All pointers are device pointers.
 *   linearize: S [(6P)^2] row-major and g [6P] are overwritten with the reduced camera system
 *     (J^T J form incl. PriorFactor<Pose3> terms of the n_pose_prior listed poses, bundle_adjust.cpp:273);
 *     info[4] = {0.5*sum|r/sigma|^2 + point priors, valid-factor count, pose-prior cost, 0 (grouped entry point: see there)}.
 *   solve: in place blocked Cholesky of (S + lambda*diag S), x (= g on entry) -> dpose; poses_out =
 *     retract(poses, dpose) when not NULL; bad[0] = 1 when S was not positive definite.  Input contract: the LOWER
 *     triangle of S (with the diagonal), as for any Cholesky routine; the strict upper triangle is overwritten (the
 *     solve mirrors lower -> upper inside the band first: its factor kernels read panel input from the mirror image).
 *     On return the lower triangle holds L, the upper triangle work values.
 *   backsub / cost as in the dense API.  mqs_sba_linearize_dev sums with fp64 atomics (not bitwise reproducible);
 *   the grouped form below does not.
 * ------------------------------------------------------------------------------------- */
int64_t mqs_sba_workspace_bytes(int64_t P, int64_t N, int64_t M);
int mqs_sba_linearize_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                          const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                          const int32_t *obs_pose, const double *obs_uv, int64_t M, const int64_t *pair_a,
                          const int64_t *pair_b, int64_t Q, const double *prior_w, const double *prior_xyz,
                          const int32_t *pose_prior_idx, const double *pose_prior_poses,
                          const double *pose_prior_sigmas, int n_pose_prior, double lambda, double *S,
                          double *g, double *info, void *workspace, int64_t workspace_bytes, void *stream);
/* The same linearisation with the pair list SORTED by (pose of pair_a, pose of pair_b) and cut into G groups of equal key
 * (group_ptr [G + 1] int64 offsets into the pair list): one wavefront per group sums its blocks in registers and writes the
 * 6 x 6 block once -- no atomics, bitwise reproducible, and about 15x faster on the pair stage.  It writes the block's
 * mirror image too (no full-matrix mirror pass): this relies on a CANONICAL grouping -- pose(pair_a) <= pose(pair_b) inside
 * every group, one group per pose pair, groups in increasing (pose_a, pose_b) order.  The grouping is checked on the device
 * (no host round trip): info[3] = number of groups that violate it; S must not be used when it is not 0. */
int mqs_sba_linearize_grouped_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                                  const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                                  const int32_t *obs_pose, const double *obs_uv, int64_t M, const int64_t *pair_a,
                                  const int64_t *pair_b, int64_t Q, const int64_t *group_ptr, int64_t G,
                                  const double *prior_w, const double *prior_xyz, const int32_t *pose_prior_idx,
                                  const double *pose_prior_poses, const double *pose_prior_sigmas, int n_pose_prior,
                                  double lambda, double *S, double *g, double *info, void *workspace,
                                  int64_t workspace_bytes, void *stream);
/* The grouped pair list itself, built on the device (csrc/pair_group.hip): every (a, b), a <= b, of observation indices inside
 * a landmark -- observations sorted by pose inside every landmark, so pose(a) <= pose(b) -- sorted by (pose of a, pose of b),
 * stably (a group's pairs stay in landmark order), with the group offsets.  pair_off [N + 1] = exclusive prefix sums of
 * k (k + 1) / 2 over the landmarks' observation counts k (pair_off[N] = Q).  All pointers are device pointers; n_groups[0]
 * (device) receives the number of groups G, group_ptr [G + 1 <= group_cap] their offsets (group_ptr[G] = Q);
 * G <= min(Q, P (P + 1) / 2).  Generation, a stable least-significant-digit radix sort and an ordered compaction: ~0.3 ms for
 * the 2.0 M pairs of an 881-pose sequence, where numpy took 0.2 s. */
int64_t mqs_sba_group_pairs_workspace_bytes(int64_t Q);
int mqs_sba_group_pairs_dev(const int64_t *obs_ptr, const int32_t *obs_pose, int64_t N, const int64_t *pair_off, int64_t Q, int64_t P,
                            int64_t *pair_a, int64_t *pair_b, int64_t *group_ptr, int64_t group_cap, int64_t *n_groups,
                            void *workspace, int64_t workspace_bytes, void *stream);
/* The observations sorted by pose index inside every landmark (stable), on the device: the order mqs_sba_group_pairs_dev and
 * the sparse linearisers assume.  obs_ptr [N + 1] CSR by landmark; obs_pose_in [M] / obs_uv_in [M][2] in any order inside a
 * landmark; obs_pose_out / obs_uv_out distinct from the inputs; order_out [M] int32 (may be NULL): the input index of every
 * output observation.  Replaces the host-side lexsort of the set-up (the recorder of slam2.py:743-865 and IO.hpp:366-406 hand
 * observations over in step order, not pose order). */
int64_t mqs_sba_sort_observations_workspace_bytes(int64_t M);
int mqs_sba_sort_observations_dev(const int64_t *obs_ptr, const int32_t *obs_pose_in, const double *obs_uv_in, int64_t N, int64_t M, int64_t P,
                                  int32_t *obs_pose_out, double *obs_uv_out, int32_t *order_out, void *workspace, int64_t workspace_bytes,
                                  void *stream);
int mqs_sba_solve_dev(double *S, double *x, int64_t P, double lambda, const double *poses, double *poses_out,
                      int *bad, void *stream);
/* The same solve for a reduced camera system known to be banded: S[i][j] == 0 for |i - j| > half_bandwidth (a
 * sequence in which a landmark is seen by poses at most d apart, and odometry links at most d apart:
 * half_bandwidth = 6 (d + 1) - 1).  The Cholesky factor has no fill outside the band; when the band is narrower than
 * a third of the matrix the factorisation and the triangular solves stay inside it. */
int mqs_sba_solve_banded_dev(double *S, double *x, int64_t P, int64_t half_bandwidth, double lambda, const double *poses,
                             double *poses_out, int *bad, void *stream);
/* How the banded solve above orders its work for a system of n6 = 6P unknowns (no GPU involved; host only).  Long narrow
 * bands are cut into 2^d independent chunks separated by ceil(half_bandwidth / 32)-block separators (nested dissection;
 * csrc/chol_nd.hip), `parts` = 0 lets the library choose as the solve does (environment MQS_SBA_PARTS overrides there),
 * 1 = no cut.  The plan comes back as int32: header[8] = {stages, descriptors, lazy tiles, contributions, lazy vectors,
 * columns, parts, 0}; per stage 8 ints {fronts, steps, descriptor offset, lazy-tile offset, lazy tiles, lazy-vector offset,
 * lazy vectors, longest front in blocks}; descriptors (32 ints: pivot block | -1, structure size, own-front blocks, tiles,
 * structure[28]); lazy tiles (8 ints: x1, x2, first contribution, count, factor flag, first partial tile, partial tiles, 0); contributions; lazy vectors
 * (4 ints: block, first column, count, 0); columns.  Returns the ints needed (0 = the solve takes the natural order for this
 * shape) and fills `out` when `cap` is large enough.  tests/test_chol_plan.py replays a plan with numpy. */
int64_t mqs_sba_solve_plan_dump(int64_t n6, int64_t half_bandwidth, int parts, int32_t *out, int64_t cap);
/* Plans (and their device scratch) are cached per (device, stream, n6, half_bandwidth) in a small LRU: a session whose
 * pose count grows with every keyframe does not accumulate them.  Returns how many are alive (diagnostic; <= 8). */
int mqs_sba_solve_plan_cache_size(void);
int mqs_sba_backsub_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                        const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                        const int32_t *obs_pose, const double *obs_uv, int64_t M, const double *prior_w,
                        const double *prior_xyz, double lambda, const double *dpose, double *points_out,
                        void *workspace, int64_t workspace_bytes, void *stream);
int mqs_sba_cost_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                     const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                     const int32_t *obs_pose, const double *obs_uv, int64_t M, const double *prior_w,
                     const double *prior_xyz, double *out, void *workspace, int64_t workspace_bytes,
                     void *stream);

/* Per landmark the largest reprojection residual of its observations in PIXELS at the estimate (poses, points):
 * worst [N] (0 for a landmark without observations, +inf when one of its observations lies behind its camera); min_depth [N]
 * (may be NULL): its smallest depth along the optical axes of the cameras that see it (+inf without observations) -- the outlier
 * screen either side of an adjustment (slam_device.py; the reference's tool has none: bundle_adjust.cpp:289-298 are plain
 * least-squares factors over a finished recording). */
int mqs_sba_worst_residual_dev(const double *poses, const int32_t *pose_cam, int64_t P, const double *calib,
                               const double *sigma, const double *points, int64_t N, const int64_t *obs_ptr,
                               const int32_t *obs_pose, const double *obs_uv, int64_t M, double *worst, double *min_depth,
                               void *workspace, int64_t workspace_bytes, void *stream);

/* Odometry: BetweenFactor<Pose3> (bundle_adjust.cpp:301-309, `useOdometry`), GTSAM 3.2.1 conventions: error
 * measured.localCoordinates(T_from^-1 T_to) = (Log(Rm^T Rh), Rm^T (th - tm)), Jacobians of `between`
 * (-Ad(h^-1), I), whitened by odo_sigmas [n_odo][6] (rotation 3, translation 3).  odo_meas [n_odo][12] =
 * R row-major + t.  Call after mqs_sba_linearize_dev: adds J^T J to S (both triangles), -J^T r to g and the
 * factors' cost to cost[0].  S == g == NULL: cost only (LM trial evaluation). */
int mqs_sba_between_dev(const double *poses, int64_t P, const int32_t *odo_from, const int32_t *odo_to,
                        const double *odo_meas, const double *odo_sigmas, int64_t n_odo, double *S, double *g,
                        double *cost, void *stream);

/* Levenberg-Marquardt over the sparse problem in ONE call: what `LevenbergMarquardtOptimizer(graph, initial).optimize()` is to
 * the reference's tool (bundle_adjust.cpp:323-324), with GTSAM 3.2.1's default schedule (LevenbergMarquardtParams: lambdaInitial
 * 1e-5, lambdaFactor 10, lambdaUpperBound 1e5, maxIterations 100, absoluteErrorTol / relativeErrorTol 1e-5; a failed trial
 * multiplies lambda and is repeated from the same linearisation point).  The driver issues the entry points above
 * (linearize_grouped, between, solve_banded, backsub, cost) on `stream` and synchronises ONCE per trial (the trial's total
 * cost and the factorisation's verdict come back together).  All pointers are device pointers of the layouts documented at
 * those entry points; `poses` / `points` hold the initial estimate on entry and the adjusted one on return (`poses_new`,
 * `points_new`, `S` [(6P)^2], `g` [6P] and `workspace` are scratch).  pair_a / pair_b / group_ptr: mqs_sba_group_pairs_dev's
 * lists (G > 0 required when Q > 0 for the atomics-free pair stage; G == 0 selects the atomic one).  Optional parts:
 * prior_w / prior_xyz (NULL: none), n_pose_prior == 0, n_odo == 0.
 * cost_history[0] = cost of the initial estimate, then one entry per ACCEPTED iteration; *n_history entries written
 * (at most history_cap; max_iterations + 1 suffices).  damping: MQS_SBA_DAMPING_GTSAM adds lambda * I (GTSAM's default,
 * diagonalDamping = false), MQS_SBA_DAMPING_MARQUARDT scales the diagonals by (1 + lambda). */
#define MQS_SBA_DAMPING_GTSAM     0
#define MQS_SBA_DAMPING_MARQUARDT 1
typedef struct mqs_sba_problem_dev {
    int64_t P, N, M, Q, G;
    double *poses, *poses_new;                    /* [P][12] camera-to-world R row-major + centre */
    double *points, *points_new;                  /* [N][3] */
    const int32_t *pose_cam;                      /* [P] camera (calibration) of every pose */
    const double *calib, *sigma;                  /* [n_cams][9], [n_cams] */
    const int64_t *obs_ptr;                       /* [N + 1] */
    const int32_t *obs_pose;                      /* [M] sorted by pose inside a landmark (mqs_sba_sort_observations_dev) */
    const double *obs_uv;                         /* [M][2] */
    const int64_t *pair_a, *pair_b, *group_ptr;   /* [Q], [Q], [G + 1] */
    const double *prior_w, *prior_xyz;            /* [N], [N][3] or NULL */
    const int32_t *pose_prior_idx;                /* [n_pose_prior] */
    const double *pose_prior_poses, *pose_prior_sigmas;   /* [n_pose_prior][12], [n_pose_prior][6] */
    const int32_t *odo_from, *odo_to;             /* [n_odo] */
    const double *odo_meas, *odo_sigmas;          /* [n_odo][12], [n_odo][6] */
    int64_t n_odo;
    int64_t half_bandwidth;                       /* of the reduced camera system (6 (dmax + 1) - 1; >= 6P: dense) */
    double *S, *g;
    void *workspace;
    int64_t workspace_bytes;                      /* >= mqs_sba_lm_workspace_bytes(P, N, M) */
    int32_t n_pose_prior, reserved;
} mqs_sba_problem_dev;
typedef struct mqs_sba_lm_params {
    double lambda_initial, lambda_factor, lambda_upper, abs_tol, rel_tol;
    int32_t max_iterations, damping;
} mqs_sba_lm_params;
int64_t mqs_sba_lm_workspace_bytes(int64_t P, int64_t N, int64_t M);
int mqs_sba_optimize_lm_dev(const mqs_sba_problem_dev *problem, const mqs_sba_lm_params *lm, double *cost_history,
                            int32_t history_cap, int32_t *n_history, void *stream);

/* ---------------------------------------------------------------------------------------
 * Camera model either side of triangulation (the published OpenCV 2.4 pinhole + distortion model):
 * intr[9] = fx, fy, cx, cy, k1, k2, p1, p2, k3.
 *   mqs_undistort_points: pixels [N][2] -> normalised undistorted coordinates [N][2]
 *       (cv2.undistortPoints(pts, K, dist) as called at slam2.py:551-552 and
 *        triangulation_comparison.py:173; 5 fixed-point iterations like OpenCV 2.4)
 *   mqs_project_points: world points [N][3] through P [3][4] (world -> camera) and the distortion
 *       model to pixels uv_out [N][2] (NULL = skip), depth_out [N] (NULL = skip) and, when imgp [N][2]
 *       and sqerr_out are given, sqerr_out[0] = sum |uv - imgp|^2 -- calibration_tools.py:116-124
 *       reprojection_error = sqrt(sqerr / N) (cv2.projectPoints with P = [Rodrigues(rvec) | tvec]).
 * ------------------------------------------------------------------------------------- */
int mqs_undistort_points(mqs_ctx *ctx, const double *pixels, const double *intr, int64_t N, double *out);
int mqs_undistort_points_dev(const double *pixels, const double *intr, int64_t N, double *out, void *stream);
int mqs_project_points(mqs_ctx *ctx, const double *points, const double *P, const double *intr,
                       const double *imgp, int64_t N, double *uv_out, double *depth_out, double *sqerr_out);
int mqs_project_points_dev(const double *points, const double *P, const double *intr, const double *imgp,
                           int64_t N, double *uv_out, double *depth_out, double *sqerr_out,
                           void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_project_workspace_bytes(void);

/* ---------------------------------------------------------------------------------------
 * Camera pose from 3D-2D correspondences: the pose step of the per-frame loop.
 * Poses are P = [R | t] (3x4 row-major, world -> camera: OpenCV's [Rodrigues(rvec) | tvec]);
 * objp [N][3] world points, imgp [N][2] pixel observations, intr as above (fx fy cx cy k1 k2 p1 p2 k3).
 *
 *   mqs_solve_pnp: replaces cv2.solvePnP(objp, imgp, K, dist[, rvec, tvec, useExtrinsicGuess=True])
 *       (slam2.py:489-490, 576-577, 1156): Levenberg-Marquardt on the pixel reprojection error, started
 *       from `pose` when use_guess != 0, else from a direct linear transform of the undistorted points
 *       (N >= 6).  pose is updated in place.  info (may be NULL): [sum of squared residuals in px^2,
 *       LM iterations, N, flags (bit 0: converged, bit 1: DLT start failed)].
 *   mqs_pnp_refine_dev: B independent problems in one launch (one wavefront each).  Problem b uses the
 *       correspondences idx[ptr[b] .. ptr[b+1]) (idx == NULL: the points ptr[b] .. ptr[b+1] themselves;
 *       ptr == NULL with B == 1: all N points).  poses_in / poses_out [B][12], info [B][4] (may be NULL).
 *       A problem with fewer than 3 correspondences is not solved: poses_out[b] = poses_in[b] (the identity pose
 *       without poses_in) and info[b] = {0, 0, 0, 2} (flags bit 1) -- what the device-resident loop relies on for
 *       a rejected frame; mqs_solve_pnp (host entry) keeps refusing N < 3 with MQS_E_ARG.
 *   mqs_solve_pnp_ransac: replaces cv2.solvePnPRansac(objp, imgp, K, dist, minInliersCount=..,
 *       reprojectionError=..) (slam2.py:453-454).  The caller draws the minimal samples
 *       (samples [B][sample_size] int32 point indices, sample_size >= 6) so that runs are repeatable; all B
 *       hypotheses are evaluated at once (DLT of the sample, `sample_iters` LM iterations on it, inlier =
 *       in front of the camera and reprojection error <= reproj_error), the one with the most inliers
 *       (lowest index on ties) is refined on its inliers like OpenCV's final solvePnP.  sel[0] = chosen
 *       hypothesis (-1: none valid), sel[1] = number of inliers; mask [N] (may be NULL) marks them.
 *       OpenCV's early exit at minInliersCount only shortens its serial loop; its accept / reject
 *       decision on the returned inlier count stays with the caller (slam2.py:461-468).
 * ------------------------------------------------------------------------------------- */
/* One keyframe step of the per-frame loop (slam2.py handle_new_frame :453-490, 541-590) in ONE launch: pose of the frame from
 * its n_old tracked landmarks (start P_prev), two-view iterative-LS triangulation of the n_new not-yet-triangulated pixel
 * tracks p0 (base keyframe, pose P0) / p1 (this frame) after cv2.undistortPoints, the points with status 1 cast to float32
 * and added to the pose problem, the refined pose, and the re-triangulation of those points with it.  Replaces two
 * cv2.solvePnP, four cv2.undistortPoints and two iterative_LS_triangulation calls (:489-490, 551-555, 576-577, 582-584) with
 * the same results.  poses [2][12] = first and refined pose ([R | t], world -> camera; equal when n_new == 0);
 * x [n_new][3], status [n_new] = second-pass status (keep >= 0, :589) or -128 for a point the first pass did not keep (x = NaN);
 * info [8] = {sqerr, iterations, points, converged} of the two pose solves (may be NULL).  n_new == 0: pose only. */
int mqs_keyframe_step(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t n_old, const double *p0, const double *p1,
                      int64_t n_new, const double *intr, const double *P_prev, const double *P0, double tolerance, int max_iter,
                      double eps, double *poses, double *x, int32_t *status, double *info);
int mqs_solve_pnp(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t N, const double *intr, double *pose,
                  int use_guess, int max_iter, double eps, double *info);
int mqs_pnp_refine_dev(const double *objp, const double *imgp, int64_t N, const int32_t *idx, const int32_t *ptr, int B,
                       const double *intr, const double *poses_in, int use_guess, int max_iter, double eps,
                       double *poses_out, double *info, void *stream);
int mqs_solve_pnp_ransac(mqs_ctx *ctx, const double *objp, const double *imgp, int64_t N, const double *intr,
                         const int32_t *samples, int B, int sample_size, double reproj_error, int sample_iters,
                         int max_iter, double eps, double *pose, int32_t *sel, uint8_t *mask, double *info);
int mqs_pnp_ransac_dev(const double *objp, const double *imgp, int64_t N, const double *intr, const int32_t *samples,
                       int B, int sample_size, double reproj_error, int sample_iters, int max_iter, double eps,
                       double *pose_out, int32_t *sel_out, uint8_t *mask, double *info, void *workspace,
                       int64_t workspace_bytes, void *stream);
int64_t mqs_pnp_workspace_bytes(int64_t N, int B);

/* ---------------------------------------------------------------------------------------
 * Image front-end of the per-frame loop.  Images are 8-bit, row-major, W x H, dense.
 * OpenCV 2.4's published methods in float32 with a fixed operation order (oracle/features_np.py); parity with
 * OpenCV itself is unpinned (the reference holds no images).
 *
 *   mqs_good_features_to_track: replaces cv2.goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance, None,
 *       mask) (cv2_helpers.py:34-37; slam2.py:665, 1174): Shi-Tomasi minimum-eigenvalue response (Sobel 3, block 3),
 *       threshold = qualityLevel x the largest response among the UNMASKED pixels (featureselect.cpp: minMaxLoc(eig, ..., mask)),
 *       candidates = thresholded 3x3 maxima under `mask` (may be NULL) off the 1-pixel border, ordered by response
 *       (ties: row-major position), greedy minimum-distance selection.  out_xy [out_capacity][2] float32 (x, y),
 *       out_n = number written (<= max_corners when max_corners > 0).
 *   mqs_calc_optical_flow_pyr_lk: replaces cv2.calcOpticalFlowPyrLK(prev, next, prevPts) (slam2.py:381; defaults
 *       21 x 21, maxLevel 3, 30 iterations, eps 0.01, minEigThreshold 1e-4).  next_pts [n][2], status [n] (1 = tracked),
 *       err [n] (mean absolute window difference at level 0).  A window may leave the image by up to its own size (lkpyramid.cpp's
 *       test); it is then read as OpenCV reads its border-extended pyramid: intensities reflected (BORDER_REFLECT_101), the
 *       derivative image zero outside (BORDER_CONSTANT).
 * ------------------------------------------------------------------------------------- */
int mqs_good_features_to_track(mqs_ctx *ctx, const uint8_t *img, int W, int H, int max_corners, double quality_level,
                               double min_distance, const uint8_t *mask, float *out_xy, int out_capacity, int32_t *out_n);
int mqs_good_features_to_track_dev(const uint8_t *img, int W, int H, int max_corners, double quality_level,
                                   double min_distance, const uint8_t *mask, float *out_xy, int out_capacity, int32_t *out_n,
                                   void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_gftt_workspace_bytes(int W, int H);
int mqs_calc_optical_flow_pyr_lk(mqs_ctx *ctx, const uint8_t *prev_img, const uint8_t *next_img, int W, int H,
                                 const float *prev_pts, int n, int win_w, int win_h, int max_level, int max_iter, double eps,
                                 double min_eig_threshold, float *next_pts, uint8_t *status, float *err);
int mqs_calc_optical_flow_pyr_lk_dev(const uint8_t *prev_img, const uint8_t *next_img, int W, int H, const float *prev_pts,
                                     int n, int win_w, int win_h, int max_level, int max_iter, double eps,
                                     double min_eig_threshold, float *next_pts, uint8_t *status, float *err, void *workspace,
                                     int64_t workspace_bytes, void *stream);
int64_t mqs_lk_workspace_bytes(int W, int H, int max_level);
/*   mqs_fast_detect: replaces cv2.FastFeatureDetector().detect(img) (slam.py:34, 62; FAST-9/16, default threshold 10,
 *       non-maximum suppression on the OpenCV 2.4 corner score).  Corners in row-major scan order: out_xy [capacity][2],
 *       out_score [capacity] (may be NULL); out_n = number FOUND (may exceed out_capacity: only the first
 *       out_capacity are written).  Integer arithmetic: exact. */
int mqs_fast_detect(mqs_ctx *ctx, const uint8_t *img, int W, int H, int threshold, int nonmax, float *out_xy,
                    int32_t *out_score, int out_capacity, int32_t *out_n);
int mqs_fast_detect_dev(const uint8_t *img, int W, int H, int threshold, int nonmax, float *out_xy, int32_t *out_score,
                        int out_capacity, int32_t *out_n, void *workspace, int64_t workspace_bytes, void *stream);
int64_t mqs_fast_workspace_bytes(int W, int H);

/* ---------------------------------------------------------------------------------------
 * The per-frame loop resident on the device (BASELINE configs[4]; slam2.py handle_new_frame :360-695, main :1136-1253):
 * the live tracks, their base-keyframe positions, the map (float32 values, :19) and the poses stay in device memory; a frame
 * is ONE call that enqueues pyramidal LK of the live tracks, the status / error filter and the two early gates, all RANSAC
 * hypotheses + selection + solvePnP on the inliers, the outlier-ratio and reprojection gates, the commit of the kept tracks
 * and the keyframe test, waits for a 320-byte result block, and on a keyframe enqueues -- without waiting -- the keyframe
 * step (triangulate the free tracks, refine the pose, re-triangulate), the map update, the coverage mask, goodFeaturesToTrack
 * and the top-up of the tracks.  Thresholds default to slam2.py:1070-1098.  Images are DEVICE pointers (8-bit, W x H, dense)
 * that must stay valid until the next call returns.  csrc/slam_frame.hip.
 * Streams: the handle works on a private non-blocking stream and takes no event from the caller, so the images must be
 * COMPLETE in device memory when mqs_slam_start / mqs_slam_track is called -- an upload still in flight on another stream
 * (torch's current stream, say) is not ordered against the library's reads: synchronise that stream (or the device) first.
 * The handle's streams (this one, and the side stream mqs_slam_set_next brings) are created with the HIGHEST stream priority: their hardware
 * queues then come from a pool no default-priority stream of the process draws from (two streams that share a queue run one behind the other:
 * csrc/slam_frame.hip), and their kernels are dispatched ahead of default-priority work of the same process on the same device.
 * n0 <= max_landmarks (the start-up points are the first entries of the map).
 *   mqs_slam_start   first frame: pose from n0 known 3-D points (HOST float32 objp0 [n0][3], imgp0 [n0][2]), which become
 *                    the first landmarks and tracks; the other tracks from goodFeaturesToTrack.  pose_out [12] host.
 *   mqs_slam_track   result [40] host doubles: [0] decision (0 rejected, 1 frame, 2 keyframe), [1] rejection reason (1 lost
 *                    tracks, 2 < 8 landmark tracks, 3 no RANSAC model, 4 outlier ratio, 5 reprojection error), [2] tracks kept,
 *                    [3] landmark tracks, [4] inliers, [5] / [6] old / new point sets of the keyframe step, [7] lost ratio,
 *                    [8] outlier ratio, [9] reprojection RMS, [10] homography w0 / w2, [11] landmarks, [12..23] pose (3x4
 *                    world -> camera; a keyframe's before refinement); [24..39]: the report of the PREVIOUS call's keyframe
 *                    branch, which runs behind the call that started it -- [24] valid, [25] landmarks added, [26] tracks
 *                    after the top-up, [27] landmarks, [28..39] refined pose.  mqs_slam_flush: that block once the stream
 *                    has drained (after the last frame).
 *   mqs_slam_read_tracks / _read_map   the state, for recorders and tests (synchronise the handle's stream).
 *   mqs_slam_set_thresholds   the gates of slam2.py:1070-1098 and max_homography_points: keyframe_test's random sample of the
 *                    tracks (slam2.py:48) -- 0 (the default after mqs_slam_create): the homography is fitted to ALL kept
 *                    tracks; k >= 4: to a uniformly random k of them, drawn from the handle's counter-based generator
 *                    (the reference's rule is k = max(4, target_keypoints / 4), slam2.py:1088-1089).  The homography itself is
 *                    cv2.findHomography(method = 0) (slam2.py:54): normalised DLT, then -- with more than four pairs -- the
 *                    Levenberg-Marquardt refinement of the transfer error (OpenCV 2.4 fundam.cpp: estimator.refine(M, m, H, 10)).
 * ------------------------------------------------------------------------------------- */
typedef struct mqs_slam mqs_slam;
/* OPTIONAL, off by default (not in slam2.py's flow): at a keyframe, a freshly triangulated point whose reprojection error in the
 * current frame -- under the pose it was triangulated with -- exceeds `max_reproj_error_px` is dropped instead of being handed to the
 * second solvePnP (slam2.py:576-577).  The reference defines the bound for exactly this place (slam2.py:1092:
 * max_2nd_solvePnP_reproj_error = max_solvePnP_reproj_error / 2 = 1 px, "used in 2nd iteration, after 1st pass of triangulation")
 * and never uses it; without it one gross two-view point (status 1 = converged and in front of both cameras, nothing about its
 * residual) throws the keyframe's pose centimetres off.  0 switches the screen off again. */
int mqs_slam_set_second_pass_screen(mqs_slam *s, double max_reproj_error_px);
int mqs_slam_create(int device, int W, int H, const double *intr, int target_keypoints, double coverage_radius,
                    double quality_level, int max_landmarks, uint64_t seed, mqs_slam **out);
void mqs_slam_destroy(mqs_slam *s);
int mqs_slam_set_thresholds(mqs_slam *s, double max_of_error, double max_lost_tracks_ratio, double max_reproj_error,
                            double max_outlier_ratio, double homography_condition_threshold, int max_homography_points);
 /* Bundle adjustment inside the loop (BASELINE configs[4]: detect -> track -> triangulate -> BA per keyframe).  The reference
 * records what its bundle adjuster needs in BundleAdjustmentInfoContainer (slam2.py:519-522, 634-641, 1167-1169) and runs the
 * adjuster as a separate program afterwards; here the record is a flat log in device memory, appended by the frame's own
 * kernels, and the result of an adjustment goes back into the live state:
 *   mqs_slam_log_enable   before mqs_slam_start: capacity in observations.  Every accepted frame appends (landmark, pose index,
 *                         pixel) for every track it keeps -- a free track under its track id, resolved to its landmark once a
 *                         keyframe has triangulated it (a new landmark so brings along its image points of every frame since
 *                         the base keyframe, slam2.py:634-641) -- and every keyframe appends the new landmarks' pixels in the
 *                         base keyframe.  Pose index = rank among the accepted frames.
 *   mqs_slam_read_log     the log so far into host arrays (lm, pose int32 [n]; uv float64 [n][2]); n = entries logged;
 *                         lm = -1 for the observations of tracks that have not (yet) become landmarks.
 *   mqs_slam_write_back   the first n landmarks of the map and the poses of the last accepted frame / the base keyframe
 *                         ([R | t] 3x4 world -> camera, NULL = unchanged) replaced by adjusted values. */
 /*   mqs_slam_reassociate  behind a keyframe (after the mqs_slam_track that reported it): the reference's match_OF_based
 *                         (slam.py:81-127) inside this loop -- BFMatcher.radiusMatch on pixel coordinates
 *                         (mqs_match_knn2_f32_dev), ratio test and one match per keypoint (mqs_match_ratio_unique_dev) between
 *                         the projections of the landmarks that are in the map but not tracked any more and the corners the
 *                         keyframe's top-up has just detected; a matched corner continues its landmark (and the observation
 *                         is logged) instead of starting a new one.  *n_matched: corners re-associated. */
int mqs_slam_reassociate(mqs_slam *s, float max_radius, double max_dist_ratio, int32_t *n_matched);
 /*   mqs_slam_bundle_adjust   the adjustment behind a keyframe as ONE call on the resident state (round 5): what
 *                         `performBundleAdjustment` (bundle_adjust.cpp:190-330, iSAM_version 0: the whole graph, one batch
 *                         LevenbergMarquardtOptimizer::optimize()) does with the reference's recording, done with the log,
 *                         the map and the trajectory where they are.  One persistent launch on the handle's stream builds the
 *                         sparse problem from the log (every accepted frame so far, every landmark seen from at least
 *                         `min_observations` frames; gauge as bundle_adjust.cpp:268-282: a pose prior on the first frame at
 *                         its start-up estimate, point priors on the start-up landmarks at their given positions; odometry
 *                         BetweenFactors keyframe -> keyframe, :301-309), runs GTSAM 3.2.1's default Levenberg-Marquardt
 *                         schedule (<= max_iterations) -- landmark elimination, reduced camera system, Cholesky, back-
 *                         substitution, trial cost, accept / reject all inside the launch, workgroups meeting at grid-wide
 *                         barriers -- screens mistracked landmarks (a landmark whose worst residual exceeds `gross_px`
 *                         before, `outlier_px` after an adjustment, or that lies closer to one of its cameras than
 *                         `min_depth_ratio` x the median depth, sits out from then on; the adjustment is redone from its
 *                         start, <= max_passes; screen_iterations = K > 0: the screen behind an adjustment also looks behind every
 *                         K of its iterations -- a pass that carries a mistracked corner is thrown away after K trials instead
 *                         of after max_iterations; a leg of K iterations that finds none goes on from its estimate with the
 *                         damping back at lambda_initial, like a second optimize() call) and writes the adjusted landmarks (float32 values, slam2.py:19), the poses of
 *                         all accepted frames and the live state's two poses back.  Nothing is copied to the host but the
 *                         report and, if asked for, the adjusted poses.
 *                         add_odometry_edge != 0: the call stands behind a keyframe; the edge (edge_from = the previous base
 *                         keyframe's pose index, edge_to = this keyframe's) is measured from the trajectory as it stands
 *                         (slam2.py:681-687) and kept for this and all later adjustments.
 *                         report [MQS_SLAM_BA_REPORT] doubles: [0] status (0 done; 1 a grid-wide wait gave up -> MQS_E_TIMEOUT;
 *                         2 capacity: a pose with more observations than a list holds, more co-observations than the hit lists
 *                         hold -> MQS_E_CAPACITY; 3 the log overflowed -> MQS_E_ARG), [1] poses of the problem, [2] landmarks,
 *                         [3] landmarks adjusted, [4] observations
 *                         used, [5] passes, [6] landmarks screened out by this call, [7] LM iterations of the last pass,
 *                         [8] cost before, [9] cost after, [10] LM trials (linearise + solve) in all, [11] observations
 *                         that repeat a (landmark, frame) pair and were left out, [12] odometry edges, [13] grid barriers
 *                         passed, [14] first accepted frame whose pose this call rewrote (0 for the plain call), [15] frames from
 *                         there to the last accepted one.  poses_out (host, may be NULL): the adjusted [R | t] world -> camera
 *                         of the accepted frames [report[14], report[14] + min(report[15], poses_cap)).
 *                         More than MQS_SLAM_BA_MAX_POSES accepted frames: MQS_E_ARG -- mqs_slam_bundle_adjust_window selects.
 *                         The launch's workgroups meet at spinning barriers, so all of them have to be resident at once: the
 *                         library sizes the grid from the device (mqs_slam_ba_resident_groups: compute units x workgroups of
 *                         this kernel per unit), an explicit `workgroups` (or MQS_SLAM_BA_GROUPS) beyond that is MQS_E_ARG at
 *                         once, and launches of one process on one device are serialised.
 *   mqs_slam_bundle_adjust_window   the same adjustment over a SELECTION of the accepted frames -- what keeps the cost of an
 *                         adjustment bounded on runs of the reference's lengths (its committed runs: 376 and 881 poses, its tool
 *                         adjusts whatever the recording holds, bundle_adjust.cpp:190-330).  `poses`: n_poses ascending indices
 *                         of accepted frames (<= MQS_SLAM_BA_MAX_POSES; the last one = the last accepted frame; the base
 *                         keyframe of the live tracks among them) become the poses of the problem -- typically the keyframes of
 *                         the run so far and every frame since the K-th keyframe from the end.  Observations at other frames
 *                         are not factors of the problem; they count towards `min_observations`, and with seen_outside_sigma
 *                         > 0 a landmark that has any keeps a prior of that sigma at the value this adjustment found it with.
 *                         Gauge: while frame 0 is selected, as the plain call (prior on pose 0 at its start-up estimate, on the
 *                         start-up landmarks at their given positions); otherwise a prior (`pose_sigmas`) on the first selected
 *                         pose at its current value, and -- second_anchor >= 0 -- on that frame's as well (two poses fix the
 *                         scale of a monocular window).  Odometry edges count when both ends are selected.  carry_unselected:
 *                         every accepted frame behind poses[0] that is not selected keeps its pose RELATIVE to the last
 *                         selected pose in front of it (M_j <- M_j inv(A_before) A_after); 0: it stays where it was. */
#define MQS_SLAM_BA_REPORT 16
#define MQS_SLAM_BA_MAX_POSES 256
typedef struct mqs_slam_ba_params {
    int32_t max_iterations, min_observations, max_passes;
    int32_t add_odometry_edge, edge_from, edge_to;
    int32_t damping;                                  /* MQS_SBA_DAMPING_GTSAM / _MARQUARDT */
    int32_t workgroups;                               /* 0: the library's choice */
    int32_t screen_iterations;                        /* 0: off; K: the residual screen also looks behind every K iterations of an adjustment (see above) */
    int32_t reserved;
    double outlier_px, gross_px, border_margin_px, min_depth_ratio;
    double point_sigma, pixel_sigma;
    double pose_sigmas[6], odometry_sigmas[6];        /* rotation (3) then translation (3), as the reference's noise files */
    double lambda_initial, lambda_factor, lambda_upper, abs_tol, rel_tol;
} mqs_slam_ba_params;
int mqs_slam_bundle_adjust(mqs_slam *s, const mqs_slam_ba_params *params, double *report, double *poses_out, int32_t poses_cap);
typedef struct mqs_slam_ba_window {
    int32_t n_poses;                                  /* 0: every accepted frame (= mqs_slam_bundle_adjust) */
    int32_t second_anchor;                            /* accepted-frame index, or -1 */
    int32_t carry_unselected;
    int32_t reserved;
    const int32_t *poses;                             /* host */
    double seen_outside_sigma;                        /* 0: no such prior */
} mqs_slam_ba_window;
int mqs_slam_bundle_adjust_window(mqs_slam *s, const mqs_slam_ba_params *params, const mqs_slam_ba_window *window, double *report,
                                  double *poses_out, int32_t poses_cap);
/* workgroups of the adjuster's persistent launch the device can hold at once (its grid is never larger) */
int mqs_slam_ba_resident_groups(mqs_slam *s, int32_t *groups);
/* test hook: the NEXT mqs_slam_bundle_adjust(_window) on this handle returns `code` (MQS_E_TIMEOUT or MQS_E_CAPACITY; 0 clears) without
 * launching or writing anything -- what a caller's fall-back path is tested with */
int mqs_debug_slam_ba_fail_next(mqs_slam *s, int code);
/* the landmarks the in-loop adjuster has retired so far (1) / not (0): host uint8 [cap]; *n = landmarks in the map */
int mqs_slam_read_ba_flags(mqs_slam *s, uint8_t *retired, int cap, int32_t *n);
/* the odometry edges the adjuster holds (pose indices, measured relative pose12 [cap][12]): what a caller needs to take the adjustment over
 * beyond MQS_SLAM_BA_MAX_POSES accepted frames; *n = edges held */
int mqs_slam_read_ba_edges(mqs_slam *s, int32_t *from, int32_t *to, double *meas, int cap, int32_t *n);
/* profiling hook: phase stamps of the adjuster's last launch (see csrc/slam_ba.hip); the first call switches them on */
int mqs_debug_slam_ba_stamps(mqs_slam *s, int64_t *out, int cap, int32_t *n);
/* test hook: the adjuster's 32 x 32 diagonal-tile Cholesky factor by itself.  form 0: four wavefronts, a workgroup barrier per pivot; 1: one
 * wavefront, panels of four pivots on the fp64 matrix pipe (round 6's A/B form: measured slower, 7.7 against 6.3 us per tile).  A, out: host, 32 x 32 row-major; out = L in
 * the lower triangle (diagonal included), inv(L)'s strictly lower part transposed in the strictly upper triangle. */
int mqs_debug_factor32(const double *A, double *out, int form, int32_t *not_positive_definite);
/* Frame ingest (csrc/slam_ingest.hip): the reference reads every frame inside its loop (slam2.py:1209-1213); here a frame goes from
 * host memory to the device on a stream of its own while the loop's kernels work on the frames before it.
 *   mqs_slam_ingest_enable  a ring of `slots` device images (W x H bytes each) owned by the handle, an upload stream, one worker
 *                         thread of the library.
 *   mqs_slam_upload       posts the upload of `host_img` (W x H bytes) into ring slot `slot` and returns at once; the worker
 *                         enqueues the copy on the upload stream -- straight from `host_img` (pinned != 0: page-locked memory
 *                         that stays valid until the frame has been processed) or through the slot's pinned staging buffer
 *                         (pinned == 0: ordinary memory, valid until mqs_slam_wait_upload of the slot has returned).
 *                         The caller reuses a slot only once the loop is done with the image in it (the previous image of the
 *                         next mqs_slam_track call is still in use).
 *   mqs_slam_wait_upload  returns the slot's device image, to be passed to mqs_slam_start / mqs_slam_track (and to nothing else: the
 *                         upload may still be under way).  The handle's streams wait for the upload on the device, each when it is about
 *                         to read the image -- with the pair tracked ahead (mqs_slam_set_next) that is the side stream; the loop's stream
 *                         only in front of a tracker or a corner detection of its own.  The host does not block beyond the worker's hand-off. */
#define MQS_SLAM_INGEST_MAX_SLOTS 16
int mqs_slam_ingest_enable(mqs_slam *s, int slots);
int mqs_slam_upload(mqs_slam *s, int slot, const uint8_t *host_img, int pinned);
int mqs_slam_wait_upload(mqs_slam *s, int slot, const uint8_t **image_dev);
/*   mqs_slam_prepare_next   the tracker's pyramid of the NEXT image pair (levels, derivatives, border-extended copies of both images),
 *                         enqueued on a stream of its own so that it runs under the CURRENT frame's pose kernels (RANSAC hypotheses,
 *                         selection, decision: ~90 us of one to 256 small workgroups).  Call it before the mqs_slam_track of the
 *                         current frame; the mqs_slam_track of the next frame recognises the pair by its two image pointers, waits (on
 *                         the device) for that launch and runs the tracker alone.  If the pair does not come (the current frame is
 *                         rejected: its predecessor stays the previous image) the pyramid is built inside the call as always.
 *                         prev_slot / next_slot >= 0: ring slots of the frame ingest, whose uploads the side stream waits for
 *                         (next_img_dev may be NULL then); -1: the caller vouches that the image is complete on the device.
 *                         Results are the same bit for bit. */
int mqs_slam_prepare_next(mqs_slam *s, const uint8_t *prev_img_dev, int prev_slot, const uint8_t *next_img_dev, int next_slot);
/*   mqs_slam_set_next     the same, folded into the loop's call: names the frame BEHIND the one the next mqs_slam_track handles (this_slot /
 *                         next_slot: ring slots or -1, as above); that mqs_slam_track then enqueues the pair's pyramid on the side stream
 *                         behind its own launches and before it waits for its result -- the host's work for it is off the frame's path too.
 *                         Where the tracker's side stream exists it also TRACKS the named pair there as soon as the current frame's
 *                         hypotheses are out (the surviving tracks are known then), beside the current frame's decision kernel; the next
 *                         mqs_slam_track uses that when the current frame was an ordinary one (MQS_SLAM_TRACK_AHEAD=0: the pyramid alone). */
int mqs_slam_set_next(mqs_slam *s, int this_slot, const uint8_t *next_img_dev, int next_slot);
/*   mqs_slam_pipeline     on = 1: the frame named by mqs_slam_set_next is also ENQUEUED -- its hypothesis and decision kernels behind the
 *                         current frame's decision, before the call waits for the current frame's result.  They read that decision on
 *                         the device: behind an ordinary frame they run (no host round trip between two ordinary frames), behind a keyframe
 *                         or a rejected frame they do nothing and the next mqs_slam_track issues the frame as always.  Contract: after a
 *                         call that reported an ordinary frame the next mqs_slam_track must be for (that image, the named one) --
 *                         MQS_E_ARG otherwise -- and reads of the handle's state between the two calls may already show the enqueued
 *                         frame.  slam2.py's loop (:1200-1253) has nothing between two ordinary frames.  Results are the same bit for bit. */
int mqs_slam_pipeline(mqs_slam *s, int on);
int mqs_slam_log_enable(mqs_slam *s, int64_t capacity);
int mqs_slam_read_log(mqs_slam *s, int32_t *lm, int32_t *pose, double *uv, int64_t cap, int64_t *n);
int mqs_slam_write_back(mqs_slam *s, const double *map, int n, const double *pose_prev, const double *pose_key);
int mqs_slam_start(mqs_slam *s, const uint8_t *img_dev, const float *objp0, const float *imgp0, int n0, double *pose_out);
int mqs_slam_track(mqs_slam *s, const uint8_t *prev_img_dev, const uint8_t *img_dev, double *result);
int mqs_slam_flush(mqs_slam *s, double *result);
int mqs_slam_read_tracks(mqs_slam *s, float *pts, float *base, int32_t *lm, int32_t *tid, int cap, int32_t *n);
int mqs_slam_read_map(mqs_slam *s, float *objp, int cap, int32_t *n);

/* ---------------------------------------------------------------------------------------
 * Timing helper used by bench.py: average duration (ms) of `reps` back-to-back launches of
 * one triangulation kernel measured with hipEvents on `stream` (kernel: 0 = linear_ls,
 * 1 = iterative_ls, 2 = linear_eigen; 10 / 11 / 12: the same with `u` pointing at float32 observations).
 * ------------------------------------------------------------------------------------- */
int mqs_time_triangulate_dev(int kernel, const double *u, const double *P, int C, int64_t N,
                             double tolerance, int max_iter, double *x, int32_t *status, uint8_t *ok,
                             int reps, void *stream, float *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* MQSLAM_H */
