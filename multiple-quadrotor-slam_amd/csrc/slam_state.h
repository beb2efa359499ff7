// State of the device-resident frame loop (slam_frame.hip) shared with its in-loop bundle adjuster (slam_ba.hip).
#pragma once
#include "mqs_common.h"

namespace mqs {
namespace slamst {

constexpr int kMaxTracks = 512;                 // capacity of the live-track arrays (the reference tops up to <= 300)
constexpr int kHyp = 256, kSample = 6;          // RANSAC: hypotheses per frame, points per minimal sample (pnp.py)
constexpr int kSampleIters = 5, kPnpIters = 100;
constexpr double kPnpEps = 1e-10;
constexpr int kRes = 40;                        // doubles in the result block
constexpr int kResSlots = 4, kResStride = 64;   // the pinned result ring: a block + its ticket word per launched decision (slot = ticket & 3)

// counters in device memory
enum { C_N = 0, C_NLAND, C_NEXT_TID, C_FRAME, C_NTRI, C_NKEEP, C_KF_PENDING, C_NLOG, C_LOG_OVERFLOW, C_LAST_DECISION, C_COUNT };   // C_LOG_OVERFLOW: sticky, an observation did not fit the log; C_LAST_DECISION: of the last frame that was decided (what an enqueued-ahead frame looks at)
// result block (doubles)
enum { R_DECISION = 0, R_REASON, R_NTRACKS, R_NTRI, R_NINL, R_NOLD, R_NNEW, R_LOST, R_OUTLIER, R_REPROJ, R_HOMOGRAPHY, R_NLAND,
       R_POSE = 12, R_KF_VALID = 24, R_KF_NGOOD, R_KF_NTRACKS, R_KF_NLAND, R_KF_POSE = 28 };

struct SlamDev {
    int32_t *cnt;
    float *pts, *base;
    int32_t *lm, *tid;
    double *map;
    double *pose_key, *pose_prev, *intr;
    float *lk_pts, *lk_err;
    uint8_t *lk_st;
    // the tracker AHEAD of its frame (slam_frame.hip, "speculative tracker"): its results for the frame's kept tracks (n = C_NKEEP of the
    // frame before, in that frame's kept order) and, written by that frame's commit, where track r of the live state sits among them
    float *lk_pts_b, *lk_err_b;
    uint8_t *lk_st_b;
    int32_t *spec_map;
    float *t_pts, *t_base;
    int32_t *t_lm, *t_tid;
    double *objp_t, *imgp_t;
    int32_t *tri_pos, *samples;
    double *pose_r;
    int32_t *sel;
    uint8_t *inl_mask;
    double *pnp_info;
    double *kf_objp, *kf_imgp, *kf_p0, *kf_p1;
    int32_t *kf_pos;
    double *kf_scratch, *kf_pose, *kf_x, *kf_info;
    int32_t *kf_status;
    uint8_t *mask;
    float *gf_xy;
    int32_t *gf_n;
    double *pnp_poses;               // the RANSAC workspace's pieces the decision kernel reads (mqs_pnp_workspace_layout)
    int32_t *pnp_counts, *pnp_inl;
    double *res;
    double *kf_hand;                 // frame_decide_kernel's hand-over between its two workgroups: ticket word, the keyframe test's ratio, its track count
    double *res_out;                 // the pinned host block the decision kernel leaves a copy of `res` in (no copy / fill launches per frame)
    // the observation log for the bundle adjuster (mqs_slam_log_enable; null: off): what slam2.py's BundleAdjustmentInfoContainer is
    // handed (:519-522, 634-641), as flat device arrays -- (landmark, pose index of the accepted frame, pixel) per observation
    int32_t *log_lm, *log_pose;
    double *log_uv;
    int log_cap;
    // a FREE track's observation is logged under -2 - (track id); when the track becomes a landmark, tid2lm[track id] says which
    // (slam2.py:634-641: a new landmark brings its image points of every frame since the base keyframe along)
    int32_t *tid2lm;
    int tid_cap;
    // with the log: the pose of every accepted frame ([R | t] world -> camera, 12 doubles; index = its number among the accepted
    // frames), as first estimated -- a keyframe's refined pose replaces its first one -- and as the in-loop adjuster rewrites it
    double *traj;
    int traj_cap;
};

struct SlamParams {
    int W, H, target, max_landmarks;
    double radius, quality;
    double max_of_error, max_lost_ratio, max_reproj, max_outlier_ratio, homography_threshold;
    double second_pass_screen_px;   // 0 (default): slam2.py's flow; > 0: mqs_slam_set_second_pass_screen
    unsigned long long seed;
    int homography_refine;          // 1 (default): DLT + the LM refinement, as cv2.findHomography(method = 0); 0: the DLT alone (A/B)
    int null_vector_jacobi;         // 0 (default): the DLT's null vector by inverse iteration, Jacobi sweeps only when that does not settle; 1: always Jacobi (A/B, tests)
    int max_homography_points;      // keyframe_test's random sample (slam2.py:48; the reference: max(4, target / 4), :1088-1089); 0 = all tracks (default)
    int pose_index, base_pose_index; // index this frame gets among the ACCEPTED frames if it is accepted; that of the base keyframe
    int spec;                        // 1: this frame's tracker ran ahead -- its results are lk_*_b[spec_map[i]] for live track i
    int gated;                       // 1: this frame was enqueued before the frame in front of it was decided (mqs_slam_pipeline): its kernels do
                                     //    nothing unless that frame was accepted as an ordinary frame (C_LAST_DECISION == 1)
    unsigned ticket;                 // of this frame's decision launch: the result block goes to slot (ticket & 3) of the pinned ring, the ticket behind it
};

}  // namespace slamst
}  // namespace mqs

struct mqs_slam_ba;                  // slam_ba.hip: the in-loop adjuster's resident state
struct mqs_slam_ingest;              // slam_ingest.hip: the frame-ingest ring, its stream and worker thread

struct mqs_slam {
    int device;
    hipStream_t stream;
    int stream_priority;             // of this stream and the side stream: the device's highest (default), or 0 with MQS_SLAM_STREAM_PRIORITY=normal (A/B)
    mqs::slamst::SlamDev d;
    mqs::slamst::SlamParams p;
    char *arena;
    double *res_host;                // pinned
    bool fused_filter;               // the filter inside the hypotheses' launch (default; MQS_SLAM_FUSED_FILTER=0: two launches)
    void *ws_lk, *ws_gftt, *ws_pnp;
    int64_t ws_lk_bytes, ws_gftt_bytes;
    bool started;
    int accepted, base_pose;         // accepted frames so far (= the next accepted frame's pose index); pose index of the base keyframe
    char *log_arena;
    char *re_arena;                  // re-association scratch (allocated on first use)
    mqs_slam_ba *ba;                 // the in-loop bundle adjuster's resident state (slam_ba.hip; allocated on first use)
    int land_ub;                     // upper bound of the landmarks in the map once the stream has drained (known without waiting)
    int key_pose;                    // pose index of the base keyframe of the live tracks
    mqs_slam_ingest *ingest;         // null until mqs_slam_ingest_enable
    // the NEXT image pair's pyramid, built ahead on a stream of its own under the current frame's pose kernels (mqs_slam_prepare_next):
    // two tracker workspaces, each with the pair whose pyramid it holds (or is being given) and the event behind that launch
    void *ws_lk2;                    // the second tracker workspace (allocated on first use)
    hipStream_t pyr_stream;
    struct { bool valid; const uint8_t *prev, *next; hipEvent_t done; bool has_event; } prep[2];
    struct { bool valid; const uint8_t *prev, *next; int ws; hipEvent_t done; } spec;      // the tracker launched ahead for the pair (prev, next) on the pyramid in workspace ws
    hipEvent_t hyp_done;             // behind a frame's hypothesis launch: its kept tracks (t_pts, C_NKEEP) are what the tracker ahead starts from
    int decide_grid;                 // 2 (default): the keyframe test beside the pose refinement (frame_decide_kernel); MQS_SLAM_DECIDE_SPLIT=0: 1
    bool spec_enabled;               // MQS_SLAM_TRACK_AHEAD=0 switches the tracker ahead off (A/B, tests); the pyramid ahead stays
    int last_decision;               // of the last mqs_slam_track: 0 rejected, 1 frame, 2 keyframe
    struct { bool set; const uint8_t *next; int prev_slot, next_slot; } ahead;      // mqs_slam_set_next: what the next mqs_slam_track prepares behind its own launches
    // mqs_slam_pipeline: the frame named by mqs_slam_set_next is ENQUEUED (hypotheses + decision, gated on the device) behind the current
    // one's kernels before the call waits for the current frame's result -- no host round trip between two ordinary frames
    bool pipeline;
    struct { bool valid; const uint8_t *prev, *img; int ws; unsigned ticket; } pre;  // the frame whose kernels the previous call enqueued
    unsigned ticket;                 // decision launches so far
};
// slam_ingest.hip: a ring slot's device image and upload event, once the worker has enqueued the copy (false: nothing was uploaded into it)
bool mqs_slam_ingest_slot(mqs_slam *s, int slot, const uint8_t **image_dev, hipEvent_t *uploaded);
int mqs_slam_ingest_main_wait(mqs_slam *s, const uint8_t *img);      // slam_ingest.hip: before a kernel on the loop's stream reads `img` (a ring slot's upload is waited for lazily)
void mqs_slam_ba_release(mqs_slam *s);          // slam_ba.hip
void mqs_slam_ingest_release(mqs_slam *s);      // slam_ingest.hip
int mqs_slam_ba_anchor(mqs_slam *s, int n0);     // slam_ba.hip: at the end of mqs_slam_start, with the log on
