/*
 * vo_hip.h -- C ABI of libvo_hip.so: the MI355X (gfx950) implementation of the per-frame hot path
 * of giuliano-97/visual_odometry.  Plain pointers and sizes only; every entry point returns an int
 * status (0 = ok, < 0 = error, see vo_strerror) and never throws or aborts.
 *
 * The reference is MATLAB; the binding a maintainer adds is a MEX gateway per operator (mex/ gateways,
 * INTEGRATION.md) that passes mxGetData pointers straight through:
 *
 *   vo_klt_track            <- KLTTracker.track                  modules/KLTTracker.m:22-41
 *   vo_select_candidates    <- selectCandidateKeypoints          modules/selectCandidateKeypoints.m:1-36
 *   vo_harris_response,     <- harris / selectKeypoints          old/harris.m:1-25, old/selectKeypoints.m:1-16
 *   vo_select_keypoints_greedy
 *   vo_triangulate_candidates <- per-candidate loop body         modules/VisualOdometry.m:87-113,
 *                                                                modules/candidateTriangulation.m:37-57
 *   vo_triangulate_points   <- the toolbox call `triangulate`    modules/VisualOdometry.m:97-101 (one candidate a call)
 *   vo_angle_deg            <- calculateAngleDeg                 utils/calculateAngleDeg.m:1-11
 *   vo_in_front             <- isInFrontOfCamera                 modules/isInFrontOfCamera.m:1-12
 *   vo_p3p_ransac           <- estimateWorldCameraPose call      modules/VisualOdometry.m:285-289,
 *                              (P3P + RANSAC/MSAC)               old/ransacLocalization.m:1-147, old/p3p.m:41
 *   vo_reproj_error         <- computeReprojectionError          modules/VisualOdometry.m:48-61
 *
 * Layout (vo_layout): col_major = 1 is MATLAB's memory layout for EVERYTHING passed in the call
 * (H x W column-major images, N x 2 / N x 3 column-major point arrays, column-major 4 x 3 poses);
 * col_major = 0 is C / numpy row-major for everything.  index_base is 1 for MATLAB pixel
 * coordinates (pixel centres at 1..W), 0 for zero-based.  K is always 9 doubles holding the usual
 * 3 x 3 intrinsic matrix row-major, which is byte-for-byte MATLAB's column-major
 * cameraParams.IntrinsicMatrix (= K^T, utils/dataLoaderKitti.m:50-53).
 * Pose = 4 x 3 [R_WC; t_WC] (modules/VisualOdometry.m:318).
 *
 * Ownership: the caller owns every buffer; the library owns device memory inside vo_ctx.
 * Threading: one vo_ctx per thread/GPU; calls on one ctx must be serialised by the caller.
 * There is NO CPU fallback: without a HIP device vo_ctx_create fails with VO_ERR_NO_DEVICE.
 */
#ifndef VO_HIP_H
#define VO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

enum {
    VO_OK = 0,
    VO_ERR_NO_DEVICE = -1,
    VO_ERR_BAD_ARG = -2,
    VO_ERR_HIP = -3,
    VO_ERR_OOM = -4,
    VO_ERR_UNSUPPORTED = -5,
    VO_END = 1 /* vo_ingest_next: the sequence is exhausted (not an error) */
};

typedef struct {
    int col_major;  /* 1 = MATLAB layout, 0 = row-major */
    int index_base; /* 1 = MATLAB pixel coordinates, 0 = zero-based */
} vo_layout;

/* modules/KLTTracker.m:11-14 (defaults 4 / 2 / [41 41] / 50); eps is the LK stop threshold */
typedef struct {
    int levels;
    double max_bidir_err; /* < 0 disables the backward pass */
    int block_h, block_w; /* BlockSize = [height width], odd */
    int max_iters;
    double eps;           /* pixels; 0 -> 0.01 */
} vo_klt_opts;

enum { VO_SCORER_MINEIGEN = 0, VO_SCORER_HARRIS = 1 };

/* modules/selectCandidateKeypoints.m:13-16 (+ scorer switch: live code is min-eigen, north_star
 * and old/harris.m name Harris) */
typedef struct {
    double min_distance;
    double min_quality;
    int filter_size; /* Gaussian window (min-eigen) / patch size (Harris), odd, <= 15 */
    int keep;        /* CandidatesToKeep */
    int scorer;
    double kappa;    /* Harris only */
} vo_sel_opts;

/* modules/VisualOdometry.m:285-289 options + reproducibility controls */
typedef struct {
    int num_hyp;          /* MaxNumTrials */
    double max_reproj_err;/* MaxReprojectionError, pixels */
    double confidence;    /* (0,1) adaptive stop with sequential-RANSAC semantics; <= 0: use all */
    uint64_t seed;        /* Philox4x32-10 key of the hypothesis stream */
    int mode;             /* 0 = inlier count (old/ransacLocalization.m:86-111), 1 = MSAC cost */
    int min_inliers;      /* old/ransacLocalization.m:102 */
} vo_ransac_opts;

typedef struct {
    int status;      /* 0 ok, 1 fewer than 4 points, 2 no hypothesis reached min_inliers */
    int best_hyp, best_sol, trials_used, num_inliers;
} vo_ransac_info;

const char* vo_strerror(int code);
int vo_version(void);

/* ---- context ---- */
int vo_ctx_create(int device, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
/* run the library's kernels on a caller-owned hipStream_t (NULL restores the ctx-owned stream) */
int vo_ctx_set_stream(vo_ctx* ctx, void* hip_stream);
void* vo_ctx_get_stream(vo_ctx* ctx);
int vo_ctx_sync(vo_ctx* ctx);
/* A branch of a frame's dependency graph on a second stream (high-priority on request: vo_ctx_lane_priority) (the reference has no counterpart: MATLAB
 * runs the steps of modules/VisualOdometry.m:264-346 one after the other, although the pose step :285-289 needs only
 * the landmark tracks of :278 and the candidate tracking of :71 needs neither).  Between _begin and _end every
 * vo_dev_* call of this ctx is enqueued on the lane stream, which starts behind everything enqueued on the ctx stream
 * so far; after _end the calls go to the ctx stream again and run beside the lane; vo_ctx_lane_join makes the ctx
 * stream wait for the lane's work (call it before the first consumer of the lane's results).  Scheduling only: the
 * results are those of the same calls made in order on one stream.  Pyramids issued ahead (vo_dev_frame_prefetch)
 * that both branches read: make the ctx stream wait for them first (vo_dev_frame_wait).  vo_ctx_lane_resume reopens
 * the lane behind its own earlier work without a new dependency on the ctx stream (to enqueue the other branch's
 * first launch before the rest of the lane's: the host then feeds both streams early). */
int vo_ctx_lane_begin(vo_ctx* ctx);
int vo_ctx_lane_resume(vo_ctx* ctx);
int vo_ctx_lane_end(vo_ctx* ctx);
int vo_ctx_lane_join(vo_ctx* ctx);
/* No counterpart in the reference.  high != 0: the lane is created as a high-priority stream, so that its tracking
 * launch is dispatched in front of the ctx stream's (the landmark track ends first and the pose chain runs beside the
 * rest of the candidate track: 0.1 ms of a 50k-keypoint step).  Only for a process that holds ONE ctx on a GPU it does
 * not share: with more hardware queues than the process may hold, a high-priority stream can wait for ever.  Before
 * the lane's first use; the default is the ctx stream's priority. */
int vo_ctx_lane_priority(vo_ctx* ctx, int high);
const char* vo_ctx_last_error(vo_ctx* ctx);
/* counters: [0] LK iterations executed (window passes), [1] keypoints tracked, [2] LK template
 * set-ups (level x direction x point), [3] hypotheses scored */
int vo_ctx_read_counters(vo_ctx* ctx, uint64_t out[4], int reset);
/* Run statistics of sequence-sharded runs (one sequence per GPU, module_tests/VisualOdometry_test.m:88-103
 * is the loop that is sharded; the only collective of the whole path).  Every rank passes its own record; all ranks get
 * the job's record back: counters summed, elapsed_s = the slowest rank's.  comm: an ncclComm_t of RCCL made by the caller
 * (any host language: ncclCommInitRank with one rank per GPU), NULL for a single rank (the record comes back unchanged).
 * RCCL is looked up at call time in the process (the copy the host already loaded, else librccl.so): the library itself
 * does not link against it.  Two ncclAllReduce of five doubles on the ctx stream, then one small copy back. */
typedef struct vo_stats {
    double frames, keypoints_tracked, keypoints_valid, ransac_inliers, elapsed_s;
} vo_stats;
int vo_stats_gather(vo_ctx* ctx, void* rccl_comm, const vo_stats* local, vo_stats* job);
/* HIP-event stopwatch on the ctx stream (bench.py's live kernel timing) */
int vo_timer_start(vo_ctx* ctx);
int vo_timer_stop_ms(vo_ctx* ctx, float* ms);
/* many intervals on the ctx stream read back with one synchronisation: begin / end bracket a lap without
 * waiting; vo_timer_laps_ms waits for the last lap, returns the sum of all laps and their number, and resets */
int vo_timer_lap_begin(vo_ctx* ctx);
int vo_timer_lap_end(vo_ctx* ctx);
int vo_timer_laps_ms(vo_ctx* ctx, float* total_ms, int* laps);

/* measurement of the pose step's scoring kernel: while on, every k_score launch of this ctx is one lap of the stopwatch
 * above (read the sum with vo_timer_laps_ms) and counts the pose x point inlier tests it executes (26 fp64 flop each:
 * 11 fused multiply-adds, 3 multiplications, 1 compare); vo_ctx_read_score_tests returns the count (synchronises) */
int vo_ctx_time_score(vo_ctx* ctx, int on);
int vo_ctx_read_score_tests(vo_ctx* ctx, uint64_t* tests, int reset);

/* ==== host-pointer entry points (what the MEX gateways bind) ==== */

/* KLTTracker.track, modules/KLTTracker.m:22-41.  pts: N x 2 [x y]; outputs follow the class of
 * the input points (pts_is_double).  N = 0 returns VO_OK and touches nothing (:33-36).
 * out_scores may be NULL: the third output of track is then not computed (the live pipeline discards it,
 * modules/VisualOdometry.m:71,278 `[.., .., ~] = obj.tracker.track(...)`; a gateway called with nargout < 3 passes NULL). */
int vo_klt_track(vo_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int H, int W, vo_layout lay,
                 const void* pts_xy, int N, int pts_is_double, const vo_klt_opts* opts,
                 void* out_pts_xy, uint8_t* out_valid, void* out_scores);

/* selectCandidateKeypoints, modules/selectCandidateKeypoints.m:1-36.  existing: N_exist x 2.
 * out_xy: room for keep x 2 floats, written as an M x 2 array in the chosen layout (M rows);
 * out_linear_idx (optional): 0-based linear memory index of each selected pixel;
 * out_metric (optional): corner metric.  *out_M = M <= keep. */
int vo_select_candidates(vo_ctx* ctx, const uint8_t* img, int H, int W, vo_layout lay,
                         const void* existing_xy, int N_exist, int existing_is_double,
                         const vo_sel_opts* opts, float* out_xy, int32_t* out_linear_idx,
                         double* out_metric, int* out_M);

/* old/harris.m:1-25 (scores, H x W doubles in the image's layout) */
int vo_harris_response(vo_ctx* ctx, const uint8_t* img, int H, int W, vo_layout lay, int patch_size,
                       double kappa, double* out_scores);
/* corner response of either scorer (VO_SCORER_*), same output shape as vo_harris_response */
int vo_corner_response(vo_ctx* ctx, const uint8_t* img, int H, int W, vo_layout lay, int scorer,
                       int filter_size, double kappa, double* out_scores);
/* old/selectKeypoints.m:1-16: greedy top-num with (2r+1)^2 suppression on a score image;
 * out_rc: 2 x num [row; col] (column-major) in index_base coordinates */
int vo_select_keypoints_greedy(vo_ctx* ctx, const double* scores, int H, int W, vo_layout lay, int num,
                               int r, int32_t* out_rc);

/* modules/VisualOdometry.m:87-113 batched: first_xy / curr_xy C x 2, first_poses P x (4x3),
 * first_pose_idx (optional, C entries into the pose table counted from lay.index_base; NULL: P == C,
 * candidate i uses pose i),
 * curr_pose 4x3, K.  Outputs: out_X C x 3, out_reproj C (mean over the two views),
 * out_angle_deg C (calculateAngleDeg), out_valid C (positive depth in both views). */
int vo_triangulate_candidates(vo_ctx* ctx, vo_layout lay, const double* first_xy, const double* curr_xy,
                              const double* first_poses, int P, const int32_t* first_pose_idx, int C,
                              const double* curr_pose, const double* K, double* out_X,
                              double* out_reproj, double* out_angle_deg, uint8_t* out_valid);

/* `[worldPoints, reprojectionErrors, validIndex] = triangulate(matchedPoints1, matchedPoints2, camMatrix1, camMatrix2)`
 * with the argument shapes of the reference's call sites (modules/VisualOdometry.m:97-101,
 * modules/candidateTriangulation.m:47-51, processFrame.m:55-59): n point pairs (n x 2 each), two 4 x 3 camera matrices
 * ([X 1] * camMat = [u v w] * w, as `cameraMatrix` returns them), X n x 3, error = mean over the two views, valid =
 * positive depth in both.  What matlab/triangulate.m binds, so that the UNMODIFIED class method
 * (modules/VisualOdometry.m:87-142) reaches the GPU -- one candidate a call; the batched form is
 * vo_triangulate_candidates. */
int vo_triangulate_points(vo_ctx* ctx, vo_layout lay, const double* pts0, const double* pts1, int n,
                          const double* cam_mat0, const double* cam_mat1, double* out_X, double* out_reproj,
                          uint8_t* out_valid);

/* utils/calculateAngleDeg.m:1-11, batched over n landmarks (n x 3) */
int vo_angle_deg(vo_ctx* ctx, vo_layout lay, const double* landmarks, int n, const double* pose0,
                 const double* pose1, double* out_angle_deg);
/* modules/isInFrontOfCamera.m:1-12: cam_mat 4 x 3, points n x 3 */
int vo_in_front(vo_ctx* ctx, vo_layout lay, const double* cam_mat, const double* points, int n,
                uint8_t* out_valid);
/* modules/VisualOdometry.m:48-61 */
int vo_reproj_error(vo_ctx* ctx, vo_layout lay, const double* keypoints_xy, const double* landmarks,
                    int n, const double* pose, const double* K, double* out_err);

/* modules/VisualOdometry.m:285-289 / old/ransacLocalization.m.  img_pts M x 2, world_pts M x 3.
 * sample_idx (optional): num_hyp x 3 zero-based indices, row-major, overriding the seeded stream.
 * hyp_poses_in (optional): num_hyp x 4 x 12 doubles [R_cw(9, row-major) t_cw(3)] scored INSTEAD of
 * generating hypotheses -- the "identical hypotheses" hook of the parity contract.
 * Outputs: R_wc (3x3 world orientation, in the chosen layout), t_wc (world location) -- i.e.
 * pose = [R_wc; t_wc]; inliers M; info.  Optional: hyp_counts / hyp_costs (num_hyp x 4),
 * hyp_poses_out (num_hyp x 4 x 12, the hypotheses that were scored). */
int vo_p3p_ransac(vo_ctx* ctx, vo_layout lay, const double* img_pts, const double* world_pts, int M,
                  const double* K, const vo_ransac_opts* opts, const int32_t* sample_idx,
                  const double* hyp_poses_in, double* out_R_wc, double* out_t_wc, uint8_t* out_inliers,
                  vo_ransac_info* info, int32_t* hyp_counts, uint64_t* hyp_costs, double* hyp_poses_out);

/* old/p3p.m:41-217 batched: world_pts n x 3 x 3 (row-major, point-major), bearings likewise;
 * poses n x 4 x 12 ([C(3) R(9 row-major, cam->world)] per solution) */
int vo_p3p_batch(vo_ctx* ctx, const double* world_pts, const double* bearings, int n, double* out_poses);

/* Relative-pose step of the bootstrap: estimateEssentialMatrix(p0, p1, cameraParams, 'MaxNumTrials', n,
 * 'Confidence', c, 'MaxDistance', d) -- modules/bootstrap.m:76-79.  MSAC around the normalised
 * eight-point algorithm (the toolbox's solver is closed; definition N10 in oracle/vo_oracle.h).
 * pts0 / pts1: M x 2 pixel coordinates in K's origin; opts->max_reproj_err carries MaxDistance (Sampson
 * distance, pixels^2), opts->min_inliers the least inlier count of a winning hypothesis.  out_E: 3 x 3 with
 * x1' E x0 = 0 on K^-1-normalised points, singular values (1, 1, 0).  info->status 0 ok / 1 fewer than 8
 * correspondences / 2 no hypothesis qualified.  sample_idx (optional) H x 8, hyp_E_in (optional) H x 9
 * row-major hypotheses to score instead of sampling; hyp_counts / hyp_costs (H) and hyp_E_out (H x 9)
 * optional inspection outputs. */
int vo_essential_ransac(vo_ctx* ctx, vo_layout lay, const double* pts0, const double* pts1, int M, const double* K,
                        const vo_ransac_opts* opts, const int32_t* sample_idx, const double* hyp_E_in, double* out_E,
                        uint8_t* out_inliers, vo_ransac_info* info, int32_t* hyp_counts, uint64_t* hyp_costs,
                        double* hyp_E_out);
/* The same step around a FIVE-POINT solver (D. Nister, PAMI 26(6), 2004), the kind of solver the toolbox function
 * wraps -- modules/bootstrap.m:76-79; definition N10b, oracle/five_point.c.  Unlike the eight-point algorithm it
 * works on planar scenes.  A sample of five correspondences gives up to ten matrices, each scored; the adaptive stop
 * counts samples (N = log(1 - p) / log(1 - w^5)).  info->status 1: fewer than 5 correspondences.  sample_idx
 * (optional) H x 5.  The inspection outputs have TEN slots per sample (10 H counts / costs, 10 H x 9 matrices; unused
 * slots: count 0, the cost of M outliers, E = 0) and info->best_hyp is a slot index (10 h + s); info->trials_used counts
 * samples. */
int vo_essential_ransac_five(vo_ctx* ctx, vo_layout lay, const double* pts0, const double* pts1, int M, const double* K,
                             const vo_ransac_opts* opts, const int32_t* sample_idx, double* out_E, uint8_t* out_inliers,
                             vo_ransac_info* info, int32_t* hyp_counts, uint64_t* hyp_costs, double* hyp_E_out);

/* ==== device-pointer entry points (inputs/outputs resident in HBM; asynchronous on the ctx
 * stream).  Internal convention: image = nv rows of nu contiguous bytes, (u,v) zero-based;
 * points structure-of-arrays. ==== */

#define VO_NUM_FRAME_SLOTS 4
/* build the float pyramid of a device image into a frame slot (no gradient planes: the tracker differentiates the
 * sampled template) */
int vo_dev_frame_build(vo_ctx* ctx, int slot, const uint8_t* d_img, int nu, int nv, int levels);
/* track slot_a -> slot_b; wu/wv = window along u / v.  d_score (and d_fb_err) may be NULL: the match score costs two
 * more patch stagings and a window pass per keypoint and is skipped when nobody reads it (modules/VisualOdometry.m:71,278
 * discard it) */
int vo_dev_klt_track(vo_ctx* ctx, int slot_a, int slot_b, const double* d_pu, const double* d_pv, int n,
                     int wu, int wv, int max_iters, double eps, double max_bidir_err, double* d_qu,
                     double* d_qv, uint8_t* d_valid, float* d_score, float* d_fb_err);
/* The frame's two track calls (modules/VisualOdometry.m:278 landmarks, :71 candidates) as ONE launch whose first n_head
 * points are a set of their own: they are dispatched first, and what vo_dev_wait_head releases starts when the last of
 * them is done -- while the rest of the launch is still running.  No counterpart in the reference (MATLAB runs :278-289
 * and :71 one after the other); it is how the pose chain, which needs the landmark tracks only, runs beside the candidate
 * track without a stream priority: order, not priority.  Points n_head .. n-1 behave as in vo_dev_klt_track.  Call it on
 * the ctx stream (outside the lane), THEN open the lane and call vo_dev_wait_head there, then the head's consumers:
 *     vo_dev_klt_track_head(ctx, ...);  vo_ctx_lane_begin(ctx);  vo_dev_wait_head(ctx);  vo_dev_p3p_ransac(ctx, ...);
 *     vo_ctx_lane_end(ctx);  ...  vo_ctx_lane_join(ctx);
 * vo_dev_wait_head enqueues a one-wave kernel that polls a word the launch's last head keypoint sets; it gives up after
 * ~0.25 s (vo_dev_head_timed_out then reports 1, and the consumers ran on unfinished data).  The head's outputs are
 * written with write-through stores, so kernels that start behind the wait read them although the launch has not ended. */
int vo_dev_klt_track_head(vo_ctx* ctx, int slot_a, int slot_b, const double* d_pu, const double* d_pv, int n_head, int n,
                          int wu, int wv, int max_iters, double eps, double max_bidir_err, double* d_qu, double* d_qv,
                          uint8_t* d_valid, float* d_score, float* d_fb_err);
int vo_dev_wait_head(vo_ctx* ctx);
int vo_dev_head_timed_out(vo_ctx* ctx, int* timed_out);   /* synchronises the ctx stream */
/* u_is_rows: 1 when the u axis is the image's row (y) axis (MATLAB layout) */
int vo_dev_select_candidates(vo_ctx* ctx, const uint8_t* d_img, int nu, int nv, int u_is_rows,
                             const float* d_eu, const float* d_ev, int n_exist, const vo_sel_opts* opts,
                             float* d_out_u, float* d_out_v, int32_t* d_out_lin, double* d_out_metric,
                             int32_t* d_out_count);
/* the same selection in two calls, so that the part that needs only the image (corner response, threshold,
 * regional maxima: modules/selectCandidateKeypoints.m:20) runs on a side stream of the ctx beside whatever is
 * enqueued between them (tracking, pose, triangulation), and only the part that needs the existing points
 * (:23-33) runs where vo_dev_select_candidates would.  Results are identical to the single call.
 * max_exist: upper bound of the n_exist that vo_dev_select_finish will pass.  At most two selections may be
 * open; vo_dev_select_finish finishes the oldest.  d_img must stay untouched until that finish has run.  Open
 * selections own the ctx's two selection work areas: any other selection call on the ctx in between makes the
 * finish fail with VO_ERR_BAD_ARG rather than return another image's corners. */
int vo_dev_select_begin(vo_ctx* ctx, const uint8_t* d_img, int nu, int nv, const vo_sel_opts* opts, int max_exist);
int vo_dev_select_finish(vo_ctx* ctx, int u_is_rows, const float* d_eu, const float* d_ev, int n_exist,
                         float* d_out_u, float* d_out_v, int32_t* d_out_lin, double* d_out_metric,
                         int32_t* d_out_count);
/* Everything about a frame that needs nothing but its image, issued ahead of the frame's turn: the pyramid of
 * `slot` (what vo_dev_frame_build computes) and, with opts != NULL, the first half of a selection on the image
 * (what vo_dev_select_begin computes; it joins the same queue, finish it with vo_dev_select_finish).  The work
 * runs on the ctx's side stream once everything enqueued on the ctx stream so far has run, i.e. beside whatever
 * is enqueued next: called right after frame k's tracking call for frame k+2's image, it fills the thinly
 * occupied tail of frame k's step (pose, triangulation, selection filter) instead of standing in front of frame
 * k+1's tracking.  The ctx stream waits for the pyramid at the first vo_dev_klt_track (or vo_dev_frame_build)
 * that names the slot; vo_ctx_sync waits for the side stream too.  The slot must not be one that work enqueued
 * later reads before that wait (three slots rotate; the reference has no counterpart: MATLAB runs imread,
 * the tracker and detectMinEigenFeatures one after the other, main.m:52-59). */
int vo_dev_frame_prefetch(vo_ctx* ctx, int slot, const uint8_t* d_img, int nu, int nv, int levels,
                          const vo_sel_opts* opts, int max_exist);
/* the ctx stream waits (once) for a pyramid that vo_dev_frame_prefetch is building into `slot` */
int vo_dev_frame_wait(vo_ctx* ctx, int slot);
/* row-major device arrays: xy C x 2, poses P x 12, X C x 3 */
int vo_dev_triangulate(vo_ctx* ctx, const double* d_xy0, const double* d_xy1, const double* d_poses0,
                       int P, const int32_t* d_pose_idx, int C, const double* d_pose1,
                       const double* d_K, double* d_X, double* d_reproj, double* d_angle,
                       uint8_t* d_valid);
/* d_out: R_cw(9) t_cw(3) as doubles followed by vo_ransac_info as 5 int32 (pass >= 128 bytes) */
int vo_dev_p3p_ransac(vo_ctx* ctx, const double* d_xy, const double* d_X, int M, const double* d_K,
                      const vo_ransac_opts* opts, const int32_t* d_sample_idx,
                      const double* d_hyp_poses_in, void* d_out, uint8_t* d_inliers);
/* vo_dev_p3p_ransac for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same kernels, same
 * results as vo_dev_p3p_ransac; the per-hypothesis work arrays (poses, counts, MSAC costs) live in a static device
 * array of the library (2.4 MB, held on every GPU the library runs on until the process ends) instead of the ctx's
 * grow-only buffer, and the call makes no hipMalloc / hipFree.  opts->num_hyp above VO_STATIC_MAX_HYP is refused
 * with VO_ERR_UNSUPPORTED.  All calls on one GPU share the array: calls that overlap on one GPU (two contexts, two
 * threads) race -- serialise them. */
#define VO_STATIC_MAX_HYP 5120
int vo_dev_p3p_ransac_static(vo_ctx* ctx, const double* d_xy, const double* d_X, int M, const double* d_K,
                             const vo_ransac_opts* opts, const int32_t* d_sample_idx,
                             const double* d_hyp_poses_in, void* d_out, uint8_t* d_inliers);

/* ==== device-resident frame loop: VisualOdometry.processFrame,
 * modules/VisualOdometry.m:264-346, with the method candidateTriangulation (:63-151), the state kept
 * in HBM across the five steps and across frames.  Replaces the MATLAB state struct
 * {landmarks, keypoints, reproError, candidate_keypoints, candidate_first_keypoints,
 * candidate_first_poses, candidate_time_indxs} (:308-315, :338-345).  Conventions: row-major host
 * arrays, (u,v) zero-based with K in the same origin, poses 4x3 row-major [R_WC; t_WC] (:318).
 * Not thread-safe per state.  A frame runs on the ctx stream, on the ctx's lane stream (the landmark track and the pose
 * chain, beside the selection of the frame before) and on its side stream (the next image's pyramid and corner
 * detection, when announced); vo_frame_collect / vo_frame_process / vo_ctx_sync wait for all of it. ==== */
typedef struct vo_frame_state vo_frame_state;
typedef struct vo_frame_opts {
    int max_landmarks;            /* maxNumLandmarks, :24 */
    int max_candidates;           /* capacity of the candidate arrays (a cell array in the reference) */
    int max_temporal_recall;      /* maxTemporalRecall, :23 */
    double angular_threshold_deg; /* angularThreshold, :22 */
    double max_reproj_err;        /* maxReprojectionError, :25; strict < at :302, <= at :106 */
    vo_klt_opts klt;              /* :33-37 */
    vo_ransac_opts pose;          /* :285-289; seed + frame number seeds each frame's sample stream */
    vo_sel_opts sel;              /* :332-335 */
} vo_frame_opts;
typedef struct vo_frame_info {
    int32_t status;        /* pose status 0 / 1 / 2 as vo_ransac_info */
    int32_t n_landmarks, n_candidates;   /* sizes of the new state */
    int32_t n_tracked;     /* landmarks' keypoints that tracked validly (:280) */
    int32_t n_inliers, trials_used;
    int32_t n_promoted;    /* candidates appended as landmarks (:123-127) */
    int32_t n_new;         /* new candidate keypoints (:330-345) */
    int32_t overflow;      /* 1 when new candidates were cut at max_candidates */
    int32_t frame_no;
} vo_frame_info;
/* The order-dependent merge of modules/VisualOdometry.m:87-142 as two decisions per pool entry, for a pool given as
 * such: d_values = reproError of the n0 landmarks the frame starts the merge with, in order, followed by the errors of
 * the np promotable candidates in candidate order; cap = maxNumLandmarks.  d_accepted[e] (meaningful for e >= n0): the
 * candidate is taken into the landmarks at its turn (:116-127: below the cap it is appended, at the cap it replaces the
 * FIRST maximum, taken before the edit, if its own error is not larger); d_survives[e]: the entry is a landmark when the
 * loop has ended.  What vo_frame_submit runs for pools beyond a few hundred entries, O(n log n): runs of the pool sorted
 * stably by error, in which every entry then looks itself up, give both (csrc/vo_frame.hip).  Device arrays,
 * asynchronous launches, one synchronisation at the end. */
int vo_dev_merge_ranks(vo_ctx* ctx, const double* d_values, int n0, int np, int cap, uint8_t* d_survives, uint8_t* d_accepted);
int vo_frame_create(vo_ctx* ctx, const vo_frame_opts* opts, int nu, int nv, const double K[9], vo_frame_state** out);
int vo_frame_destroy(vo_ctx* ctx, vo_frame_state* st);
/* load a state (e.g. from bootstrap): landmarks N x 3, keypoints N x 2, repro N, candidates C x 2 twice,
 * first poses C x 12, time indices C (<= -1); frame_no = frames processed so far */
int vo_frame_set_state(vo_ctx* ctx, vo_frame_state* st, const double* landmarks, const double* keypoints,
                       const double* repro_err, int N, const double* cand_keypoints, const double* cand_first_keypoints,
                       const double* cand_first_poses, const int32_t* cand_time_idx, int C, int frame_no);
int vo_frame_get_counts(vo_ctx* ctx, vo_frame_state* st, int* N, int* C);
/* outputs sized by vo_frame_get_counts; any pointer may be NULL */
int vo_frame_get_state(vo_ctx* ctx, vo_frame_state* st, double* landmarks, double* keypoints, double* repro_err,
                       double* cand_keypoints, double* cand_first_keypoints, double* cand_first_poses,
                       int32_t* cand_time_idx);
/* the previous image of the next vo_frame_process call (nv rows of nu bytes; on_device: a device pointer) */
int vo_frame_set_image(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device);
/* one processFrame: prev image = the last image given; returns the pose [R_WC; t_WC] and the counts */
int vo_frame_process(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device, double pose43[12],
                     vo_frame_info* info);
/* The same in two calls, for a loop that does not wait for a frame before it enqueues the next (no counterpart in
 * the reference: modules/VisualOdometry.m:264-346 returns a frame's state before the next begins; here every launch
 * of a frame is sized by the state's capacities and reads the counts on the device, so the host need not know them).
 * vo_frame_submit enqueues one processFrame and returns; vo_frame_collect waits for the OLDEST submitted frame and
 * returns its pose and counts.  At most VO_FRAME_MAX_PENDING frames may be submitted and not collected; results are
 * those of vo_frame_process (which is submit + collect).  A host image is copied during the call; a device image
 * must stay valid and unchanged until its frame has been collected.  vo_frame_set_state / _get_state / _get_counts /
 * _set_image refuse to run while frames are pending. */
#define VO_FRAME_MAX_PENDING 4
int vo_frame_submit(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device);
int vo_frame_collect(vo_ctx* ctx, vo_frame_state* st, double pose43[12], vo_frame_info* info);
/* Look-ahead (optional; no counterpart in the reference, whose loop reads, tracks and detects one after the
 * other, main.m:52-59).  Announce, before a vo_frame_process call, the image that the call AFTER it will
 * process: that image's copy, pyramid and corner detection are then issued on the ctx's side stream behind the
 * coming call's tracking launch, beside the thinly occupied rest of that frame, and the call that processes the
 * announced image (same pointer, same on_device) only waits for them.  Results are identical with and without.
 * The announced image must stay valid and unchanged until the call that processes it has returned; any other
 * image given to that call, vo_frame_set_image, vo_frame_set_state, or another selection on the same ctx in
 * between (a second loop, a stand-alone vo_select_candidates) abandon what was issued.  NULL withdraws. */
int vo_frame_announce(vo_ctx* ctx, vo_frame_state* st, const uint8_t* next_img, int on_device);

/* ==== frame ingest: imread + rgb2gray + im2uint8 as the reference's frame loops use
 * them (main.m:15-30,52-59; utils/dataLoaderKitti.m:29, utils/dataLoaderParking.m:29,
 * utils/dataLoaderMalaga.m:29-30).  Decoded: PNG (non-interlaced; every colour type and bit depth;
 * alpha dropped; 16-bit samples reduced as im2uint8 does, round(x / 257); gray below 8 bits scaled to
 * 0..255; palette expanded to RGB), baseline JPEG (8-bit sequential Huffman, gray or YCbCr with chroma at
 * 1:1 or 2:1 per axis, restart intervals; progressive / arithmetic / CMYK / multi-scan files are
 * VO_ERR_UNSUPPORTED) and binary PGM / PPM.  In the decode pool a JPEG's bit stream is decoded on the
 * host and everything after it (dequantisation, inverse DCT, chroma upsampling, YCbCr -> RGB, rgb2gray)
 * runs on the device; vo_image_decode does the same arithmetic on the host, bit for bit. ==== */

/* host only, no GPU: size of the decoded image (channels = 1 or 3, interleaved rows) */
int vo_image_info(const uint8_t* file_bytes, size_t n, int* H, int* W, int* channels, int* bit_depth);
/* host only, no GPU: imread + im2uint8 -> H x W x channels uint8, row-major interleaved */
int vo_image_decode(const uint8_t* file_bytes, size_t n, uint8_t* out, size_t out_cap);
/* rgb2gray (main.m:20-30,54-59) of an 8-bit image: planar = 0 interleaved rows (H x W x 3 row-major),
 * planar = 1 MATLAB's H x W x 3 column-major planes; the output has the input's pixel order.
 * gray = floor(0.298936021293775390 R + 0.587043074451121360 G + 0.114020904255103250 B + 0.5) in fp64.
 * on_device: both pointers are device memory and the call is asynchronous on the ctx stream. */
int vo_rgb2gray(vo_ctx* ctx, const uint8_t* rgb, int H, int W, int planar, int on_device, uint8_t* gray);

/* Decode pool: worker threads read and decode the files ahead of the consumer into pinned memory, upload
 * on a copy stream of the pool's own and convert colour images to gray there; vo_ingest_next returns the
 * frames in order as device images (H rows of W bytes, what vo_frame_process takes with on_device = 1)
 * and makes the ctx stream wait for the upload.  The image stays valid until the next vo_ingest_next /
 * vo_ingest_close on the same pool (with vo_ingest_set_hold(2): until the one after the next, which is what
 * vo_frame_announce needs).  One consumer thread per pool. */
typedef struct vo_ingest vo_ingest;
typedef struct vo_ingest_opts {
    int threads;  /* decode threads; <= 0: min(8, host threads) */
    int depth;    /* frames in flight (ring slots); <= 0: 2 x threads */
    int keep_rgb; /* 1: hand out colour images as interleaved RGB instead of gray */
} vo_ingest_opts;
int vo_ingest_open(vo_ctx* ctx, const char* const* paths, int n_paths, const vo_ingest_opts* opts, vo_ingest** out);
int vo_ingest_count(vo_ingest* ing);
/* VO_OK and the next frame, VO_END after the last one, or the decode error of that frame (the pool
 * stays usable: the next call moves on to the following file) */
int vo_ingest_next(vo_ingest* ing, const uint8_t** d_img, int* H, int* W, int* channels, int* index);
/* copy the current frame (the one vo_ingest_next returned last) to the host; synchronises the ctx stream */
int vo_ingest_read(vo_ingest* ing, uint8_t* host_out, size_t out_cap);
/* summed wall time the workers spent reading + decoding + enqueueing, and the pool's actual sizes */
/* how many of the images handed out stay valid (default 1); before the first vo_ingest_next, <= depth - 1 */
int vo_ingest_set_hold(vo_ingest* ing, int hold);
int vo_ingest_stats(vo_ingest* ing, double* decode_seconds_total, int* threads, int* depth);
int vo_ingest_close(vo_ingest* ing);

#ifdef __cplusplus
}
#endif
#endif
