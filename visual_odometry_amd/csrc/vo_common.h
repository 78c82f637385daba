// vo_common.h -- internal declarations shared by the HIP translation units of libvo_hip.so.
// Each .hip file holds its kernels and the host launchers for them; vo_api.hip is the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/vo_hip.h"

#define VO_MAX_LEVELS 8
// Every pyramid level is stored with a replicated border of VO_PYR_PAD pixels on each side (pitch = nu + 2 PAD):
// a tracking patch around any position inside the image then lies inside the stored plane, so the tracker stages it
// with unclamped 16-byte rows instead of per-lane row / column clamps (vo_klt.hip stage()).  offs[l] is the start of
// level l's padded plane; pixel (u, v) of the level sits at offs[l] + (v + PAD) * (nus[l] + 2 PAD) + u + PAD.
#define VO_PYR_PAD 32
#define VO_MIN_EIG 1e-3

#define VO_COUNTER_SHARDS 256 // counters are sharded by keypoint / block index to avoid one hot word

struct vo_pyr {            // one frame slot: float pyramid, all levels contiguous
    bool valid = false;
    uint64_t key = 0;      // host-side content hash (host API cache)
    uint64_t stamp = 0;    // LRU
    int nu = 0, nv = 0, levels = 0;
    int nus[VO_MAX_LEVELS], nvs[VO_MAX_LEVELS];
    long offs[VO_MAX_LEVELS];
    long total = 0;
    float* I = nullptr;    // total floats
    size_t cap = 0;        // floats allocated
    uint8_t* img = nullptr; // device copy of the uint8 image (host API)
    size_t img_cap = 0;
    hipEvent_t ready = nullptr;   // vo_dev_frame_prefetch: recorded on the side stream behind the pyramid build
    bool ready_pending = false;   // the ctx stream has not waited for `ready` yet
};

struct vo_buf {            // grow-only device buffer
    void* p = nullptr;
    size_t cap = 0;
};

struct vo_sel_pending;
struct vo_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> lap_ev;   // vo_timer_lap_*: begin / end events of the laps recorded so far
    int laps = 0; bool lap_open = false;
    // side stream for work that depends on the new image only (corner detection) and runs beside tracking
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // lane: a second, high-priority stream for a branch of the frame's dependency graph that runs beside the main
    // one (landmark tracking -> pose, beside candidate tracking); vo_ctx_lane_begin / _end / _join
    hipStream_t lane = nullptr, lane_saved = nullptr;
    hipEvent_t ev_lane_fork = nullptr, ev_lane_done = nullptr;
    bool in_lane = false, lane_pending = false;
    bool lane_high_priority = false;   // vo_ctx_lane_priority: asked for before the lane exists (and granted)
    bool lane_is_high = false;         // the lane stream that exists was created with the high priority
    // set (by whichever thread creates another ctx or a decode pool on this device) when the process stops being
    // "one ctx alone on its GPU": the owner takes its lane back to the ctx stream's priority before the lane's next use
    std::atomic<bool> lane_demote{false};
    hipEvent_t ev_lane_guard = nullptr;   // out-of-lane users of lane-shared buffers wait for the lane's work so far
    struct vo_sel_pending* sel_pending = nullptr;  // vo_dev_select_begin .. vo_dev_select_finish
    uint64_t sel_gen[2] = {0, 0};   // bumped whenever a selection work area is handed out: work issued ahead into it
                                    // (look-ahead, open selections) is valid only while the number stands
    bool resp_attr_set = false;  // k_response's dynamic-LDS limit raised on this context's device (a per-device attribute)
    vo_pyr frames[VO_NUM_FRAME_SLOTS];
    uint64_t clock = 0;
    unsigned long long* d_counters = nullptr; // VO_COUNTER_SHARDS x 4 u64
    vo_buf scratch[20];                        // purpose-indexed scratch (see vo_api.hip; [16] = vo_rgb2gray staging, [17] = second selection set, [18] = response / greedy entry points, [19] = vo_dev_merge_ranks)
    bool time_score = false;   // vo_ctx_time_score: laps around the pose step's k_score launches + a count of their tests
    bool head_open = false;    // a vo_dev_klt_track_head launch has been enqueued (vo_dev_wait_head may follow)
    unsigned head_epoch = 0;   // vo_dev_klt_track_head launches so far (the flag word holds the last finished one's number)
    std::string last_error;
};
// behind the counters, the statistics staging (256 B) and the pose step's ticket (256 B): 256 B for the head words
inline unsigned long long* vo_int_head_words(vo_ctx* ctx) {
    return reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->d_counters) + VO_COUNTER_SHARDS * 4 * sizeof(unsigned long long) + 512);
}

#define VO_HIP_CHECK(ctx, expr)                                                        \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__);    \
            return e__ == hipErrorOutOfMemory ? VO_ERR_OOM : VO_ERR_HIP;               \
        }                                                                              \
    } while (0)

int vo_buf_reserve(vo_ctx* ctx, vo_buf* b, size_t bytes);

// ---- launchers (each returns VO_OK or an error code; all asynchronous on ctx->stream) ----
int vo_launch_frame_build(vo_ctx* ctx, vo_pyr* f, const uint8_t* d_img, hipStream_t stream = nullptr);  // nullptr: ctx->stream
int vo_launch_klt(vo_ctx* ctx, const vo_pyr* A, const vo_pyr* B, const double* d_pu, const double* d_pv, int n,
                  int wu, int wv, int max_iters, double eps, double max_bidir, double* d_qu, double* d_qv,
                  uint8_t* d_valid, float* d_score, float* d_fb,
                  const int32_t* d_n = nullptr,    // d_n: the keypoint count on the device, n its upper bound
                  int n_head = 0);                 // > 0: the first n_head keypoints are a head (vo_dev_klt_track_head)
int vo_launch_wait_head(vo_ctx* ctx, const int32_t* d_n = nullptr, int32_t* host_fault = nullptr);   // ctx->stream waits (a polling kernel) for the last head launched (d_n: that launch's device count; host_fault: a pinned host word that receives 2 when the wait gives up)
// three words of the ctx for the head of a tracking launch: arrival count, flag (epoch of the last finished head), time-out
// (vo_int_head_words, above)

struct vo_sel_work {       // device scratch of the selection pipeline
    double* resp; unsigned long long* maxbits; int32_t* blk_counts; int32_t* blk_offs;
    int32_t* lin; float* lu; float* lv; double* metric; uint8_t* keep; uint8_t* sel;
    int32_t* m0; unsigned long long* cell_key; int32_t* cell_idx; int32_t *klist, *kcell; int cap; int nblk;
    // uniform grid over the detected corners (distance filter)
    float grid_cs; int grid_gu, grid_gv;
    int32_t *grid_count, *grid_cursor, *grid_offs, *grid_total; float *grid_su, *grid_sv; int32_t* grid_si;
};
void vo_select_grid_dims(int nu, int nv, double min_distance, float* cs, int* gu, int* gv);
int vo_launch_response(vo_ctx* ctx, hipStream_t stream, const uint8_t* d_img, int nu, int nv, int scorer, int fs,
                       double kappa, double* d_resp, unsigned long long* d_maxbits, unsigned* d_zero = nullptr,
                       long zero_words = 0);   // d_zero: a word range cleared in the launch that clears *d_maxbits
// selection in two phases: detect needs the image only (response, threshold, regional maxima, corner list);
// filter needs the existing points (distance filter, selectUniform).  ne_dev (optional): the number of
// existing points is ne_dev[0] + ne_dev[1] read on the device, n_exist is then only its upper bound.
int vo_launch_select_detect(vo_ctx* ctx, hipStream_t stream, const uint8_t* d_img, int nu, int nv, const vo_sel_opts* o,
                            vo_sel_work* w);
int vo_launch_select_filter(vo_ctx* ctx, hipStream_t stream, int nu, int nv, int u_is_rows, const float* d_eu,
                            const float* d_ev, int n_exist, const int32_t* ne_dev, const vo_sel_opts* o, vo_sel_work* w,
                            float* d_out_u, float* d_out_v, int32_t* d_out_lin, double* d_out_metric,
                            int32_t* d_out_count,
                            const double* d_kpL = nullptr, int capL = 0, const double* d_cuC = nullptr, int capC = 0,
                            hipEvent_t after_kill = nullptr);   // recorded between the two launches (vo_frame_submit)
// side stream: fork = the side stream waits for everything enqueued on the ctx stream so far; join = the ctx
// stream waits for everything enqueued on the side stream so far
int vo_int_side_ensure(vo_ctx* ctx);   // the side stream exists
int vo_int_side_fork(vo_ctx* ctx);
int vo_int_lane_ensure(vo_ctx* ctx);   // creates the ctx's lane stream (vo_ctx_lane_*) if it does not exist yet; demotes it when asked to
// Every owner of HIP streams of the library on a device announces itself here (vo_ctx_create, vo_ingest_open): a
// high-priority lane is only granted to, and only kept by, a ctx that is alone on its device (vo_ctx_lane_priority).
void vo_int_register_streams(int device, vo_ctx* owner_or_null);
// the ctx stream waits for whatever the lane has been given so far (no-op without a lane, or inside it)
int vo_int_wait_for_lane(vo_ctx* ctx);
int vo_int_side_join(vo_ctx* ctx);
int vo_launch_select(vo_ctx* ctx, const uint8_t* d_img, int nu, int nv, int u_is_rows, const float* d_eu,
                     const float* d_ev, int n_exist, const vo_sel_opts* o, vo_sel_work* w, float* d_out_u,
                     float* d_out_v, int32_t* d_out_lin, double* d_out_metric, int32_t* d_out_count);
int vo_greedy_num_blocks(long n);
int vo_launch_greedy(vo_ctx* ctx, double* d_scores_work, int nu, int nv, int num, int r, int32_t* d_out_uv,
                     unsigned long long* d_tmp);

struct vo_strides { long xi, xc; }; // element (i, c) at base[i*xi + c*xc]
int vo_launch_triangulate(vo_ctx* ctx, const double* d_xy0, const double* d_xy1, vo_strides sxy,
                          const double* d_poses0, int pose_col_major, const int32_t* d_pose_idx, int C, int P,
                          const double* d_pose1, const double* d_K, double* d_X, vo_strides sX,
                          double* d_reproj, double* d_angle, uint8_t* d_valid,
                          const int32_t* d_C = nullptr);   // d_C: the candidate count on the device, C its upper bound
int vo_launch_triangulate_cam(vo_ctx* ctx, const double* d_xy0, const double* d_xy1, vo_strides sxy, const double* d_cam0,
                              const double* d_cam1, int cam_col_major, int n, double* d_X, vo_strides sX, double* d_reproj,
                              uint8_t* d_valid);
int vo_launch_angle(vo_ctx* ctx, const double* d_X, vo_strides sX, int n, const double* d_pose0,
                    const double* d_pose1, int pose_col_major, double* d_angle);
int vo_launch_in_front(vo_ctx* ctx, const double* d_cam, int cam_col_major, const double* d_X, vo_strides sX,
                       int n, uint8_t* d_valid);
// flags for events that only order streams of one device (not read by the host): VO_EVENT_FLAGS overrides (A/B)
unsigned vo_int_event_flags();
int vo_launch_reproj(vo_ctx* ctx, const double* d_xy, vo_strides sxy, const double* d_X, vo_strides sX, int n,
                     const double* d_pose, int pose_col_major, const double* d_K, double* d_err,
                     const int32_t* d_n = nullptr);   // d_n: the count on the device (n is then its upper bound)

struct vo_ransac_work {
    double* Rt; int32_t* cnt; unsigned long long* cost;
    // cnt | cost | {bound[2], list_n} are one contiguous block of zero_bytes bytes, zeroed at the start of a run
    // (k_score adds into cnt / cost); list: pose ids that survive the first quarter of the points (k_survivors)
    unsigned long long* bound = nullptr; int32_t* list_n = nullptr; int32_t* list = nullptr; size_t zero_bytes = 0;
    int32_t* ticket = nullptr;   // one int that is zero between launches (k_pose_small); null: that form is not used
};
int vo_launch_ransac(vo_ctx* ctx, const double* d_xy, vo_strides sxy, const double* d_X, vo_strides sX, int M,
                     const double* d_K, const vo_ransac_opts* o, const int32_t* d_sample_idx,
                     const double* d_poses_in, vo_ransac_work* w, void* d_out, uint8_t* d_inliers,
                     const int32_t* d_M = nullptr,    // d_M: the point count on the device (M is then its upper bound)
                     int score_all = 0,               // 1: no early exit of the adaptive run (per-hypothesis outputs wanted)
                     double* d_pose_out = nullptr);   // optional: the winning pose as [R_WC; t_WC], written by the last kernel
int vo_launch_essential(vo_ctx* ctx, const double* d_p0, const double* d_p1, vo_strides sp, int M, const double* d_K,
                        const vo_ransac_opts* o, const int32_t* d_sample_idx, const double* d_E_in, vo_ransac_work* w,
                        void* d_out, uint8_t* d_inliers, int solver = 8);
int vo_static_scratch_resolve(int device);   // vo_ransac.hip: address of this device's copy of the static work array
int vo_launch_p3p_batch(vo_ctx* ctx, const double* d_P, const double* d_f, int n, double* d_poses);

// ---- helpers of vo_api.hip shared with vo_frame.hip ----
int vo_int_frame_prepare(vo_ctx* ctx, vo_pyr* f, int nu, int nv, int levels);
int vo_int_sel_work_prepare(vo_ctx* ctx, int nu, int nv, int keep, int n_exist, double min_distance, vo_sel_work* w,
                            int set = 0);   // set: which of the ctx's two selection work areas
int vo_int_ransac_work_prepare(vo_ctx* ctx, int H, vo_ransac_work* w);
