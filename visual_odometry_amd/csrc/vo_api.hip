// vo_api.hip -- the C ABI of libvo_hip.so (include/vo_hip.h): context, host-pointer entry points
// (marshal -> device -> launch -> copy back) and device-pointer entry points.
#include <stdlib.h>
#include "vo_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <vector>

// scratch buffer purposes
enum { SB_PU = 0, SB_PV, SB_QU, SB_QV, SB_VALID, SB_SCORE, SB_IMG, SB_A, SB_B, SB_C, SB_D, SB_E, SB_F, SB_G, SB_H, SB_I,
       SB_RGB = 16, SB_A2 = 17,
       SB_RESP = 18 };   // response / greedy entry points: never a selection work area (those are guarded by sel_gen only)

// Selections begun and not yet finished (vo_dev_select_begin / vo_dev_frame_prefetch .. vo_dev_select_finish):
// a queue of at most two, each with its own work buffers (set 0: scratch[SB_A], set 1: scratch[SB_A2]) and its
// own join event, finished oldest first.
struct vo_sel_pending {
    struct Entry {
        int nu = 0, nv = 0, max_exist = 0;
        vo_sel_opts o;
        vo_sel_work w;
        hipEvent_t join = nullptr;  // recorded on the side stream behind the detect half
        int set = 0; uint64_t gen = 0;   // work area and its hand-out number (vo_ctx::sel_gen)
    } q[2];
    int head = 0, count = 0;        // q[head] is the oldest open entry
};

const char* vo_strerror(int code) {
    switch (code) {
        case VO_OK: return "ok";
        case VO_ERR_NO_DEVICE: return "no HIP device available (there is no CPU fallback)";
        case VO_ERR_BAD_ARG: return "bad argument";
        case VO_ERR_HIP: return "HIP runtime error";
        case VO_ERR_OOM: return "out of device memory";
        case VO_ERR_UNSUPPORTED: return "unsupported configuration";
        case VO_END: return "end of the sequence";
        default: return "unknown error";
    }
}
int vo_version(void) { return 100; }

// all streams a ctx enqueues on have drained: the ctx stream (the lane itself while a lane is open), the stream the
// lane was opened from, the side stream and the lane
static hipError_t vo_int_quiesce(vo_ctx* ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ctx->in_lane) e = hipStreamSynchronize(ctx->lane_saved);
    if (e == hipSuccess && ctx->side) e = hipStreamSynchronize(ctx->side);
    if (e == hipSuccess && ctx->lane) e = hipStreamSynchronize(ctx->lane);
    return e;
}

// ---- who holds streams of the library on a device (one process): see vo_ctx_lane_priority ----
namespace {
constexpr int VO_MAX_DEVICES = 64;
struct DeviceRegistry { int ctx_ever = 0, pools_ever = 0; vo_ctx* hp_owner = nullptr; };
std::mutex g_reg_mu;
DeviceRegistry g_reg[VO_MAX_DEVICES];
}
void vo_int_register_streams(int device, vo_ctx* new_ctx_or_null) {
    if (device < 0 || device >= VO_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    DeviceRegistry& r = g_reg[device];
    if (new_ctx_or_null) ++r.ctx_ever; else ++r.pools_ever;
    // whoever held a high-priority lane is no longer alone: it gives the priority up before the lane's next use
    if (r.hp_owner && r.hp_owner != new_ctx_or_null) r.hp_owner->lane_demote.store(true, std::memory_order_release);
    if (r.hp_owner && !new_ctx_or_null) r.hp_owner->lane_demote.store(true, std::memory_order_release);   // a pool of its own too
}

int vo_buf_reserve(vo_ctx* ctx, vo_buf* b, size_t bytes) {
    if (bytes <= b->cap) return VO_OK;
    if (b->p) {
        VO_HIP_CHECK(ctx, vo_int_quiesce(ctx));   // every stream of the ctx: work issued ahead or on the lane may still read the old block
        VO_HIP_CHECK(ctx, hipFree(b->p));
        b->p = nullptr; b->cap = 0;
    }
    size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
    VO_HIP_CHECK(ctx, hipMalloc(&b->p, cap));
    b->cap = cap;
    return VO_OK;
}

int vo_ctx_create(int device, vo_ctx** out) {
    if (!out) return VO_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VO_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return VO_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return VO_ERR_HIP;
    vo_ctx* c = new (std::nothrow) vo_ctx();
    if (!c) return VO_ERR_OOM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void**)&c->d_counters, VO_COUNTER_SHARDS * 4 * sizeof(unsigned long long) + 768) != hipSuccess   /* + vo_stats_gather's staging (256 B) + the pose step's ticket (256 B) + the head words of a tracking launch (256 B) */ ||
        hipMemset(c->d_counters, 0, VO_COUNTER_SHARDS * 4 * sizeof(unsigned long long) + 768) != hipSuccess) {
        vo_ctx_destroy(c);
        return VO_ERR_HIP;
    }
    c->stream = c->own_stream;
    vo_int_register_streams(device, c);
    if (vo_static_scratch_resolve(device) != VO_OK) {   // vo_dev_p3p_ransac_static's work array on this GPU
        vo_ctx_destroy(c);
        return VO_ERR_HIP;
    }
    *out = c;
    return VO_OK;
}

void vo_ctx_destroy(vo_ctx* c) {
    if (!c) return;
    if (c->device >= 0 && c->device < VO_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (g_reg[c->device].hp_owner == c) g_reg[c->device].hp_owner = nullptr;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    for (auto& f : c->frames) {
        if (f.I) (void)hipFree(f.I);
        if (f.img) (void)hipFree(f.img);
        if (f.ready) (void)hipEventDestroy(f.ready);
    }
    for (auto& b : c->scratch) if (b.p) (void)hipFree(b.p);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->lap_ev) (void)hipEventDestroy(e);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->lane) {
        (void)hipStreamSynchronize(c->lane);
        (void)hipStreamDestroy(c->lane);
    }
    if (c->ev_lane_guard) (void)hipEventDestroy(c->ev_lane_guard);
    if (c->ev_lane_fork) (void)hipEventDestroy(c->ev_lane_fork);
    if (c->ev_lane_done) (void)hipEventDestroy(c->ev_lane_done);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->sel_pending)
        for (auto& e : c->sel_pending->q) if (e.join) (void)hipEventDestroy(e.join);
    delete c->sel_pending;
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int vo_ctx_set_stream(vo_ctx* ctx, void* s) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->in_lane) { ctx->last_error = "vo_ctx_set_stream: inside vo_ctx_lane_begin / _end"; return VO_ERR_BAD_ARG; }
    if (ctx->lane) VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane));
    ctx->lane_pending = false;
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->side));
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return VO_OK;
}
void* vo_ctx_get_stream(vo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int vo_ctx_sync(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->side));   // work begun ahead (prefetch, detect half)
    if (ctx->lane) VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane));   // a branch begun with vo_ctx_lane_begin
    if (ctx->in_lane) VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane_saved));
    return VO_OK;
}

// ---- lane: a branch of the frame's dependency graph on a second stream ----
// Events that only order the ctx's streams on one device (never read by the host) are created without the system-scope
// fence a recorded event performs by default -- a cache write-back towards the host that the next kernel on another
// stream does not need (kernel boundaries release and acquire at device scope by themselves) and that cost ~1 us per
// event in the frame loop's chain (0.179 -> 0.174 ms per frame, 0.948 -> 0.941 ms per 50k-keypoint step).  Events the
// host waits on (a frame's record, the ingest slots) keep the default.  VO_EVENT_FLAGS=<flags> overrides (A/B).
unsigned vo_int_event_flags() {
    static const unsigned f = [] {
        const char* e = getenv("VO_EVENT_FLAGS");
        return e ? (unsigned)strtoul(e, nullptr, 0) : (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    }();
    return f;
}

int vo_int_lane_ensure(vo_ctx* ctx) {
    const bool demote = ctx->lane_demote.load(std::memory_order_acquire) || getenv("VO_LANE_NORMAL_PRIORITY");
    if (ctx->lane && ctx->lane_is_high && demote && !ctx->in_lane) {
        // The ctx is no longer alone on its GPU (another ctx or a decode pool was created since the priority was
        // granted): the high-priority stream is drained and replaced by one at the ctx stream's priority.  Events
        // recorded on the old stream stay valid (all of them have completed by now).
        VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->lane));
        VO_HIP_CHECK(ctx, hipStreamDestroy(ctx->lane));
        ctx->lane = nullptr; ctx->lane_is_high = false; ctx->lane_high_priority = false;
        VO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->lane, hipStreamNonBlocking));
    }
    if (!ctx->lane) {
        int lo = 0, hi = 0;   // numerically lower = higher priority
        VO_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        // The lane has the ctx stream's priority unless the caller asked for more AND was granted it
        // (vo_ctx_lane_priority: a ctx that is alone on its GPU).  A high-priority stream beside more hardware queues
        // than the process may hold (GPU_MAX_HW_QUEUES = 4: a second ctx's streams, even destroyed ones; a decode
        // pool's copy stream; another process on the GPU) waited for ever for an event of a stream that no longer got
        // to run: the frame loop stood still in exactly that order of tests (DESIGN "Schedule of a step").
        // VO_LANE_NORMAL_PRIORITY=1 overrules the request (A/B switch).
        if (!ctx->lane_high_priority || demote) {
            VO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->lane, hipStreamNonBlocking));
            ctx->lane_is_high = false;
        } else {
            VO_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->lane, hipStreamNonBlocking, hi));
            ctx->lane_is_high = true;
        }
        if (!ctx->ev_lane_fork) VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_lane_fork, vo_int_event_flags()));
        if (!ctx->ev_lane_done) VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_lane_done, vo_int_event_flags()));
    }
    return VO_OK;
}
// An entry point called OUTSIDE the lane that uses a buffer the lane's work may still be using (the pose step's work
// arrays: the frame loop's pose chain runs on the lane, asynchronously) first makes the ctx stream wait for whatever
// the lane has been given so far.  Inside the lane, or without one, nothing is enqueued.
int vo_int_wait_for_lane(vo_ctx* ctx) {
    if (!ctx->lane || ctx->in_lane) return VO_OK;
    if (!ctx->ev_lane_guard) VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_lane_guard, vo_int_event_flags()));
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_lane_guard, ctx->lane));
    VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lane_guard, 0));
    return VO_OK;
}
int vo_ctx_lane_begin(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->in_lane) { ctx->last_error = "vo_ctx_lane_begin: a lane is already open"; return VO_ERR_BAD_ARG; }
    (void)hipSetDevice(ctx->device);
    { int rc = vo_int_lane_ensure(ctx); if (rc) return rc; }
    if (ctx->lane_pending) {   // an earlier branch nobody joined: the new one comes behind it anyway (same stream)
        ctx->lane_pending = false;
    }
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_lane_fork, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->lane, ctx->ev_lane_fork, 0));
    ctx->lane_saved = ctx->stream;
    ctx->stream = ctx->lane;
    ctx->in_lane = true;
    return VO_OK;
}
int vo_ctx_lane_priority(vo_ctx* ctx, int high) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->lane) { ctx->last_error = "vo_ctx_lane_priority: the lane exists already (call this before its first use)"; return VO_ERR_BAD_ARG; }
    if (ctx->device < 0 || ctx->device >= VO_MAX_DEVICES) return VO_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    DeviceRegistry& r = g_reg[ctx->device];
    if (!high) {
        ctx->lane_high_priority = false;
        if (r.hp_owner == ctx) r.hp_owner = nullptr;
        return VO_OK;
    }
    // granted only to a ctx that is, and always was, alone on its device in this process: every vo_ctx ever created
    // there (destroyed ones too: their hardware queues were part of the order that hung) and every decode pool counts
    if (r.ctx_ever > 1 || r.pools_ever > 0) {
        ctx->last_error = "vo_ctx_lane_priority: refused -- this process has held another vo_ctx or a decode pool on this GPU "
                          "(a high-priority lane beside more hardware queues than the process may hold can wait for ever); "
                          "the lane keeps the ctx stream's priority";
        return VO_ERR_UNSUPPORTED;
    }
    ctx->lane_high_priority = true;
    ctx->lane_demote.store(false, std::memory_order_release);
    r.hp_owner = ctx;
    return VO_OK;
}
int vo_ctx_lane_resume(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->in_lane || !ctx->lane) { ctx->last_error = "vo_ctx_lane_resume: no lane to resume (or one is open)"; return VO_ERR_BAD_ARG; }
    { int rc = vo_int_lane_ensure(ctx); if (rc) return rc; }   // (a lane that has to give its priority up does so here)
    ctx->lane_saved = ctx->stream;
    ctx->stream = ctx->lane;
    ctx->in_lane = true;
    return VO_OK;
}
int vo_ctx_lane_end(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (!ctx->in_lane) { ctx->last_error = "vo_ctx_lane_end: no lane is open"; return VO_ERR_BAD_ARG; }
    ctx->stream = ctx->lane_saved;
    ctx->in_lane = false;
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_lane_done, ctx->lane));
    ctx->lane_pending = true;
    return VO_OK;
}
int vo_ctx_lane_join(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->in_lane) { ctx->last_error = "vo_ctx_lane_join: the lane is still open"; return VO_ERR_BAD_ARG; }
    if (ctx->lane_pending) {
        VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lane_done, 0));
        ctx->lane_pending = false;
    }
    return VO_OK;
}
const char* vo_ctx_last_error(vo_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }

int vo_ctx_read_counters(vo_ctx* ctx, uint64_t out[4], int reset) {
    if (!ctx || !out) return VO_ERR_BAD_ARG;
    std::vector<unsigned long long> h(VO_COUNTER_SHARDS * 4);
    const size_t bytes = h.size() * sizeof(unsigned long long);
    VO_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), ctx->d_counters, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) VO_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, bytes, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k) out[k] = 0;
    for (int s = 0; s < VO_COUNTER_SHARDS; ++s)
        for (int k = 0; k < 4; ++k) out[k] += h[4 * s + k];
    return VO_OK;
}
// ---- statistics collective (RCCL, resolved at call time) ----
int vo_stats_gather(vo_ctx* ctx, void* rccl_comm, const vo_stats* local, vo_stats* job) {
    if (!ctx || !local || !job) return VO_ERR_BAD_ARG;
    *job = *local;
    if (!rccl_comm) return VO_OK;   // one rank
    // ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    // ncclFloat64 = 8, ncclSum = 0, ncclMax = 2 (rccl.h)
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static const allreduce_fn fn = [] {   // resolved once (thread-safe static initialisation)
        allreduce_fn f = (allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");       // the RCCL the host process already holds
        if (!f) {
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) f = (allreduce_fn)dlsym(h, "ncclAllReduce");
        }
        return f;
    }();
    if (!fn) { ctx->last_error = "vo_stats_gather: no RCCL in the process and librccl.so cannot be loaded"; return VO_ERR_UNSUPPORTED; }
    (void)hipSetDevice(ctx->device);
    double* d = (double*)ctx->d_counters;   // not the counters themselves: a 160-byte staging area behind them
    d += VO_COUNTER_SHARDS * 4;
    double h[5] = {local->frames, local->keypoints_tracked, local->keypoints_valid, local->ransac_inliers, local->elapsed_s};
    VO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    if (fn(d, d + 5, 5, 8, 0, rccl_comm, ctx->stream) != 0 || fn(d, d + 10, 5, 8, 2, rccl_comm, ctx->stream) != 0) {
        ctx->last_error = "vo_stats_gather: ncclAllReduce failed";
        return VO_ERR_HIP;
    }
    double r[10];
    VO_HIP_CHECK(ctx, hipMemcpyAsync(r, d + 5, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    job->frames = r[0]; job->keypoints_tracked = r[1]; job->keypoints_valid = r[2]; job->ransac_inliers = r[3];
    job->elapsed_s = r[9];
    return VO_OK;
}

int vo_timer_start(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return VO_OK;
}
int vo_timer_stop_ms(vo_ctx* ctx, float* ms) {
    if (!ctx || !ms) return VO_ERR_BAD_ARG;
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    VO_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    VO_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VO_OK;
}

// measurement of the pose step's scoring kernel (bench.py): while on, every k_score launch is a lap of the stopwatch below
// and adds the pose x point tests it executes to one word; vo_ctx_read_score_tests returns (and clears) the count
int vo_ctx_time_score(vo_ctx* ctx, int on) {
    if (!ctx) return VO_ERR_BAD_ARG;
    ctx->time_score = on != 0;
    return VO_OK;
}
int vo_ctx_read_score_tests(vo_ctx* ctx, uint64_t* tests, int reset) {
    if (!ctx || !tests) return VO_ERR_BAD_ARG;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(vo_int_head_words(ctx)) + 64);
    unsigned long long h = 0;
    VO_HIP_CHECK(ctx, hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) VO_HIP_CHECK(ctx, hipMemsetAsync(d, 0, sizeof h, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *tests = h;
    return VO_OK;
}

// laps: many intervals on the ctx stream, read back with ONE synchronisation (the host never idles the GPU
// between the intervals, so the kernels inside see the clocks of a saturated device)
int vo_timer_lap_begin(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (ctx->lap_open) { ctx->last_error = "vo_timer_lap_begin: the previous lap was not ended"; return VO_ERR_BAD_ARG; }
    while ((int)ctx->lap_ev.size() < 2 * (ctx->laps + 1)) {
        hipEvent_t e = nullptr;
        VO_HIP_CHECK(ctx, hipEventCreate(&e));
        ctx->lap_ev.push_back(e);
    }
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->lap_ev[2 * ctx->laps], ctx->stream));
    ctx->lap_open = true;
    return VO_OK;
}
int vo_timer_lap_end(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    if (!ctx->lap_open) { ctx->last_error = "vo_timer_lap_end: no lap was begun"; return VO_ERR_BAD_ARG; }
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->lap_ev[2 * ctx->laps + 1], ctx->stream));
    ctx->lap_open = false;
    ctx->laps += 1;
    return VO_OK;
}
int vo_timer_laps_ms(vo_ctx* ctx, float* total_ms, int* laps) {
    if (!ctx || !total_ms || !laps || ctx->lap_open) return VO_ERR_BAD_ARG;
    float tot = 0.0f;
    for (int i = 0; i < ctx->laps; ++i) {
        float ms = 0.0f;
        VO_HIP_CHECK(ctx, hipEventSynchronize(ctx->lap_ev[2 * i + 1]));
        VO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->lap_ev[2 * i], ctx->lap_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *laps = ctx->laps;
    ctx->laps = 0;
    return VO_OK;
}

// ---- frame slots ----
static int frame_prepare(vo_ctx* ctx, vo_pyr* f, int nu, int nv, int levels) {
    if (nu < 8 || nv < 8 || levels < 1 || levels > VO_MAX_LEVELS) { ctx->last_error = "frame: bad shape or level count"; return VO_ERR_BAD_ARG; }
    long tot = 0;
    int a = nu, b = nv;
    for (int l = 0; l < levels; ++l) {
        f->nus[l] = a; f->nvs[l] = b; f->offs[l] = tot;
        tot += (long)(a + 2 * VO_PYR_PAD) * (b + 2 * VO_PYR_PAD);
        a = (a + 1) / 2; b = (b + 1) / 2;
    }
    if (f->nus[levels - 1] < 2 || f->nvs[levels - 1] < 2) { ctx->last_error = "frame: too many pyramid levels for this image"; return VO_ERR_BAD_ARG; }
    if ((size_t)tot > f->cap) {
        VO_HIP_CHECK(ctx, vo_int_quiesce(ctx));
        if (f->I) (void)hipFree(f->I);
        f->I = nullptr; f->cap = 0;
        VO_HIP_CHECK(ctx, hipMalloc((void**)&f->I, tot * sizeof(float)));
        f->cap = (size_t)tot;
    }
    f->nu = nu; f->nv = nv; f->levels = levels; f->total = tot;
    return VO_OK;
}

// the ctx stream waits, once, for a pyramid that vo_dev_frame_prefetch is building on the side stream
static int frame_await(vo_ctx* ctx, vo_pyr* f) {
    if (f->ready_pending) {
        VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, f->ready, 0));
        f->ready_pending = false;
    }
    return VO_OK;
}

int vo_dev_frame_wait(vo_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= VO_NUM_FRAME_SLOTS) return VO_ERR_BAD_ARG;
    if (!ctx->frames[slot].valid) { ctx->last_error = "vo_dev_frame_wait: frame slot not built"; return VO_ERR_BAD_ARG; }
    return frame_await(ctx, &ctx->frames[slot]);
}

int vo_dev_frame_build(vo_ctx* ctx, int slot, const uint8_t* d_img, int nu, int nv, int levels) {
    if (!ctx || !d_img || slot < 0 || slot >= VO_NUM_FRAME_SLOTS) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    vo_pyr* f = &ctx->frames[slot];
    f->valid = false;
    int rc = frame_await(ctx, f);   // a prefetch into this slot still in flight: this build comes after it
    if (rc) return rc;
    rc = frame_prepare(ctx, f, nu, nv, levels);
    if (rc) return rc;
    rc = vo_launch_frame_build(ctx, f, d_img);
    if (rc) return rc;
    f->valid = true; f->key = 0; f->stamp = ++ctx->clock;
    return VO_OK;
}

static int select_begin_on_side(vo_ctx* ctx, const uint8_t* d_img, int nu, int nv, const vo_sel_opts* o, int max_exist);
static int check_sel_opts(vo_ctx* ctx, const vo_sel_opts* o);

int vo_dev_frame_prefetch(vo_ctx* ctx, int slot, const uint8_t* d_img, int nu, int nv, int levels, const vo_sel_opts* o,
                          int max_exist) {
    if (!ctx || !d_img || slot < 0 || slot >= VO_NUM_FRAME_SLOTS || max_exist < 0) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (o) {
        int rc = check_sel_opts(ctx, o);
        if (rc) return rc;
        if (ctx->sel_pending && ctx->sel_pending->count == 2) {
            ctx->last_error = "vo_dev_frame_prefetch: two selections are already open";
            return VO_ERR_BAD_ARG;
        }
    }
    vo_pyr* f = &ctx->frames[slot];
    f->valid = false;
    int rc = frame_prepare(ctx, f, nu, nv, levels);   // may grow the slot (synchronises the ctx stream when it does)
    if (rc) return rc;
    if (!f->ready) VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&f->ready, vo_int_event_flags()));
    if ((rc = vo_int_side_fork(ctx))) return rc;      // behind every reader and writer of the slot enqueued so far
    if ((rc = vo_launch_frame_build(ctx, f, d_img, ctx->side))) return rc;
    VO_HIP_CHECK(ctx, hipEventRecord(f->ready, ctx->side));
    f->ready_pending = true;
    f->valid = true; f->key = 0; f->stamp = ++ctx->clock;
    return o ? select_begin_on_side(ctx, d_img, nu, nv, o, max_exist) : VO_OK;
}

int vo_dev_klt_track(vo_ctx* ctx, int sa, int sb, const double* d_pu, const double* d_pv, int n, int wu, int wv,
                     int max_iters, double eps, double max_bidir, double* d_qu, double* d_qv, uint8_t* d_valid,
                     float* d_score, float* d_fb) {
    if (!ctx || sa < 0 || sa >= VO_NUM_FRAME_SLOTS || sb < 0 || sb >= VO_NUM_FRAME_SLOTS) return VO_ERR_BAD_ARG;
    if (n == 0) return VO_OK;
    if (n < 0 || !d_pu || !d_pv || !d_qu || !d_qv || !d_valid) return VO_ERR_BAD_ARG;   // d_score may be NULL: scores not wanted
    if (!ctx->frames[sa].valid || !ctx->frames[sb].valid) { ctx->last_error = "klt: frame slot not built"; return VO_ERR_BAD_ARG; }
    if (eps <= 0) eps = 0.01;
    (void)hipSetDevice(ctx->device);
    int rc = frame_await(ctx, &ctx->frames[sa]);
    if (!rc) rc = frame_await(ctx, &ctx->frames[sb]);
    if (rc) return rc;
    return vo_launch_klt(ctx, &ctx->frames[sa], &ctx->frames[sb], d_pu, d_pv, n, wu, wv, max_iters, eps, max_bidir,
                         d_qu, d_qv, d_valid, d_score, d_fb);
}

int vo_dev_klt_track_head(vo_ctx* ctx, int sa, int sb, const double* d_pu, const double* d_pv, int n_head, int n, int wu, int wv,
                          int max_iters, double eps, double max_bidir, double* d_qu, double* d_qv, uint8_t* d_valid,
                          float* d_score, float* d_fb) {
    if (!ctx || sa < 0 || sa >= VO_NUM_FRAME_SLOTS || sb < 0 || sb >= VO_NUM_FRAME_SLOTS) return VO_ERR_BAD_ARG;
    if (n < 0 || n_head < 0 || n_head > n) return VO_ERR_BAD_ARG;
    if (ctx->in_lane) { ctx->last_error = "vo_dev_klt_track_head: call it outside the lane (the lane is where the head's consumers go)"; return VO_ERR_BAD_ARG; }
    if (n == 0) return VO_OK;
    if (!d_pu || !d_pv || !d_qu || !d_qv || !d_valid) return VO_ERR_BAD_ARG;
    if (!ctx->frames[sa].valid || !ctx->frames[sb].valid) { ctx->last_error = "klt: frame slot not built"; return VO_ERR_BAD_ARG; }
    if (eps <= 0) eps = 0.01;
    (void)hipSetDevice(ctx->device);
    int rc = frame_await(ctx, &ctx->frames[sa]);
    if (!rc) rc = frame_await(ctx, &ctx->frames[sb]);
    if (rc) return rc;
    rc = vo_launch_klt(ctx, &ctx->frames[sa], &ctx->frames[sb], d_pu, d_pv, n, wu, wv, max_iters, eps, max_bidir, d_qu, d_qv,
                       d_valid, d_score, d_fb, nullptr, n_head);
    if (!rc) ctx->head_open = n_head > 0;
    return rc;
}

int vo_dev_wait_head(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_BAD_ARG;
    // Only behind a head that HAS been enqueued: a waiting kernel in front of its launch in a shared hardware queue
    // would sit there until its time-out.  (The launch itself is on another stream: the order of the two calls is the
    // order of the packets wherever the two streams share a queue.)
    if (!ctx->head_open) { ctx->last_error = "vo_dev_wait_head: no vo_dev_klt_track_head launch to wait for"; return VO_ERR_BAD_ARG; }
    (void)hipSetDevice(ctx->device);
    return vo_launch_wait_head(ctx);
}

int vo_dev_head_timed_out(vo_ctx* ctx, int* timed_out) {
    if (!ctx || !timed_out) return VO_ERR_BAD_ARG;
    unsigned w = 0;
    VO_HIP_CHECK(ctx, hipMemcpyAsync(&w, reinterpret_cast<unsigned*>(vo_int_head_words(ctx)) + 2, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *timed_out = (int)w;
    return VO_OK;
}

// content hash for the host-API pyramid cache (the reference calls track twice per frame on the
// same image pair, modules/VisualOdometry.m:71,278, and re-uses curr as next frame's prev)
static uint64_t hash_image(const uint8_t* p, size_t n, int nu, int nv, int levels) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)nu << 40) ^ ((uint64_t)nv << 16) ^ (uint64_t)levels;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001B3ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    return h ? h : 1;
}

// returns the slot holding the pyramid of this host image (building it if needed)
static int frame_from_host(vo_ctx* ctx, const uint8_t* img, int nu, int nv, int levels, int avoid_slot, int* slot_out) {
    const size_t nbytes = (size_t)nu * nv;
    const uint64_t key = hash_image(img, nbytes, nu, nv, levels);
    int lru = -1;
    for (int s = 0; s < VO_NUM_FRAME_SLOTS; ++s) {
        vo_pyr* f = &ctx->frames[s];
        if (f->valid && f->key == key && f->nu == nu && f->nv == nv && f->levels == levels) {
            f->stamp = ++ctx->clock;
            *slot_out = s;
            return VO_OK;
        }
        if (s == avoid_slot) continue;
        if (lru < 0 || !f->valid || (ctx->frames[lru].valid && f->stamp < ctx->frames[lru].stamp)) lru = s;
    }
    vo_pyr* f = &ctx->frames[lru];
    f->valid = false;
    { int rc = frame_await(ctx, f); if (rc) return rc; }   // a prefetch into this slot may still be running
    if (nbytes > f->img_cap) {
        VO_HIP_CHECK(ctx, vo_int_quiesce(ctx));
        if (f->img) (void)hipFree(f->img);
        f->img = nullptr; f->img_cap = 0;
        VO_HIP_CHECK(ctx, hipMalloc((void**)&f->img, nbytes));
        f->img_cap = nbytes;
    }
    VO_HIP_CHECK(ctx, hipMemcpyAsync(f->img, img, nbytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = frame_prepare(ctx, f, nu, nv, levels);
    if (rc) return rc;
    rc = vo_launch_frame_build(ctx, f, f->img);
    if (rc) return rc;
    f->valid = true; f->key = key; f->stamp = ++ctx->clock;
    *slot_out = lru;
    return VO_OK;
}

template <typename T>
static int dbuf(vo_ctx* ctx, int which, size_t count, T** out) {
    int rc = vo_buf_reserve(ctx, &ctx->scratch[which], count * sizeof(T));
    *out = (T*)ctx->scratch[which].p;
    return rc;
}

int vo_klt_track(vo_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int H, int W, vo_layout lay,
                 const void* pts_xy, int N, int pts_is_double, const vo_klt_opts* o, void* out_pts_xy,
                 uint8_t* out_valid, void* out_scores) {
    if (!ctx || !o) return VO_ERR_BAD_ARG;
    if (N == 0) return VO_OK; // modules/KLTTracker.m:33-36: outputs stay the caller's typed empties
    if (N < 0 || !img1 || !img2 || !pts_xy || !out_pts_xy || !out_valid || H <= 0 || W <= 0) return VO_ERR_BAD_ARG;   // out_scores may be NULL
    (void)hipSetDevice(ctx->device);
    const int nu = lay.col_major ? H : W, nv = lay.col_major ? W : H;
    const int wu = lay.col_major ? o->block_h : o->block_w, wv = lay.col_major ? o->block_w : o->block_h;
    int sa, sb, rc;
    if ((rc = frame_from_host(ctx, img1, nu, nv, o->levels, -1, &sa))) return rc;
    if ((rc = frame_from_host(ctx, img2, nu, nv, o->levels, sa, &sb))) return rc;
    // marshal points to internal (u,v) zero-based SoA doubles
    std::vector<double> h(2 * (size_t)N);
    const double base = (double)lay.index_base;
    for (int i = 0; i < N; ++i) {
        double x, y;
        const size_t ix = lay.col_major ? (size_t)i : 2 * (size_t)i, iy = lay.col_major ? (size_t)N + i : 2 * (size_t)i + 1;
        if (pts_is_double) { x = ((const double*)pts_xy)[ix]; y = ((const double*)pts_xy)[iy]; }
        else { x = ((const float*)pts_xy)[ix]; y = ((const float*)pts_xy)[iy]; }
        x -= base; y -= base;
        h[i] = lay.col_major ? y : x;
        h[(size_t)N + i] = lay.col_major ? x : y;
    }
    double *d_p, *d_q; uint8_t* d_valid; float* d_score;
    if ((rc = dbuf(ctx, SB_PU, 2 * (size_t)N, &d_p))) return rc;
    if ((rc = dbuf(ctx, SB_QU, 2 * (size_t)N, &d_q))) return rc;
    if ((rc = dbuf(ctx, SB_VALID, (size_t)N, &d_valid))) return rc;
    if ((rc = dbuf(ctx, SB_SCORE, (size_t)N, &d_score))) return rc;
    if (!out_scores) d_score = nullptr;   // the caller discards the scores (modules/VisualOdometry.m:71,278 do): they are not computed
    VO_HIP_CHECK(ctx, hipMemcpyAsync(d_p, h.data(), 2 * (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const double eps = o->eps > 0 ? o->eps : 0.01;
    rc = vo_launch_klt(ctx, &ctx->frames[sa], &ctx->frames[sb], d_p, d_p + N, N, wu, wv, o->max_iters, eps,
                       o->max_bidir_err, d_q, d_q + N, d_valid, d_score, nullptr);
    if (rc) return rc;
    std::vector<float> hs((size_t)N);
    VO_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), d_q, 2 * (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipMemcpyAsync(out_valid, d_valid, (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    if (d_score) VO_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), d_score, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < N; ++i) {
        const double u = h[i] + base, v = h[(size_t)N + i] + base;
        const double x = lay.col_major ? v : u, y = lay.col_major ? u : v;
        const size_t ix = lay.col_major ? (size_t)i : 2 * (size_t)i, iy = lay.col_major ? (size_t)N + i : 2 * (size_t)i + 1;
        if (pts_is_double) {
            ((double*)out_pts_xy)[ix] = x; ((double*)out_pts_xy)[iy] = y;
            if (out_scores) ((double*)out_scores)[i] = (double)hs[i];
        } else {
            ((float*)out_pts_xy)[ix] = (float)x; ((float*)out_pts_xy)[iy] = (float)y;
            if (out_scores) ((float*)out_scores)[i] = hs[i];
        }
    }
    return VO_OK;
}

#include "vo_api_rest.inc"

int vo_int_side_ensure(vo_ctx* ctx) {
    if (!ctx->side) {
        VO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, vo_int_event_flags()));
        VO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, vo_int_event_flags()));
    }
    return VO_OK;
}
int vo_int_side_fork(vo_ctx* ctx) {
    int rc = vo_int_side_ensure(ctx);
    if (rc) return rc;
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    return VO_OK;
}
int vo_int_side_join(vo_ctx* ctx) {
    VO_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, ctx->side));
    VO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return VO_OK;
}

int vo_int_frame_prepare(vo_ctx* ctx, vo_pyr* f, int nu, int nv, int levels) { return frame_prepare(ctx, f, nu, nv, levels); }
int vo_int_sel_work_prepare(vo_ctx* ctx, int nu, int nv, int keep, int n_exist, double min_distance, vo_sel_work* w, int set) {
    return sel_work_prepare(ctx, nu, nv, keep, n_exist, min_distance, w, set);
}
int vo_int_ransac_work_prepare(vo_ctx* ctx, int H, vo_ransac_work* w) { return ransac_work_prepare(ctx, H, w); }
