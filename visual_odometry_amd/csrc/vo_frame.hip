// vo_frame.hip -- device-resident frame loop: VisualOdometry.processFrame
// (modules/VisualOdometry.m:264-346) with the method candidateTriangulation (:63-151).
// The MATLAB state struct lives in HBM as structure-of-arrays; the five steps of a frame run on the
// ctx stream through the same launchers as the stand-alone entry points (vo_launch_klt, _ransac,
// _reproj, _triangulate, _select), glued by the small ordered-compaction / merge kernels below.
// Per frame the host reads back one small record at the end (counts + pose + status); every count -- the state's
// own sizes included -- stays on the device and its consumers are launched for the capacities and read it there, so a
// frame can be enqueued before the one in front of it has finished (vo_frame_submit / vo_frame_collect); no point
// data crosses PCIe.
// Normative restatement: oracle/vo_frame.py.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "vo_common.h"
#include "vo_dev_math.h"

namespace {

// exclusive position of every set flag among the block's 1024 threads, and their total
__device__ __forceinline__ int block_excl_scan_1024(int flag, int* total, int* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                       // s_w may still be read from the previous call
    if (lane == 0) s_w[w] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = s_w[k];
        if (k < w) base += c;
        tot += c;
    }
    *total = tot;
    return base + within;
}

// ordered compaction by one block: Op::load(i) reads element i (what decides and what moves, in one batch of loads:
// a flag read first and the sources read behind a branch are two memory latencies in a row), Op::flag(regs) says
// whether it stays, Op::store(d, regs) puts it at position d of the output; *count = number kept.  The block walks the
// n elements 1024 at a time.
template <class Op>
__device__ __forceinline__ void compact_block(const Op& op, int n, int32_t* count, const int32_t* n_dev, int* s_w) {
    if (n_dev) n = min(n, *n_dev);   // the count lives on the device, n is its upper bound
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const typename Op::Regs r = op.load(i < n ? i : 0);
        const int f = ((i < n) & op.flag(r)) ? 1 : 0;
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) op.store(base + pos, r);
        base += tot;
    }
    if (threadIdx.x == 0 && count) *count = base;
}
// two independent compactions in one launch: block 0 does A, block 1 does B
template <class OpA, class OpB>
// (count_b_clamped: the same count cut at cap_b -- the number of landmarks the next frame starts from)
__global__ __launch_bounds__(1024) void k_compact2(OpA a, int na, int32_t* count_a, const int32_t* na_dev, OpB b, int nb, int32_t* count_b,
                                                   int32_t* count_b_clamped, int cap_b) {
    __shared__ int s_w[16];
    if (blockIdx.x == 0) { compact_block(a, na, count_a, na_dev, s_w); return; }
    compact_block(b, nb, count_b, nullptr, s_w);
    __syncthreads();
    if (threadIdx.x == 0) *count_b_clamped = *count_b < cap_b ? *count_b : cap_b;
}

// ---- the same ordered compactions for states of tens of thousands of entries (the stress size: 25k landmarks, 25k
// candidates), where one block walking 1024 elements a step is 25-50 steps of two barriers each (k_tracked 105-146 us,
// k_pose_tail 70, k_compact2 84 in the trace of a 25k + 25k frame) and a 1024-thread block beside a tracking launch
// waits for a compute unit to itself.  One-wave workgroups of CW_BLOCK elements: a wave counts its kept elements row by
// row (ballots), PUBLISHES the count in a word of its own (launch epoch << 32 | count, an agent-scope atomic store:
// written through), adds up the counts of the workgroups in front of it (every lane polls some of them -- nobody waits
// for a sum, only for counts that their owners publish before they wait for anything themselves, and workgroups are
// dispatched in index order: the ones in front are resident or done) and moves its elements.  The last workgroup
// writes the total.  The epoch (a host counter, one per launch) tells this launch's words from an earlier one's, so the
// words are never cleared; a poll gives up after CW_SPINS looks, so the grid drains whatever happens.
constexpr int CW_EPL = 8, CW_BLOCK = 64 * CW_EPL, CW_MAX_BLOCKS = 256, CW_SITES = 8, CW_SPINS = 1 << 20;
// fault: a word in pinned host memory that a poll which gives up raises (vo_frame_collect reports it: the frame's counts
// are then not to be trusted); skip: the workgroup that does NOT publish, -1 in every real launch (fault injection of
// tests/test_gpu_frame.py, VO_FRAME_FAULT_INJECT)
struct Chain { unsigned long long* w; unsigned epoch; int32_t* fault; int skip; };
__device__ __forceinline__ int wave_sum_i(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}
__device__ __forceinline__ int chain_exclusive(const Chain& c, int b, int tot) {
    const int lane = threadIdx.x;
    if (lane == 0 && b != c.skip)
        __hip_atomic_store(&c.w[b], ((unsigned long long)c.epoch << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int sum = 0;
    for (int j = lane; j < b; j += 64) {
        unsigned long long v = 0;
        for (int spin = 0; spin < CW_SPINS; ++spin) {
            v = __hip_atomic_load(&c.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == c.epoch) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if ((unsigned)(v >> 32) != c.epoch) {   // gave up: the count is unknown (taken as 0: every position stays inside the arrays)
            v = 0;
            if (c.fault) __hip_atomic_store(c.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        sum += (int)(unsigned)v;
    }
    return wave_sum_i(sum);
}
// rows of 64 consecutive elements: element i0 + 64 k + lane; keep(k) says whether the lane's element of row k stays.
// Returns the lane's output positions (pos[k], valid where keep) through `pos`, the workgroup's count in *tot and its
// exclusive base.
template <class Keep>
__device__ __forceinline__ int wave_positions(const Chain& ch, Keep keep, int (&pos)[CW_EPL], unsigned& fbits, int* tot) {
    int run = 0;
    fbits = 0;
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {
        const bool f = keep(k);
        const unsigned long long bal = __ballot(f);
        pos[k] = run + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        run += __popcll(bal);
        fbits |= (f ? 1u : 0u) << k;
    }
    *tot = run;
    return chain_exclusive(ch, blockIdx.x, run);
}
template <class Op>
__device__ __forceinline__ void compact_wave(const Op& op, int n, const Chain& ch, int32_t* count, int32_t* count_clamped = nullptr, int cap = 0) {
    const int i0 = blockIdx.x * CW_BLOCK + (int)threadIdx.x;
    // every row's loads first, in one batch (rows beyond n read element 0 and are dropped): a wave that reads a row's
    // flag, branches, reads the row's sources, stores and goes on to the next row waits for memory sixteen times in a
    // row -- 15-25 us for a launch whose work is nothing
    typename Op::Regs r[CW_EPL];
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) { const int i = i0 + 64 * k; r[k] = op.load(i < n ? i : 0); }
    int pos[CW_EPL], tot;
    unsigned fbits;
    const int base = wave_positions(ch, [&](int k) { return (i0 + 64 * k < n) & op.flag(r[k]); }, pos, fbits, &tot);
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k)
        if ((fbits >> k) & 1u) op.store(base + pos[k], r[k]);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (count) *count = base + tot;
        if (count_clamped) *count_clamped = base + tot < cap ? base + tot : cap;
    }
}
inline int cw_blocks(int n) { return (n + CW_BLOCK - 1) / CW_BLOCK; }
// two independent compactions in one launch: blockIdx.y = 0 does A, 1 does B (k_compact2 for large states)
template <class OpA, class OpB>
__global__ __launch_bounds__(64) void k_compact2_w(OpA a, int na, int32_t* count_a, const int32_t* na_dev, Chain cha, OpB b, int nb,
                                                   int32_t* count_b, int32_t* count_b_clamped, int cap_b, Chain chb) {
    if (blockIdx.y == 0) compact_wave(a, na_dev ? min(na, *na_dev) : na, cha, count_a);
    else compact_wave(b, nb, chb, count_b, count_b_clamped, cap_b);
}

// landmarks whose keypoint tracked validly (:280-281): (q, L) -> (vk, vl), pitch = capL
struct TrackedOp {
    const uint8_t* valid; const double *qu, *qv, *L; int capL; double *vk, *vl;
    struct Regs { uint8_t ok; double u, v, x, y, z; };
    __device__ Regs load(int i) const { return {valid[i], qu[i], qv[i], L[i], L[capL + i], L[2 * capL + i]}; }
    __device__ bool flag(const Regs& r) const { return r.ok != 0; }
    __device__ void store(int d, const Regs& r) const {
        vk[d] = r.u; vk[capL + d] = r.v;
        vl[d] = r.x; vl[capL + d] = r.y; vl[2 * capL + d] = r.z;
    }
};
// :302 strict <, :308-315 survivors into the merge work arrays (pitch PW), alive = 1
struct KeptOp {
    const double* err; double thr; const double *vk, *vl; int capL, PW; double *wkp, *wL, *wrep; uint8_t* alive;
    // (the pose tail has the element in registers when it decides: keypoint, landmark, error)
    __device__ void store(int d, double ku, double kv, double x, double y, double z, double e) const {
        wkp[d] = ku; wkp[PW + d] = kv;
        wL[d] = x; wL[PW + d] = y; wL[2 * PW + d] = z;
        wrep[d] = e;
        alive[d] = 1;
    }
};
// The end of the pose step and what follows it on the tracked landmarks, in one block (three launches before: the
// pose step's k_inliers, k_reproj, the ordered compaction): the pose [R_WC; t_WC] of the record (:318), the inlier
// flags and their number (the arithmetic of k_inliers, vo_ransac.hip), the reprojection error of EVERY tracked
// landmark (:298-300, the arithmetic of k_reproj, vo_geom.hip), the strict filter (:302) and the survivors in order
// into the merge arrays (:308-315).  The device functions are the same source in all three files (vo_dev_math.h).
struct PoseTailArgs {
    const double *vk, *vl; int capL; const int32_t* M_dev; const double* K; double thr2;
    double* rsout; uint8_t* inl; double* pose_now; double* err; KeptOp kop; int32_t* count;
};
__global__ __launch_bounds__(1024) void k_pose_tail(PoseTailArgs a) {
    __shared__ int s_w[16];
    __shared__ double s_pose[12], s_P[12];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    int32_t* info = (int32_t*)(a.rsout + 12);
    const int Md = min(a.capL, *a.M_dev);
    if (tid < 12) {
        const double* out = a.rsout;
        const int c = tid - 9;
        const double p = tid < 9 ? out[tid] : -((out[c] * out[9] + out[3 + c] * out[10]) + out[6 + c] * out[11]);
        a.pose_now[tid] = p;
        s_pose[tid] = p;
        s_P[tid] = proj_elem(out, a.K, tid >> 2, tid & 3);
    }
    __syncthreads();
    const bool solved = info[0] == 0;
    double P[12], K[9], cam[12];
    for (int k = 0; k < 12; ++k) P[k] = s_P[k];
    for (int k = 0; k < 9; ++k) K[k] = a.K[k];
    camera_matrix(s_pose, K, cam);
    int base = 0, n_in = 0;
    for (int i0 = 0; i0 < a.capL; i0 += 1024) {
        const int i = i0 + tid;
        bool in = false;
        int f = 0;
        double Xi[3] = {0.0, 0.0, 0.0}, x = 0.0, y = 0.0, e = 0.0;
        if (i < Md) {
            Xi[0] = a.vl[i]; Xi[1] = a.vl[a.capL + i]; Xi[2] = a.vl[2 * a.capL + i];
            x = a.vk[i]; y = a.vk[a.capL + i];
            if (solved) {
                const ProjT p = proj_terms(P, Xi[0], Xi[1], Xi[2], x, y);
                in = p.ez < a.thr2 * p.z2;
            }
            double u, v, w;
            project(cam, Xi, u, v, w);
            const double du = x - u, dv = y - v;
            e = sqrt(du * du + dv * dv);
            a.err[i] = e;
            f = e < a.kop.thr ? 1 : 0;
        }
        if (i < a.capL) a.inl[i] = in ? 1 : 0;
        const unsigned long long bal = __ballot(in);
        if ((tid & 63) == 0) n_in += __popcll(bal);
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) a.kop.store(base + pos, x, y, Xi[0], Xi[1], Xi[2], e);
        base += tot;
    }
    if ((tid & 63) == 0 && n_in) atomicAdd(&info[4], n_in);
    if (tid == 0) *a.count = base;
}

// the same for large states: one-wave workgroups (every one folds the record's pose itself: twelve values)
__global__ __launch_bounds__(64) void k_pose_tail_w(PoseTailArgs a, Chain ch) {
    __shared__ double s_pose[12], s_P[12];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    int32_t* info = (int32_t*)(a.rsout + 12);
    const int Md = min(a.capL, *a.M_dev);
    if (tid < 12) {
        const double* out = a.rsout;
        const int c = tid - 9;
        const double p = tid < 9 ? out[tid] : -((out[c] * out[9] + out[3 + c] * out[10]) + out[6 + c] * out[11]);
        if (blockIdx.x == 0) a.pose_now[tid] = p;
        s_pose[tid] = p;
        s_P[tid] = proj_elem(out, a.K, tid >> 2, tid & 3);
    }
    __syncthreads();
    const bool solved = info[0] == 0;
    double P[12], K[9], cam[12];
    for (int k = 0; k < 12; ++k) P[k] = s_P[k];
    for (int k = 0; k < 9; ++k) K[k] = a.K[k];
    camera_matrix(s_pose, K, cam);
    const int i0 = blockIdx.x * CW_BLOCK + tid;
    // all rows' loads in one batch (rows beyond the tracked landmarks read element 0 and are dropped), then the arithmetic
    double X0[CW_EPL], X1[CW_EPL], X2[CW_EPL], kx[CW_EPL], ky[CW_EPL], er[CW_EPL];
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {
        const int i = i0 + 64 * k, ic = i < Md ? i : 0;
        X0[k] = a.vl[ic]; X1[k] = a.vl[a.capL + ic]; X2[k] = a.vl[2 * a.capL + ic];
        kx[k] = a.vk[ic]; ky[k] = a.vk[a.capL + ic];
    }
    int n_in = 0;
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {
        const int i = i0 + 64 * k;
        const double Xi[3] = {X0[k], X1[k], X2[k]};
        bool in = false;
        if (solved) {
            const ProjT p = proj_terms(P, Xi[0], Xi[1], Xi[2], kx[k], ky[k]);
            in = (i < Md) & (p.ez < a.thr2 * p.z2);
        }
        double u, v, w;
        project(cam, Xi, u, v, w);
        const double du = kx[k] - u, dv = ky[k] - v;
        er[k] = sqrt(du * du + dv * dv);
        if (i < Md) a.err[i] = er[k];
        if (i < a.capL) a.inl[i] = in ? 1 : 0;
        n_in += __popcll(__ballot(in));
    }
    int pos[CW_EPL], tot;
    unsigned fbits;
    const int base = wave_positions(ch, [&](int k) { return (i0 + 64 * k < Md) & (er[k] < a.kop.thr); }, pos, fbits, &tot);
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k)
        if ((fbits >> k) & 1u) a.kop.store(base + pos[k], kx[k], ky[k], X0[k], X1[k], X2[k], er[k]);
    if (tid == 0 && n_in) atomicAdd(&info[4], n_in);
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *a.count = base + tot;
}

// :106-113, :130: 0 drop, 1 promote (angle large), 2 keep as candidate (angle small, young enough)
struct ClassifyArgs {
    const uint8_t *tracked, *ok; const double *rep, *ang; const int32_t* ctime; double max_rep, ang_thr; int recall;
    struct Regs { uint8_t t, o; double rep, ang; int32_t ct; };
    __device__ Regs load(int i) const { return {tracked[i], ok[i], rep[i], ang[i], ctime[i]}; }
    // (no branch between the loads: five reads, one wait)
    __device__ int cls_of(const Regs& r) const {
        const bool good = (r.t != 0) & (r.o != 0) & (r.rep <= max_rep);
        return good ? (r.ang > ang_thr ? 1 : (r.ct > -recall ? 2 : 0)) : 0;
    }
    __device__ int cls(int i) const { return cls_of(load(i)); }
};
// :130-139 kept candidates, aged by one frame
struct CandOp {
    ClassifyArgs c; int capC; const double *cq, *cf; const int32_t *cpose, *ctime; double *ncu, *ncf; int32_t *ncpose, *nctime;
    struct Regs { ClassifyArgs::Regs c; double qu, qv, fu, fv; int32_t pose; };
    __device__ Regs load(int i) const { return {c.load(i), cq[i], cq[capC + i], cf[i], cf[capC + i], cpose[i]}; }   // (c.ctime is ctime)
    __device__ bool flag(const Regs& r) const { return c.cls_of(r.c) == 2; }
    __device__ void store(int d, const Regs& r) const {
        ncu[d] = r.qu; ncu[capC + d] = r.qv;
        ncf[d] = r.fu; ncf[capC + d] = r.fv;
        ncpose[d] = r.pose; nctime[d] = r.c.ct - 1;
    }
};
// the live landmarks, in order, into the next state (pitch capL)
struct LandmarkOp {
    const uint8_t* alive; int PW, capL; const double *wkp, *wL, *wrep; double *kp, *L, *rep;
    struct Regs { uint8_t alive; double ku, kv, x, y, z, e; };
    __device__ Regs load(int i) const { return {alive[i], wkp[i], wkp[PW + i], wL[i], wL[PW + i], wL[2 * PW + i], wrep[i]}; }
    __device__ bool flag(const Regs& r) const { return r.alive != 0; }
    __device__ void store(int d, const Regs& r) const {
        if (d >= capL) return;
        kp[d] = r.ku; kp[capL + d] = r.kv;
        L[d] = r.x; L[capL + d] = r.y; L[2 * capL + d] = r.z;
        rep[d] = r.e;
    }
};
// start of a frame: the frame's counters and the tombstone array are cleared.  The two track calls of a frame (:278
// landmarks, :71 candidates) read their points straight from the state arrays and are two launches -- the landmarks'
// on the ctx's lane stream in front of the pose chain, the candidates' on the ctx stream (vo_frame_submit).
// ... in the launch that keeps the validly tracked landmarks in order (:280-281): one block, clear first
__global__ __launch_bounds__(1024) void k_tracked(int32_t* counts, uint8_t* alive, int pw, TrackedOp top, int capL, const int32_t* nL_dev) {
    __shared__ int s_w[16];
    if (threadIdx.x < 16) counts[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < pw; i += 1024) alive[i] = 0;
    __syncthreads();                                     // counts[6] is cleared before the compaction writes it
    compact_block(top, capL, counts + 6, nL_dev, s_w);
}

// the same for large states: one-wave workgroups; every counter but counts[6] is cleared by the first workgroup, counts[6]
// is written by the last (no word has two writers)
__global__ __launch_bounds__(64) void k_tracked_w(int32_t* counts, uint8_t* alive, int pw, TrackedOp top, int capL, const int32_t* nL_dev, Chain ch) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < 16 && tid != 6) counts[tid] = 0;
    // tombstones: four bytes a lane, pw is a multiple of ... anything: the tail byte by byte
    for (int i = 4 * ((int)blockIdx.x * 64 + tid); i < pw; i += 4 * 64 * (int)gridDim.x) {
        if (i + 3 < pw) *reinterpret_cast<uint32_t*>(alive + i) = 0u;
        else for (int k = i; k < pw; ++k) alive[k] = 0;
    }
    compact_wave(top, min(capL, *nL_dev), ch, counts + 6);
}

// The order-dependent merge of :87-142 over the promotable candidates, in candidate order: at the landmark
// cap a candidate replaces the FIRST maximum of reproError if its own error is not larger (:116-122, maximum
// taken before the edit), below the cap it is appended (:123-127).  Landmarks are never moved: an evicted
// one is tombstoned, an accepted one appended; the ordered compaction afterwards gives the MATLAB order.
struct MergeArgs {
    const int32_t* nC_dev;          // the number of candidates (device)
    int cap, PW, capC;
    const double *X, *crep, *cq;    // candidates: pitch capC
    double *wkp, *wL, *wrep; uint8_t* alive;
    const int32_t* n0;              // landmarks that survived the reprojection filter
    int32_t* plist;                 // scratch, nC
    int32_t* n_promoted;
    ClassifyArgs c;                 // which candidates are promotable (class 1)
};
constexpr int MERGE_LDS = 4096;       // pool entries (landmarks + promotable candidates) of the one-block form
constexpr int MERGE_ONE_BLOCK = 1280; // landmarks + candidates up to which the single block is the fastest form

// No sequential loop: at the cap every accepted candidate evicts the current
// first maximum, so the landmark set is a streaming "cap smallest so far": the maximum never grows, an
// element below the final maximum T is never evicted, one above T never survives, and among elements equal
// to T the ones latest in order survive (eviction takes the first maximum; accepted candidates are appended
// behind everything).  Order = work-array order: surviving landmarks, then candidates in candidate order.
//   accepted(j)  <=>  #{x before j in order : v_x < r_j} < cap            (what :116-127 decides at j's turn)
//   survives(e)  <=>  #{x : v_x < v_e} + #{x : v_x = v_e, x after e} < cap
// Both are all-pairs counts over the pool (landmarks + promotable candidates): one block, every thread
// owns elements and walks the pool in LDS (broadcast reads).  The equivalence with the sequential policy is
// checked on tie-heavy random pools (tests/test_host_logic.py) and per frame against the oracle's sequential
// restatement (tests/test_gpu_frame.py).
// ... and, in the same launch, what k_compact2 does behind the three-launch form: block 1 keeps the aged candidates
// (it needs the classes only, not the merge), block 0 goes on to the final landmark order once its merge is written
// (one launch less in the chain every frame runs through).
__global__ __launch_bounds__(1024) void k_merge_par(MergeArgs a, CandOp cop, int nc, int32_t* count_c, LandmarkOp lop, int32_t* count_l,
                                                    int32_t* sizes) {
    __shared__ double s_v[MERGE_LDS];
    __shared__ int32_t s_pidx[MERGE_LDS];
    __shared__ uint8_t s_acc[MERGE_LDS], s_surv[MERGE_LDS];
    __shared__ int s_w[16];
    if (blockIdx.x == 1) { compact_block(cop, nc, count_c, sizes + 1, s_w); return; }
    const int tid = threadIdx.x;
    const int n0 = *a.n0, nC = *a.nC_dev;
    // ordered list of the promotable candidates
    int np = 0;
    for (int i0 = 0; i0 < nC; i0 += 1024) {
        const int i = i0 + tid;
        const int f = (i < nC && a.c.cls(i) == 1) ? 1 : 0;
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) s_pidx[np + pos] = i;
        np += tot;
    }
    __syncthreads();
    const int P = n0 + np;                 // <= PW <= MERGE_LDS (checked by the host)
    for (int e = tid; e < P; e += 1024) s_v[e] = e < n0 ? a.wrep[e] : a.crep[s_pidx[e - n0]];
    __syncthreads();
    // 16 lanes per element, 64 elements per round: a pool of a few hundred entries keeps the whole block busy
    // (one thread per element left most of it idle while the busy threads walked the whole pool)
    const int slice = tid & 15, eg = tid >> 4;
    for (int e0 = 0; e0 < P; e0 += 64) {
        const int e = e0 + eg;
        const bool live = e < P;
        const double ve = live ? s_v[e] : 0.0;
        // before = #{x < e : v_x < v_e};  rank = before + #{x > e : v_x <= v_e}: one compare and one add per pair
        int before = 0, after = 0;
        // (four independent LDS reads per step: one read per step made the loop a chain of LDS latencies)
        if (live) {
            int x = slice;
            for (; x + 48 < e; x += 64) {
                const double v0 = s_v[x], v1 = s_v[x + 16], v2 = s_v[x + 32], v3 = s_v[x + 48];
                before += (int)(v0 < ve) + (int)(v1 < ve) + (int)(v2 < ve) + (int)(v3 < ve);
            }
            for (; x < e; x += 16) before += s_v[x] < ve;
            if (x == e) x += 16;
            for (; x + 48 < P; x += 64) {
                const double v0 = s_v[x], v1 = s_v[x + 16], v2 = s_v[x + 32], v3 = s_v[x + 48];
                after += (int)(v0 <= ve) + (int)(v1 <= ve) + (int)(v2 <= ve) + (int)(v3 <= ve);
            }
            for (; x < P; x += 16) after += s_v[x] <= ve;
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { before += __shfl_xor(before, d, 64); after += __shfl_xor(after, d, 64); }
        const int rank = before + after;
        if (live && slice == 0) {
            s_surv[e] = rank < a.cap;
            s_acc[e] = before < a.cap;     // meaningful for candidates (e >= n0)
        }
    }
    __syncthreads();
    // accepted candidates, in order, become work-array entries n0, n0 + 1, ...
    int prom = 0;
    for (int j0 = 0; j0 < np; j0 += 1024) {
        const int j = j0 + tid;
        const int f = (j < np && s_acc[n0 + j]) ? 1 : 0;
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) {
            const int i = s_pidx[j], d = n0 + prom + pos;
            a.wL[d] = a.X[i]; a.wL[a.PW + d] = a.X[a.capC + i]; a.wL[2 * a.PW + d] = a.X[2 * a.capC + i];
            a.wkp[d] = a.cq[i]; a.wkp[a.PW + d] = a.cq[a.capC + i];
            a.wrep[d] = s_v[n0 + j];
            a.alive[d] = s_surv[n0 + j];
        }
        prom += tot;
    }
    for (int i = tid; i < n0; i += 1024) a.alive[i] = s_surv[i];
    if (tid == 0) *a.n_promoted = prom;
    __syncthreads();                       // the block's own stores (alive, the appended entries) are visible to all of it
    compact_block(lop, a.PW, count_l, nullptr, s_w);
    __syncthreads();
    if (tid == 0) sizes[0] = *count_l < a.cap ? *count_l : a.cap;
}

// The same for pools beyond a few hundred entries, where one block walking all pairs is too slow -- and without the
// all-pairs walk at all (round 3; the first wide form spread the O(pool^2) counts over the chip): both counts come out
// of ONE stable merge sort of the pool by reproError, O(n log n):
//   before(e) = #{x earlier in order : v_x < v_e}: in a bottom-up merge sort over the pool in order, the elements earlier
//               than e are exactly the left-hand runs e's run is merged with, level by level; e's rank among the
//               strictly smaller keys of that left run (a binary search the merge needs anyway) is that level's share;
//   rank(e)   = #{x : v_x < v_e} + #{x : v_x = v_e, x later}: e's place in the sorted sequence with the ties of equal
//               keys reversed -- its sorted position mirrored inside its group of equal keys.
// Launches: the list and the pool, runs of SORT_TILE elements sorted in LDS (one block per run), then -- the runs are
// never merged into one sequence, the two counts are all the policy needs -- ONE launch in which every element looks
// itself up in every run (k_sort_decide: up to SORT_WAYS binary searches side by side instead of log2(runs) dependent
// launches of one search each; 40 us of merge levels + 9 us of ranks -> one launch at the stress size).  Pools of more
// than SORT_WAYS runs first double their run length (k_sort_merge) until SORT_WAYS runs are left: O(n log n) either way.
// Then the ordered write.  Keys beyond the pool are +inf (never below or equal to a real key: reproError is finite
// wherever the classes let a candidate in).
// pool: the keys in pool order; key / idx: two buffers each of the padded length; before: one int per pool entry.
struct MergeWideArgs { MergeArgs m; double* pool; uint8_t *surv, *acc; int32_t* np_out; double* key[2]; int32_t* idx[2]; int32_t* before; int padded; };
constexpr int SORT_TILE = 2048;     // elements sorted by one block in LDS (2 per thread: every step of a level is a chain of dependent LDS reads)
constexpr int SORT_WAYS = 32;       // runs that k_sort_decide looks through per element (a choice of speed: any number is right)
// the run length the decisions are taken at, for a pool of (about) P entries
static inline int sort_decide_run(long P) {
    int L = SORT_TILE;
    while ((P + L - 1) / L > SORT_WAYS) L <<= 1;
    return L;
}

__global__ __launch_bounds__(1024) void k_merge_list(MergeWideArgs w) {
    __shared__ int s_w[16];
    const MergeArgs& a = w.m;
    const int tid = threadIdx.x;
    const int n0 = *a.n0, nC = *a.nC_dev;
    int np = 0;
    for (int i0 = 0; i0 < nC; i0 += 1024) {
        const int i = i0 + tid;
        const int f = (i < nC && a.c.cls(i) == 1) ? 1 : 0;
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) { a.plist[np + pos] = i; w.pool[n0 + np + pos] = a.crep[i]; }
        np += tot;
    }
    for (int e = tid; e < n0; e += 1024) w.pool[e] = a.wrep[e];
    if (tid == 0) *w.np_out = np;
}

// first index in [lo, hi) whose key is >= v (STRICT = false) / > v (STRICT = true); keys ascending
template <bool STRICT, class Ptr>
__device__ __forceinline__ int bound(Ptr key, int lo, int hi, double v) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double k = key[mid];
        if (STRICT ? (k <= v) : (k < v)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(1024) void k_sort_tiles(MergeWideArgs w) {
    __shared__ double s_key[2][SORT_TILE];
    __shared__ int32_t s_idx[2][SORT_TILE];
    __shared__ int32_t s_before[SORT_TILE];
    const int P = *w.m.n0 + *w.np_out;
    const int t0 = blockIdx.x * SORT_TILE;
    if (t0 >= P) {      // a tile of padding (the buffers are sized for the full state): in order as it is
        for (int k = threadIdx.x; k < SORT_TILE; k += 1024) {
            w.key[0][t0 + k] = __builtin_huge_val(); w.idx[0][t0 + k] = t0 + k; w.before[t0 + k] = 0;
        }
        return;
    }
    for (int k = threadIdx.x; k < SORT_TILE; k += 1024) {
        const int e = t0 + k;
        s_key[0][k] = e < P ? w.pool[e] : __builtin_huge_val();
        s_idx[0][k] = e;
        s_before[k] = 0;
    }
    __syncthreads();
    // A level: every element counts the keys of its partner run that are < its own -- a search of log2(L) + 1 steps
    // whatever the element, so a wave does not diverge and the two elements of a thread are searched side by side (a
    // level is a chain of dependent LDS reads: this is what it costs).  For a left-hand element that count is its
    // shift; a right-hand element is shifted by the keys <= its own, which differ from the count only where the
    // partner holds an equal key (a second search, taken by the lanes that meet one).
    int cur = 0;
    for (int L = 1; L < SORT_TILE; L <<= 1) {
        const double* key = s_key[cur];
        double v[2]; int pb[2], lb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = threadIdx.x + 1024 * q;
            v[q] = key[k];
            pb[q] = (k & ~(2 * L - 1)) + ((k & L) ? 0 : L);       // the partner run's first element
            lb[q] = 0;
        }
        for (int s2 = L >> 1; s2 >= 1; s2 >>= 1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) lb[q] += key[pb[q] + lb[q] + s2 - 1] < v[q] ? s2 : 0;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) lb[q] += key[pb[q] + lb[q]] < v[q] ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = threadIdx.x + 1024 * q;
            const int i = s_idx[cur][k];
            const int base = k & ~(2 * L - 1);
            int shift = lb[q];
            if (k & L) {                                           // right-hand run: behind the left keys that are <= v
                if (lb[q] < L && key[pb[q] + lb[q]] <= v[q]) shift = bound<true>(key, pb[q] + lb[q], pb[q] + L, v[q]) - pb[q];
                if (lb[q]) s_before[i - t0] += lb[q];              // every element has one thread per level: no two writers
            }
            const int d = base + (k & (L - 1)) + shift;
            s_key[cur ^ 1][d] = v[q]; s_idx[cur ^ 1][d] = i;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int k = threadIdx.x; k < SORT_TILE; k += 1024) {
        w.key[0][t0 + k] = s_key[cur][k]; w.idx[0][t0 + k] = s_idx[cur][k];
        w.before[t0 + k] = s_before[k];
    }
}

// one doubling of the run length in global memory: buffer `from` -> the other one
__global__ __launch_bounds__(256) void k_sort_merge(MergeWideArgs w, int L, int from) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= w.padded) return;
    const double* key = w.key[from];
    const double v = key[p];
    const int i = w.idx[from][p];
    int sm;
    int d;
    const int run = p / L, base = (run >> 1) * 2 * L;
    if (base + L >= w.padded) { d = p; sm = 0; }     // a last run without a partner moves as it is
    else {
        // the partner may be shorter than L (the padded length is a multiple of SORT_TILE, not of 2 L)
        const int rend = min(base + 2 * L, w.padded);
        if (run & 1) {
            const int le = bound<true>(key, base, base + L, v) - base;
            sm = bound<false>(key, base, base + le, v) - base;
            d = base + (p - (base + L)) + le;
        } else {
            sm = 0;
            d = base + (p - base) + (bound<false>(key, base + L, rend, v) - (base + L));
        }
    }
    w.key[from ^ 1][d] = v; w.idx[from ^ 1][d] = i;
    if (sm) w.before[i] += sm;
}

// The two decisions of the merge from the sorted runs of length L (buffer `in`; L = SORT_TILE << levels merged so far).
// One lane per sorted position; with lo_r / hi_r = the keys of run r that are < / <= the lane's own key v:
//   before(e) = before inside e's run (counted while it was sorted) + sum of lo_r over the EARLIER runs (runs are
//               consecutive stretches of the pool, so an earlier run holds earlier entries only);
//   rank(e)   = sum of lo_r over all runs + the equal keys that come later in the pool: hi_r - lo_r of the later runs
//               and, the sort being stable, the equal keys behind e in its own run.
// A search is a fixed number of steps (L is a power of two), SORT_SIDE runs searched side by side: the loads of a step
// are independent (the launch costs the latency of ONE search), and the lanes of a wave -- neighbours in their run --
// walk nearly the same path through every other run.  One-wave workgroups (the launch may run beside a tracking launch).
constexpr int SORT_SIDE = 16;       // runs searched side by side
template <bool LE>
__device__ __forceinline__ void search_runs(const double* key, long base0, int L, int padded, double v, int (&out)[SORT_SIDE]) {
    const double inf = __builtin_huge_val();
#pragma unroll
    for (int q = 0; q < SORT_SIDE; ++q) out[q] = 0;
    for (int s = L >> 1; s >= 1; s >>= 1) {
#pragma unroll
        for (int q = 0; q < SORT_SIDE; ++q) {
            const long i = base0 + (long)q * L + out[q] + s - 1;
            const double k = i < padded ? key[i] : inf;
            out[q] += (LE ? k <= v : k < v) ? s : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < SORT_SIDE; ++q) {           // the steps above cover L - 1 keys: the last one
        const long i = base0 + (long)q * L + out[q];
        const double k = i < padded ? key[i] : inf;
        out[q] += (LE ? k <= v : k < v) ? 1 : 0;
    }
}
__global__ __launch_bounds__(64) void k_sort_decide(MergeWideArgs w, int L, int in) {
    const int P = *w.m.n0 + *w.np_out;
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= w.padded) return;
    const int e = w.idx[in][p];
    if (e >= P) return;                      // padding
    const double* key = w.key[in];
    const double v = key[p];
    const int own = p / L, nruns = (P + L - 1) / L;    // (a wave lies inside one run: L is a multiple of 64)
    const int pl = p - own * L;
    int smaller = 0, smaller_earlier = 0, equal_later = 0;
    for (int r0 = 0; r0 < nruns; r0 += SORT_SIDE) {
        int lo[SORT_SIDE], hi[SORT_SIDE];
        search_runs<false>(key, (long)r0 * L, L, w.padded, v, lo);
        // the <= search only where another run holds an equal key (rare with measured errors); the lane's own run
        // holds one -- the lane's -- and its equal keys are its neighbours
        bool any_equal = false;
#pragma unroll
        for (int q = 0; q < SORT_SIDE; ++q) {
            const long i = (long)(r0 + q) * L + lo[q];
            hi[q] = lo[q];
            any_equal |= r0 + q != own && lo[q] < L && i < w.padded && key[i] <= v;
        }
        if (r0 <= own && own < r0 + SORT_SIDE) any_equal |= pl + 1 < L && p + 1 < w.padded && key[p + 1] <= v;
        if (any_equal) search_runs<true>(key, (long)r0 * L, L, w.padded, v, hi);
#pragma unroll
        for (int q = 0; q < SORT_SIDE; ++q) {
            const int r = r0 + q;
            if (r >= nruns) continue;
            smaller += lo[q];
            if (r < own) smaller_earlier += lo[q];
            else if (r > own) equal_later += hi[q] - lo[q];
            else equal_later += any_equal ? hi[q] - 1 - pl : 0;     // (no equal key behind the lane's: hi = pl + 1)
        }
    }
    w.surv[e] = smaller + equal_later < w.m.cap;
    w.acc[e] = w.before[e] + smaller_earlier < w.m.cap;
}

__global__ __launch_bounds__(1024) void k_merge_write(MergeWideArgs w) {
    __shared__ int s_w[16];
    const MergeArgs& a = w.m;
    const int tid = threadIdx.x;
    const int n0 = *a.n0, np = *w.np_out;
    int prom = 0;
    for (int j0 = 0; j0 < np; j0 += 1024) {
        const int j = j0 + tid;
        const int f = (j < np && w.acc[n0 + j]) ? 1 : 0;
        int tot;
        const int pos = block_excl_scan_1024(f, &tot, s_w);
        if (f) {
            const int i = a.plist[j], d = n0 + prom + pos;
            a.wL[d] = a.X[i]; a.wL[a.PW + d] = a.X[a.capC + i]; a.wL[2 * a.PW + d] = a.X[2 * a.capC + i];
            a.wkp[d] = a.cq[i]; a.wkp[a.PW + d] = a.cq[a.capC + i];
            a.wrep[d] = w.pool[n0 + j];
            a.alive[d] = w.surv[n0 + j];
        }
        prom += tot;
    }
    for (int i = tid; i < n0; i += 1024) a.alive[i] = w.surv[i];
    if (tid == 0) *a.n_promoted = prom;
}

// k_merge_list and k_merge_write for large states: one-wave workgroups (compact_wave's scheme)
__global__ __launch_bounds__(64) void k_merge_list_w(MergeWideArgs w, Chain ch) {
    const MergeArgs& a = w.m;
    const int tid = threadIdx.x;
    const int n0 = *a.n0, nC = *a.nC_dev;
    const int i0 = blockIdx.x * CW_BLOCK + tid;      // the launch covers max(landmarks, candidates): rows of both
    ClassifyArgs::Regs c[CW_EPL];
    double lrep[CW_EPL];
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {               // one batch of loads
        const int i = i0 + 64 * k;
        c[k] = a.c.load(i < nC ? i : 0);
        lrep[k] = a.wrep[i < n0 ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) { const int i = i0 + 64 * k; if (i < n0) w.pool[i] = lrep[k]; }
    int pos[CW_EPL], tot;
    unsigned fbits;
    const int base = wave_positions(ch, [&](int k) { return (i0 + 64 * k < nC) & (a.c.cls_of(c[k]) == 1); }, pos, fbits, &tot);
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k)
        if ((fbits >> k) & 1u) { const int d = base + pos[k]; a.plist[d] = i0 + 64 * k; w.pool[n0 + d] = c[k].rep; }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *w.np_out = base + tot;
}
__global__ __launch_bounds__(64) void k_merge_write_w(MergeWideArgs w, Chain ch) {
    const MergeArgs& a = w.m;
    const int tid = threadIdx.x;
    const int n0 = *a.n0, np = *w.np_out;
    const int j0 = blockIdx.x * CW_BLOCK + tid;
    uint8_t lsurv[CW_EPL], acc[CW_EPL], csurv[CW_EPL];
    int32_t ci[CW_EPL];
    double crep[CW_EPL], X0[CW_EPL], X1[CW_EPL], X2[CW_EPL], qu[CW_EPL], qv[CW_EPL];
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {               // first batch: the decisions and the list
        const int j = j0 + 64 * k, jc = j < np ? j : 0;
        lsurv[k] = w.surv[j < n0 ? j : 0];
        acc[k] = w.acc[n0 + jc]; csurv[k] = w.surv[n0 + jc]; crep[k] = w.pool[n0 + jc];
        ci[k] = a.plist[jc];
    }
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) {               // second batch: the listed candidates (a row beyond the list reads candidate 0)
        const int i = j0 + 64 * k < np ? ci[k] : 0;
        X0[k] = a.X[i]; X1[k] = a.X[a.capC + i]; X2[k] = a.X[2 * a.capC + i];
        qu[k] = a.cq[i]; qv[k] = a.cq[a.capC + i];
    }
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k) { const int j = j0 + 64 * k; if (j < n0) a.alive[j] = lsurv[k]; }
    int pos[CW_EPL], tot;
    unsigned fbits;
    const int base = wave_positions(ch, [&](int k) { return (j0 + 64 * k < np) & (acc[k] != 0); }, pos, fbits, &tot);
#pragma unroll
    for (int k = 0; k < CW_EPL; ++k)
        if ((fbits >> k) & 1u) {
            const int d = n0 + base + pos[k];
            a.wL[d] = X0[k]; a.wL[a.PW + d] = X1[k]; a.wL[2 * a.PW + d] = X2[k];
            a.wkp[d] = qu[k]; a.wkp[a.PW + d] = qv[k];
            a.wrep[d] = crep[k];
            a.alive[d] = csurv[k];
        }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *a.n_promoted = base + tot;
}

// :338-345 append the new candidates: first keypoint = keypoint, first pose = this frame's, time -1
// ... and, last kernel of a frame, gather what the host reads back into one record (one copy instead of three):
// the pose step's record (16 doubles), the pose (12), the counters (16 int32)
// ... and the sizes of the state the NEXT frame starts from
__global__ void k_append_new(int32_t* counts, int capL, int capC, const float* su, const float* sv, int pose_slot, double* cu,
                             double* cf, int32_t* cpose, int32_t* ctime, const double* rsout, const double* pose_now,
                             double* rec, int32_t* sizes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nC = counts[1];
    int m = counts[2];
    const int room = capC - nC;
    const int take = m < room ? m : room;
    if (i < take) {
        const int d = nC + i;
        const double u = (double)su[i], v = (double)sv[i];
        cu[d] = u; cu[capC + d] = v; cf[d] = u; cf[capC + d] = v;
        cpose[d] = pose_slot; ctime[d] = -1;
    }
    if (i == 0) {
        counts[3] = nC + take; counts[4] = m > room ? 1 : 0; counts[5] = take;
        sizes[1] = nC + take;   // (sizes[0], the landmarks, was written by k_compact2: the next frame's landmark branch starts there)
    }
    // the record, straight into the pinned host slot (36 lanes of the first block, one word each: no copy command
    // behind the frame's last kernel)
    if (i < 28) rec[i] = i < 16 ? rsout[i] : pose_now[i - 16];
    else if (i < 36) {
        int32_t pc[2];
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * (i - 28) + j;
            pc[j] = k == 3 ? nC + take : (k == 4 ? (m > room ? 1 : 0) : (k == 5 ? take : counts[k]));
        }
        int32_t* out = (int32_t*)(rec + i);
        out[0] = pc[0]; out[1] = pc[1];
    }
}

// diagnosis (VO_FRAME_STAMPS): the constant 100 MHz device clock at a point of a stream, written to pinned host memory
__global__ void k_stamp(long long* p) { *p = wall_clock64(); }
constexpr int STAMP_FRAMES = 4096, STAMP_N = 16;

} // namespace

// ------------------------------------------------------------------------------------------------
struct vo_frame_state {
    vo_frame_opts o;
    int nu, nv, capL, capC, PW, ring, P;   // P = ring + upload pose slots
    vo_pyr pyr[3];                          // previous image, the one being processed, the one announced after it
    uint8_t* img[3];
    int cur;                                // slot holding the previous image
    // look-ahead (vo_frame_announce): the announced image's copy, pyramid and detect half are issued on the side
    // stream during the vo_frame_process call before its own, behind that call's tracking launch
    const uint8_t* next_src = nullptr; int next_on_device = 0;    // announced, not issued yet
    const uint8_t* ahead_src = nullptr; int ahead_on_device = 0;  // issued into slot (cur + 1) % 3
    uint64_t ahead_gen = 0;                 // hand-out number of the selection work area it used (vo_ctx::sel_gen)
    hipEvent_t ev_ready = nullptr;          // side stream: copy + pyramid of the look-ahead slot
    hipEvent_t ev_fork_la = nullptr;        // ctx stream: in front of the frame's candidate track
    hipEvent_t ev_detect[2] = {nullptr, nullptr};  // side stream: detect half, per selection work set (frame parity)
    vo_sel_work selw[2];
    bool have_prev;
    int nL, nC, frame_no;
    char* arena;
    // state (S) and next state (N): landmarks 3 x capL, keypoints 2 x capL, repro capL; candidates
    double *L[2], *kp[2], *rep[2], *cu[2], *cf[2];
    int32_t *cpose[2], *ctime[2];
    int s;                                  // index of the current state
    double *poses, *K, *rsout2[2], *packed;   // packed: what a frame's host read-back takes (k_append_new); rsout2: by frame parity
    // work
    double *pin, *q, *vk, *vl, *err, *cq, *X, *crep, *cang, *wkp, *wL, *wrep;
    float *score, *su, *sv;
    uint8_t *valid, *cvalid, *cok, *alive, *inl;
    int32_t *pos, *plist, *counts2[2], *slin;   // counts2: a frame's counters, by frame parity
    double* skey; int32_t *sidx, *sbefore; int sort_padded;   // the wide merge's sort: two key / index buffers, the `before` counts
    int32_t* sizes;                         // device: {landmarks, candidates} of state s (k_append_new / set_state)
    unsigned long long* chain;              // the wide compactions' count words: CW_SITES call sites x CW_MAX_BLOCKS (zeroed once)
    unsigned chain_epoch = 0;               // one per wide compaction launched
    int32_t* h_fault = nullptr;             // pinned: raised by a device-side wait that gave up (Chain::fault, k_wait_head); read by vo_frame_collect
    bool fault_inject = false;              // VO_FRAME_FAULT_INJECT=1 (tests): one compaction workgroup of the next wide frame does not publish its count
    int32_t* h_counts;                      // pinned
    double* h_rec;                          // pinned: VO_FRAME_MAX_PENDING records of 40 doubles (ransac record 16, pose 12, counters)
    // host images go through a ring of pinned slots (one per pending frame + the look-ahead + one): the caller's buffer
    // is read during the call by a host memcpy, the upload from the slot is a truly asynchronous copy -- a pageable
    // source would make the runtime wait until the stream has reached the copy, behind the previous frame's chain
    static constexpr int PIN_SLOTS = VO_FRAME_MAX_PENDING + 2;
    uint8_t* h_img[PIN_SLOTS] = {};
    hipEvent_t ev_img[PIN_SLOTS] = {};      // the upload out of slot r is done (the slot may be written again)
    bool img_used[PIN_SLOTS] = {};
    int pin_next = 0;
    // frames submitted and not collected yet, oldest first
    // two branches of a frame (vo_frame_submit): the landmark branch on the ctx's lane stream, the rest on the ctx stream
    hipEvent_t ev_state = nullptr;          // ctx stream: the next state's landmarks are in place (k_compact2)
    hipEvent_t ev_pose = nullptr;           // lane: pose, reprojection filter and its compaction are done
    hipEvent_t ev_ltrack = nullptr;         // lane: the landmarks' tracking launch is done (large states: the candidates' starts behind it)
    bool state_event = false;               // ev_state stands for the current state (false after set_state / set_image)
    bool overlap = true;                    // VO_FRAME_SERIAL=1: both branches on the ctx stream
    bool narrow = false;                    // VO_FRAME_NARROW=1: one-block compactions at every size (A/B switch, read at create)
    bool tracks_together = false;           // VO_FRAME_TRACKS_TOGETHER=1: both tracking launches start together at every size
    int wide_min = 2048;                    // capacities beyond which the one-wave compactions run (VO_FRAME_WIDE_MIN: A/B switch)
    int one_block = MERGE_ONE_BLOCK;        // pools up to which the one-block merge runs (VO_FRAME_ONE_BLOCK: A/B switch)
    int l_first_min = 32768;                // keypoints of a frame from which the landmarks' launch goes first (VO_FRAME_L_FIRST_MIN)
    int l_head_pct = 85;                    // ... and the share of it behind which the candidates' launch starts (VO_FRAME_L_HEAD_PCT; 100: its end, an event)
    hipEvent_t ev_rec[VO_FRAME_MAX_PENDING] = {};
    long long* h_stamps = nullptr;          // VO_FRAME_STAMPS=<file>: device clock at 16 points of each frame (diagnosis only)
    std::string stamp_path;
    unsigned stamp_mask = 0xffffu;          // VO_FRAME_STAMP_MASK: which of the points (every stamp is a launch of its own)
    int pend_frame_no[VO_FRAME_MAX_PENDING] = {};
    int pend_head = 0, pend_count = 0;
};

#define ST_CHECK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e__); return e__ == hipErrorOutOfMemory ? VO_ERR_OOM : VO_ERR_HIP; } } while (0)

namespace {
struct Bump {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
void carve(vo_frame_state* st, char* base, size_t* total) {
    Bump b; b.base = base;
    const size_t cl = st->capL, cc = st->capC, pw = st->PW;
    for (int k = 0; k < 2; ++k) {
        st->L[k] = b.take<double>(3 * cl); st->kp[k] = b.take<double>(2 * cl); st->rep[k] = b.take<double>(cl);
        st->cu[k] = b.take<double>(2 * cc); st->cf[k] = b.take<double>(2 * cc);
        st->cpose[k] = b.take<int32_t>(cc); st->ctime[k] = b.take<int32_t>(cc);
    }
    st->poses = b.take<double>(12 * (size_t)st->P); st->K = b.take<double>(9); st->rsout2[0] = b.take<double>(16); st->rsout2[1] = b.take<double>(16);
    st->packed = b.take<double>(40);
    st->pin = b.take<double>(2 * pw);   /* scratch of the wide merge */ st->q = b.take<double>(2 * pw); st->vk = b.take<double>(2 * cl); st->vl = b.take<double>(3 * cl); st->err = b.take<double>(cl);
    st->cq = b.take<double>(2 * cc); st->X = b.take<double>(3 * cc); st->crep = b.take<double>(cc); st->cang = b.take<double>(cc);
    st->wkp = b.take<double>(2 * pw); st->wL = b.take<double>(3 * pw); st->wrep = b.take<double>(pw);
    st->score = b.take<float>(pw);
    const size_t keep = (size_t)std::max(st->o.sel.keep, 1);
    st->su = b.take<float>(keep); st->sv = b.take<float>(keep); st->slin = b.take<int32_t>(keep);
    st->valid = b.take<uint8_t>(pw); st->cvalid = b.take<uint8_t>(cc); st->cok = b.take<uint8_t>(cc);
    st->alive = b.take<uint8_t>(pw); st->inl = b.take<uint8_t>(cl);
    st->skey = b.take<double>(2 * (size_t)st->sort_padded); st->sidx = b.take<int32_t>(2 * (size_t)st->sort_padded);
    st->sbefore = b.take<int32_t>((size_t)st->sort_padded);
    st->pos = b.take<int32_t>(pw); st->plist = b.take<int32_t>(cc); st->counts2[0] = b.take<int32_t>(16); st->counts2[1] = b.take<int32_t>(16);
    st->sizes = b.take<int32_t>(16);
    st->chain = b.take<unsigned long long>((size_t)CW_SITES * CW_MAX_BLOCKS);
    *total = b.off + 256;
}
int copy_image(vo_ctx* ctx, vo_frame_state* st, int slot, const uint8_t* img, int on_device, hipStream_t stream = nullptr) {
    const size_t nb = (size_t)st->nu * st->nv;
    hipStream_t q = stream ? stream : ctx->stream;
    if (on_device) {
        ST_CHECK(hipMemcpyAsync(st->img[slot], img, nb, hipMemcpyDeviceToDevice, q));
        return VO_OK;
    }
    const int r = st->pin_next;
    st->pin_next = (r + 1) % vo_frame_state::PIN_SLOTS;
    if (!st->h_img[r]) {
        ST_CHECK(hipHostMalloc((void**)&st->h_img[r], nb, 0));
        ST_CHECK(hipEventCreateWithFlags(&st->ev_img[r], hipEventDisableTiming));
    }
    if (st->img_used[r]) ST_CHECK(hipEventSynchronize(st->ev_img[r]));   // PIN_SLOTS uploads ago: long done
    memcpy(st->h_img[r], img, nb);                                         // the caller's buffer is free when the call returns
    ST_CHECK(hipMemcpyAsync(st->img[slot], st->h_img[r], nb, hipMemcpyHostToDevice, q));
    ST_CHECK(hipEventRecord(st->ev_img[r], q));
    st->img_used[r] = true;
    return VO_OK;
}
int build_pyramid(vo_ctx* ctx, vo_frame_state* st, int slot, hipStream_t stream = nullptr) {
    int rc = vo_int_frame_prepare(ctx, &st->pyr[slot], st->nu, st->nv, st->o.klt.levels);
    if (rc) return rc;
    return vo_launch_frame_build(ctx, &st->pyr[slot], st->img[slot], stream);
}
// whatever was issued ahead is abandoned: the ctx stream waits for it (its buffers are about to be reused)
int drop_lookahead(vo_ctx* ctx, vo_frame_state* st) {
    if (st->ahead_src) {
        ST_CHECK(hipStreamWaitEvent(ctx->stream, st->ev_ready, 0));
        for (int k = 0; k < 2; ++k)
            if (st->ev_detect[k]) ST_CHECK(hipStreamWaitEvent(ctx->stream, st->ev_detect[k], 0));
    }
    st->ahead_src = nullptr; st->next_src = nullptr;
    return VO_OK;
}
int upload_image(vo_ctx* ctx, vo_frame_state* st, int slot, const uint8_t* img, int on_device) {
    int rc = copy_image(ctx, st, slot, img, on_device);
    return rc ? rc : build_pyramid(ctx, st, slot);
}
inline int blocks(int n, int b = 256) { return (n + b - 1) / b; }
} // namespace

extern "C" {

int vo_frame_create(vo_ctx* ctx, const vo_frame_opts* o, int nu, int nv, const double K[9], vo_frame_state** out) {
    if (!ctx || !o || !K || !out) return VO_ERR_BAD_ARG;
    *out = nullptr;
    if (o->max_landmarks < 4 || o->max_candidates < 1 || o->max_temporal_recall < 1 || !(o->max_reproj_err > 0) ||
        o->sel.keep < 0 || o->pose.num_hyp < 1 || nu < 8 || nv < 8) { ctx->last_error = "frame: bad options"; return VO_ERR_BAD_ARG; }
    (void)hipSetDevice(ctx->device);
    vo_frame_state* st = new vo_frame_state();
    st->o = *o; st->nu = nu; st->nv = nv;
    st->capL = o->max_landmarks; st->capC = o->max_candidates; st->PW = st->capL + st->capC;
    st->ring = o->max_temporal_recall + 2; st->P = st->ring + o->max_temporal_recall + 64;
    st->sort_padded = (st->PW + SORT_TILE - 1) / SORT_TILE * SORT_TILE;
    st->cur = 0; st->have_prev = false; st->nL = st->nC = 0; st->frame_no = 0; st->s = 0;
    st->arena = nullptr; st->img[0] = st->img[1] = st->img[2] = nullptr; st->h_counts = nullptr; st->h_rec = nullptr;
    size_t total = 0;
    carve(st, nullptr, &total);
    hipError_t e = hipMalloc((void**)&st->arena, total);
    if (e == hipSuccess) e = hipMemsetAsync(st->arena, 0, total, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&st->img[0], (size_t)nu * nv);
    if (e == hipSuccess) e = hipMalloc((void**)&st->img[1], (size_t)nu * nv);
    if (e == hipSuccess) e = hipMalloc((void**)&st->img[2], (size_t)nu * nv);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_ready, vo_int_event_flags());
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_fork_la, vo_int_event_flags());
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_state, vo_int_event_flags());
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_pose, vo_int_event_flags());
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_ltrack, vo_int_event_flags());
    st->overlap = !getenv("VO_FRAME_SERIAL");
    st->narrow = getenv("VO_FRAME_NARROW") != nullptr;
    st->tracks_together = getenv("VO_FRAME_TRACKS_TOGETHER") != nullptr;
    if (const char* wm = getenv("VO_FRAME_WIDE_MIN")) st->wide_min = atoi(wm);
    if (const char* ob = getenv("VO_FRAME_ONE_BLOCK")) st->one_block = atoi(ob);
    if (const char* lf = getenv("VO_FRAME_L_FIRST_MIN")) st->l_first_min = atoi(lf);
    if (const char* lh = getenv("VO_FRAME_L_HEAD_PCT")) st->l_head_pct = std::min(100, std::max(1, atoi(lh)));
    if (const char* sp = getenv("VO_FRAME_STAMPS")) {
        st->stamp_path = sp;
        if (const char* mk = getenv("VO_FRAME_STAMP_MASK")) st->stamp_mask = (unsigned)strtoul(mk, nullptr, 0);
        if (e == hipSuccess) e = hipHostMalloc((void**)&st->h_stamps, (size_t)STAMP_FRAMES * STAMP_N * sizeof(long long), 0);
        if (e == hipSuccess) memset(st->h_stamps, 0, (size_t)STAMP_FRAMES * STAMP_N * sizeof(long long));
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_detect[0], vo_int_event_flags());
    if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev_detect[1], vo_int_event_flags());
    if (e == hipSuccess) e = hipHostMalloc((void**)&st->h_fault, 64, 0);
    if (e == hipSuccess) memset(st->h_fault, 0, 64);
    st->fault_inject = getenv("VO_FRAME_FAULT_INJECT") != nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void**)&st->h_counts, 16 * sizeof(int32_t), 0);
    if (e == hipSuccess) e = hipHostMalloc((void**)&st->h_rec, VO_FRAME_MAX_PENDING * 40 * sizeof(double), 0);
    for (int k = 0; k < VO_FRAME_MAX_PENDING && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&st->ev_rec[k], hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx->last_error = std::string("frame: allocation failed: ") + hipGetErrorString(e);
        vo_frame_destroy(ctx, st);
        return e == hipErrorOutOfMemory ? VO_ERR_OOM : VO_ERR_HIP;
    }
    carve(st, st->arena, &total);
    ST_CHECK(hipMemcpyAsync(st->K, K, 72, hipMemcpyHostToDevice, ctx->stream));
    ST_CHECK(hipStreamSynchronize(ctx->stream));
    *out = st;
    return VO_OK;
}

int vo_frame_destroy(vo_ctx* ctx, vo_frame_state* st) {
    if (!ctx || !st) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);   // a look-ahead may still be running
    if (ctx->lane) (void)hipStreamSynchronize(ctx->lane);   // ... or a landmark branch
    if (st->arena) (void)hipFree(st->arena);
    for (int k = 0; k < 3; ++k) {
        if (st->img[k]) (void)hipFree(st->img[k]);
        if (st->pyr[k].I) (void)hipFree(st->pyr[k].I);
    }
    if (st->ev_ready) (void)hipEventDestroy(st->ev_ready);
    if (st->ev_fork_la) (void)hipEventDestroy(st->ev_fork_la);
    if (st->ev_state) (void)hipEventDestroy(st->ev_state);
    if (st->ev_pose) (void)hipEventDestroy(st->ev_pose);
    if (st->ev_ltrack) (void)hipEventDestroy(st->ev_ltrack);
    for (int k = 0; k < 2; ++k) if (st->ev_detect[k]) (void)hipEventDestroy(st->ev_detect[k]);
    for (int k = 0; k < VO_FRAME_MAX_PENDING; ++k) if (st->ev_rec[k]) (void)hipEventDestroy(st->ev_rec[k]);
    if (st->h_fault) (void)hipHostFree(st->h_fault);
    if (st->h_counts) (void)hipHostFree(st->h_counts);
    if (st->h_rec) (void)hipHostFree(st->h_rec);
    for (int r = 0; r < vo_frame_state::PIN_SLOTS; ++r) {
        if (st->ev_img[r]) (void)hipEventDestroy(st->ev_img[r]);
        if (st->h_img[r]) (void)hipHostFree(st->h_img[r]);
    }
    if (st->h_stamps) {
        if (FILE* fp = fopen(st->stamp_path.c_str(), "w")) {
            for (int f = 0; f < STAMP_FRAMES; ++f) {
                const long long* r = st->h_stamps + (size_t)f * STAMP_N;
                bool any = false;
                for (int k = 0; k < STAMP_N; ++k) any = any || r[k];
                if (!any) continue;
                fprintf(fp, "%d", f);
                for (int k = 0; k < STAMP_N; ++k) fprintf(fp, " %lld", r[k]);
                fprintf(fp, "\n");
            }
            fclose(fp);
        }
        (void)hipHostFree(st->h_stamps);
    }
    delete st;
    return VO_OK;
}

int vo_frame_set_state(vo_ctx* ctx, vo_frame_state* st, const double* landmarks, const double* keypoints,
                       const double* repro, int N, const double* ck, const double* cfk, const double* cposes,
                       const int32_t* ctime, int C, int frame_no) {
    if (!ctx || !st || N < 0 || C < 0 || frame_no < 0) return VO_ERR_BAD_ARG;
    if (N > st->capL || C > st->capC) { ctx->last_error = "frame: state larger than max_landmarks / max_candidates"; return VO_ERR_BAD_ARG; }
    if ((N > 0 && (!landmarks || !keypoints || !repro)) || (C > 0 && (!ck || !cfk || !cposes || !ctime))) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (st->pend_count) { ctx->last_error = "frame: frames were submitted and not collected yet (vo_frame_collect)"; return VO_ERR_BAD_ARG; }
    { int rc = drop_lookahead(ctx, st); if (rc) return rc; }   // the frame counter decides the look-ahead's work area
    const int s = st->s, cl = st->capL, cc = st->capC;
    std::vector<double> hL(3 * (size_t)std::max(N, 1)), hk(2 * (size_t)std::max(N, 1)), hc(2 * (size_t)std::max(C, 1)), hf(2 * (size_t)std::max(C, 1));
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) hL[(size_t)c * N + i] = landmarks[3 * (size_t)i + c];
        for (int c = 0; c < 2; ++c) hk[(size_t)c * N + i] = keypoints[2 * (size_t)i + c];
    }
    // distinct first poses go to the upload slots behind the ring
    std::vector<double> table; std::vector<int32_t> idx(std::max(C, 1));
    for (int i = 0; i < C; ++i) {
        for (int c = 0; c < 2; ++c) { hc[(size_t)c * C + i] = ck[2 * (size_t)i + c]; hf[(size_t)c * C + i] = cfk[2 * (size_t)i + c]; }
        const double* p = cposes + 12 * (size_t)i;
        int found = -1;
        for (size_t k = 0; k < table.size() / 12 && found < 0; ++k)
            if (!memcmp(table.data() + 12 * k, p, 96)) found = (int)k;
        if (found < 0) { found = (int)(table.size() / 12); table.insert(table.end(), p, p + 12); }
        idx[i] = st->ring + found;
    }
    if ((int)(table.size() / 12) > st->P - st->ring) { ctx->last_error = "frame: too many distinct first poses in the uploaded state"; return VO_ERR_BAD_ARG; }
    ST_CHECK(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < 3 && N > 0; ++c) ST_CHECK(hipMemcpyAsync(st->L[s] + (size_t)c * cl, hL.data() + (size_t)c * N, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
    for (int c = 0; c < 2 && N > 0; ++c) ST_CHECK(hipMemcpyAsync(st->kp[s] + (size_t)c * cl, hk.data() + (size_t)c * N, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) ST_CHECK(hipMemcpyAsync(st->rep[s], repro, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream));
    for (int c = 0; c < 2 && C > 0; ++c) {
        ST_CHECK(hipMemcpyAsync(st->cu[s] + (size_t)c * cc, hc.data() + (size_t)c * C, (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
        ST_CHECK(hipMemcpyAsync(st->cf[s] + (size_t)c * cc, hf.data() + (size_t)c * C, (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (C > 0) {
        ST_CHECK(hipMemcpyAsync(st->cpose[s], idx.data(), (size_t)C * 4, hipMemcpyHostToDevice, ctx->stream));
        ST_CHECK(hipMemcpyAsync(st->ctime[s], ctime, (size_t)C * 4, hipMemcpyHostToDevice, ctx->stream));
        ST_CHECK(hipMemcpyAsync(st->poses + 12 * (size_t)st->ring, table.data(), table.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    ST_CHECK(hipStreamSynchronize(ctx->stream));
    const int32_t sz[2] = {N, C};
    ST_CHECK(hipMemcpyAsync(st->sizes, sz, sizeof sz, hipMemcpyHostToDevice, ctx->stream));
    ST_CHECK(hipStreamSynchronize(ctx->stream));
    st->nL = N; st->nC = C; st->frame_no = frame_no;
    st->state_event = false;
    return VO_OK;
}

int vo_frame_get_counts(vo_ctx* ctx, vo_frame_state* st, int* N, int* C) {
    if (!ctx || !st) return VO_ERR_BAD_ARG;
    if (st->pend_count) { ctx->last_error = "frame: frames were submitted and not collected yet (vo_frame_collect)"; return VO_ERR_BAD_ARG; }
    if (N) *N = st->nL;
    if (C) *C = st->nC;
    return VO_OK;
}

int vo_frame_get_state(vo_ctx* ctx, vo_frame_state* st, double* landmarks, double* keypoints, double* repro,
                       double* ck, double* cfk, double* cposes, int32_t* ctime) {
    if (!ctx || !st) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (st->pend_count) { ctx->last_error = "frame: frames were submitted and not collected yet (vo_frame_collect)"; return VO_ERR_BAD_ARG; }
    const int s = st->s, N = st->nL, C = st->nC, cl = st->capL, cc = st->capC;
    std::vector<double> hL(3 * (size_t)std::max(N, 1)), hk(2 * (size_t)std::max(N, 1)), hc(2 * (size_t)std::max(C, 1)), hf(2 * (size_t)std::max(C, 1)),
        table(12 * (size_t)st->P);
    std::vector<int32_t> idx(std::max(C, 1));
    for (int c = 0; c < 3 && N > 0; ++c) ST_CHECK(hipMemcpyAsync(hL.data() + (size_t)c * N, st->L[s] + (size_t)c * cl, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    for (int c = 0; c < 2 && N > 0; ++c) ST_CHECK(hipMemcpyAsync(hk.data() + (size_t)c * N, st->kp[s] + (size_t)c * cl, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (N > 0 && repro) ST_CHECK(hipMemcpyAsync(repro, st->rep[s], (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    for (int c = 0; c < 2 && C > 0; ++c) {
        ST_CHECK(hipMemcpyAsync(hc.data() + (size_t)c * C, st->cu[s] + (size_t)c * cc, (size_t)C * 8, hipMemcpyDeviceToHost, ctx->stream));
        ST_CHECK(hipMemcpyAsync(hf.data() + (size_t)c * C, st->cf[s] + (size_t)c * cc, (size_t)C * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (C > 0) {
        ST_CHECK(hipMemcpyAsync(idx.data(), st->cpose[s], (size_t)C * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (ctime) ST_CHECK(hipMemcpyAsync(ctime, st->ctime[s], (size_t)C * 4, hipMemcpyDeviceToHost, ctx->stream));
        ST_CHECK(hipMemcpyAsync(table.data(), st->poses, table.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    ST_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < N; ++i) {
        if (landmarks) for (int c = 0; c < 3; ++c) landmarks[3 * (size_t)i + c] = hL[(size_t)c * N + i];
        if (keypoints) for (int c = 0; c < 2; ++c) keypoints[2 * (size_t)i + c] = hk[(size_t)c * N + i];
    }
    for (int i = 0; i < C; ++i) {
        if (ck) for (int c = 0; c < 2; ++c) ck[2 * (size_t)i + c] = hc[(size_t)c * C + i];
        if (cfk) for (int c = 0; c < 2; ++c) cfk[2 * (size_t)i + c] = hf[(size_t)c * C + i];
        if (cposes) {
            const int k = idx[i];
            if (k < 0 || k >= st->P) { ctx->last_error = "frame: corrupt pose index"; return VO_ERR_HIP; }
            memcpy(cposes + 12 * (size_t)i, table.data() + 12 * (size_t)k, 96);
        }
    }
    return VO_OK;
}

int vo_frame_set_image(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device) {
    if (!ctx || !st || !img) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (st->pend_count) { ctx->last_error = "frame: frames were submitted and not collected yet (vo_frame_collect)"; return VO_ERR_BAD_ARG; }
    int rc = drop_lookahead(ctx, st);
    if (rc) return rc;
    rc = upload_image(ctx, st, st->cur, img, on_device);
    if (rc) return rc;
    st->have_prev = true; st->state_event = false;
    return VO_OK;
}

int vo_frame_announce(vo_ctx* ctx, vo_frame_state* st, const uint8_t* next_img, int on_device) {
    if (!ctx || !st) return VO_ERR_BAD_ARG;
    st->next_src = next_img;    // NULL withdraws the announcement
    st->next_on_device = on_device;
    return VO_OK;
}

int vo_frame_submit(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device) {
    if (!ctx || !st || !img) return VO_ERR_BAD_ARG;
    if (!st->have_prev) { ctx->last_error = "frame: no previous image (vo_frame_set_image)"; return VO_ERR_BAD_ARG; }
    if (st->pend_count == VO_FRAME_MAX_PENDING) { ctx->last_error = "frame: VO_FRAME_MAX_PENDING frames are waiting to be collected"; return VO_ERR_BAD_ARG; }
    (void)hipSetDevice(ctx->device);
    hipStream_t sm = ctx->stream;
    const vo_frame_opts& o = st->o;
    const int s = st->s, n = 1 - s, a = st->cur, b = (st->cur + 1) % 3, c = (st->cur + 2) % 3;
    const int cl = st->capL, cc = st->capC, pw = st->PW;
    // Every launch below is sized by the capacities and reads the counts on the device (st->sizes for the state the
    // frame starts from, `counts` for what the frame finds): nothing here depends on a value the host would have
    // to wait for.  st->nL / st->nC are the counts of the last COLLECTED frame -- a hint for the form of the merge.
    const int nL_hint = st->nL, nC_hint = st->nC;
    const int frame_no = st->frame_no + 1;
    const int slot = frame_no % st->ring;
    double* pose_now = st->poses + 12 * (size_t)slot;
    int32_t* counts = st->counts2[frame_no & 1];      // by frame parity: the frame before may still be packing its own
    double* rsout = st->rsout2[frame_no & 1];
    const double eps = o.klt.eps > 0 ? o.klt.eps : 0.01;
    int rc;
    // Two branches (modules/VisualOdometry.m: the pose step :278-318 needs the landmarks only, and the landmarks of
    // the state are final once the frame before has merged, :322 -- its selection :330-345 only adds candidates):
    //   lane stream   landmark track -> tracked landmarks -> pose -> reprojection filter       (waits for ev_state)
    //   ctx stream    [the frame before: selection, append]  candidate track -> (ev_pose) triangulation, merge,
    //                 compactions -> ev_state -> selection, append, record
    // so a frame's landmark branch runs beside the selection of the frame before it.  The loop is a chain of
    // latency-bound launches on a nearly empty chip: what runs side by side is simply saved.
    hipStream_t sl = sm;
    if (st->overlap) {
        if ((rc = vo_int_lane_ensure(ctx))) return rc;
        sl = ctx->lane;
    }
    // [5a] :330 the image-only half of the selection (corner response, threshold, regional maxima) runs on the
    // ctx's side stream, beside the pyramid build / tracking / pose / triangulation; its other half waits for it
    // at [5b].  If this image was announced to the previous call (vo_frame_announce), its copy, pyramid and detect
    // half were issued back then and only have to be waited for.
    const int set = frame_no & 1;                       // the selection work area of this frame
    vo_sel_work& selw = st->selw[set];
    // ... unless another selection on this ctx has taken the look-ahead's work area since (another loop, a
    // stand-alone vo_select_candidates): then it is abandoned like a wrong announcement
    const bool ahead = st->ahead_src && st->ahead_src == img && st->ahead_on_device == on_device &&
                       (o.sel.keep <= 0 || ctx->sel_gen[set] == st->ahead_gen);
    bool detect_joined = false;      // the ctx stream already waits for this frame's detect half
    if (ahead) {
        // the side stream ran copy, pyramid (ev_ready), detect half (ev_detect) in that order: one wait of the ctx
        // stream for the later event covers both (every wait in the chain of a frame costs 5-10 us of its own, even
        // when long satisfied); the landmark branch needs the pyramid only
        if (o.sel.keep > 0) { ST_CHECK(hipStreamWaitEvent(sm, st->ev_detect[set], 0)); detect_joined = true; }
        else ST_CHECK(hipStreamWaitEvent(sm, st->ev_ready, 0));
        if (sl != sm) ST_CHECK(hipStreamWaitEvent(sl, st->ev_ready, 0));
        st->ahead_src = nullptr;
    } else {
        const uint8_t* announced = st->next_src; const int announced_dev = st->next_on_device;
        if ((rc = drop_lookahead(ctx, st))) return rc;    // something else was issued ahead: it is abandoned
        st->next_src = announced; st->next_on_device = announced_dev;
        if ((rc = copy_image(ctx, st, b, img, on_device))) return rc;
        if (o.sel.keep > 0) {
            if ((rc = vo_int_sel_work_prepare(ctx, st->nu, st->nv, o.sel.keep, cl + cc, o.sel.min_distance, &selw, set))) return rc;
            if ((rc = vo_int_side_fork(ctx))) return rc;
            if ((rc = vo_launch_select_detect(ctx, ctx->side, st->img[b], st->nu, st->nv, &o.sel, &selw))) return rc;
            ST_CHECK(hipEventRecord(st->ev_detect[set], ctx->side));
        }
        if ((rc = build_pyramid(ctx, st, b))) return rc;
        st->state_event = false;    // the pyramid was built on the ctx stream: the landmark branch starts behind it
    }
    if (!st->state_event) {         // first frame after a state / image load, or no look-ahead: behind everything so far
        ST_CHECK(hipEventRecord(st->ev_state, sm));
        st->state_event = true;
    }

    // ---- landmark branch ----
    if (sl != sm) {
        ST_CHECK(hipStreamWaitEvent(sl, st->ev_state, 0));
        ctx->stream = sl; ctx->in_lane = true;           // the launchers below enqueue on ctx->stream
    }
    auto leave_lane = [&]() { ctx->stream = sm; ctx->in_lane = false; };
    // large states: the ordered compactions as launches of one-wave workgroups (compact_wave) instead of one block each
    const bool wide = (cl > st->wide_min || cc > st->wide_min) && cw_blocks(pw) <= CW_MAX_BLOCKS && !st->narrow;
    auto chain = [&](int site) {
        if (++st->chain_epoch == 0) ++st->chain_epoch;     // 0 is what a never-written word holds
        int skip = -1;
        if (st->fault_inject && site == 0) { skip = 0; st->fault_inject = false; }   // (tests: the first workgroup of the next tracked-landmark compaction stays silent)
        return Chain{st->chain + (size_t)site * CW_MAX_BLOCKS, st->chain_epoch, st->h_fault, skip};
    };
    auto stamp = [&](hipStream_t q, int id) {
        if (st->h_stamps && frame_no < STAMP_FRAMES && ((st->stamp_mask >> id) & 1)) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, q, st->h_stamps + (size_t)frame_no * STAMP_N + id);
    };
    stamp(sl, 0);
    // [1] :278-281 track the landmarks' keypoints (read straight from the state) and keep the validly tracked ones in
    // order.  The number M of those stays on the device (counts[6]): the pose step, the reprojection filter and the
    // compaction are sized by the capacity and read M themselves.
    const bool l_first = wide && sl != sm && !st->tracks_together && nL_hint + nC_hint >= st->l_first_min;
    // (landmarks first, below: the candidates' launch starts when the first l_head_pct % of the landmarks are out)
    const int l_head = l_first && st->l_head_pct < 100 ? std::max(1, (int)((long)std::min(cl, std::max(nL_hint, 1)) * st->l_head_pct / 100)) : 0;
    if ((rc = vo_launch_klt(ctx, &st->pyr[a], &st->pyr[b], st->kp[s], st->kp[s] + cl, cl, o.klt.block_w, o.klt.block_h,
                            o.klt.max_iters, eps, o.klt.max_bidir_err, st->q, st->q + pw, st->valid, /* scores: processFrame discards them, :278 */ nullptr, nullptr, st->sizes + 0,
                            l_head))) { leave_lane(); return rc; }
    // Large states: two tracking launches that start together share the chip and end together, and the pose chain -- the
    // frame's critical path -- starts when both are done.  The candidates' launch starts behind the landmarks' instead (it
    // is needed from the triangulation on) and runs beside the pose chain, whose one-wave kernels take the holes it leaves.
    // From about 32k keypoints a frame, that is: measured with as many seeded candidates as landmarks, the two orders are
    // equal at 20k + 13k keypoints (0.696 ms per frame either way), landmarks first wins at 24k + 21k (0.807 against 0.836)
    // and loses below (16k + 7k: 0.600 against 0.584; 8k + 8k: 0.401 against 0.369) -- two launches that do not fill the
    // chip several times over lose more by running one after the other than the pose chain gains by starting early.
    if (l_first && st->l_head_pct >= 100) ST_CHECK(hipEventRecord(st->ev_ltrack, sl));
    stamp(sl, 1);
    {   // (the frame's counters and tombstones are cleared by the same launch)
        TrackedOp top = {st->valid, st->q, st->q + pw, st->L[s], cl, st->vk, st->vl};
        if (wide) hipLaunchKernelGGL(k_tracked_w, dim3(cw_blocks(cl)), dim3(64), 0, sl, counts, st->alive, pw, top, cl, st->sizes + 0, chain(0));
        else hipLaunchKernelGGL(k_tracked, dim3(1), dim3(1024), 0, sl, counts, st->alive, pw, top, cl, st->sizes + 0);
    }
    stamp(sl, 2);
    // [2] :285-289 pose (fewer than 4 tracked points: k_best gives status 1 and the identity, as the toolbox function errors out)
    {
        vo_ransac_opts ro = o.pose;
        ro.seed = o.pose.seed + (uint64_t)frame_no;
        vo_ransac_work w;
        if ((rc = vo_int_ransac_work_prepare(ctx, ro.num_hyp, &w))) { leave_lane(); return rc; }
        vo_strides s2 = {1, cl}, s3 = {1, cl};
        if ((rc = vo_launch_ransac(ctx, st->vk, s2, st->vl, s3, cl, st->K, &ro, nullptr, nullptr, &w, rsout, nullptr, counts + 6, 0,
                                   nullptr))) { leave_lane(); return rc; }   // (no inlier flags asked for: k_pose_tail below ends the pose step)
        stamp(sl, 3);
        // [3] :298-315 the record's pose and inliers, reprojection error, strict filter, survivors into the merge arrays
        PoseTailArgs pt;
        pt.vk = st->vk; pt.vl = st->vl; pt.capL = cl; pt.M_dev = counts + 6; pt.K = st->K;
        pt.thr2 = ro.max_reproj_err * ro.max_reproj_err; pt.rsout = rsout; pt.inl = st->inl; pt.pose_now = pose_now; pt.err = st->err;
        pt.kop = KeptOp{st->err, o.max_reproj_err, st->vk, st->vl, cl, pw, st->wkp, st->wL, st->wrep, st->alive};
        pt.count = counts + 7;
        if (wide) hipLaunchKernelGGL(k_pose_tail_w, dim3(cw_blocks(cl)), dim3(64), 0, sl, pt, chain(1));
        else hipLaunchKernelGGL(k_pose_tail, dim3(1), dim3(1024), 0, sl, pt);
    }
    stamp(sl, 4);
    if (sl != sm) {
        leave_lane();
        ST_CHECK(hipEventRecord(st->ev_pose, sl));
    }

    // ---- candidate branch and the rest, on the ctx stream ----
    // look-ahead, part 1: the point of the ctx stream behind which the side stream may start.  The look-ahead
    // overwrites the image slot that held the PREVIOUS frame's first image -- last read by that frame's two tracking
    // launches, both finished before its triangulation -- and the selection work area of the previous frame, whose
    // filter ran on this stream: everything it must not overtake lies behind this point.  (It used to start behind
    // this frame's tracking launches and then ran beside the selection filter and the next landmark track, all of
    // them single chains of dependent instructions that slow each other down on a shared SIMD.)
    if (st->next_src) ST_CHECK(hipEventRecord(st->ev_fork_la, sm));
    stamp(sm, 5);
    // :71 track the candidates: points from the state, results straight into the candidates' work arrays
    if (l_first) {
        if (st->l_head_pct >= 100) ST_CHECK(hipStreamWaitEvent(sm, st->ev_ltrack, 0));
        else if ((rc = vo_launch_wait_head(ctx, st->sizes + 0, st->h_fault))) return rc;      // (ctx->stream is sm again)
    }
    if ((rc = vo_launch_klt(ctx, &st->pyr[a], &st->pyr[b], st->cu[s], st->cu[s] + cc, cc, o.klt.block_w, o.klt.block_h,
                            o.klt.max_iters, eps, o.klt.max_bidir_err, st->cq, st->cq + cc, st->cvalid, /* :71 `~` */ nullptr, nullptr, st->sizes + 1))) return rc;
    stamp(sm, 6);
    if (sl != sm) ST_CHECK(hipStreamWaitEvent(sm, st->ev_pose, 0));
    stamp(sm, 7);
    // [4] :322 -> :63-151 triangulate the tracked candidates, classify, merge in candidate order
    {
        vo_strides c2 = {1, cc}, c3 = {1, cc};
        if ((rc = vo_launch_triangulate(ctx, st->cf[s], st->cq, c2, st->poses, 0, st->cpose[s], cc, st->P, pose_now, st->K, st->X, c3,
                                        st->crep, st->cang, st->cok, st->sizes + 1))) return rc;
        const ClassifyArgs ca = {st->cvalid, st->cok, st->crep, st->cang, st->ctime[s], o.max_reproj_err, o.angular_threshold_deg,
                                 o.max_temporal_recall};
        MergeArgs m;
        m.nC_dev = st->sizes + 1; m.cap = cl; m.PW = pw; m.capC = cc; m.X = st->X; m.crep = st->crep; m.cq = st->cq;
        m.wkp = st->wkp; m.wL = st->wL; m.wrep = st->wrep; m.alive = st->alive; m.n0 = counts + 7; m.plist = st->plist;
        m.n_promoted = counts + 8; m.c = ca;
        // small pools: one block does it all.  Either form is right for any pool that fits the work arrays; which
        // one is faster is guessed from the last collected frame's counts (the pool grows by at most `keep` a frame)
        // the aged candidates and the final landmark order are independent: a block each -- of the one-block merge's
        // launch, or of a launch of their own behind the wide form; the landmarks' block also leaves the number of
        // landmarks the next frame starts from (st->sizes[0])
        CandOp cop = {ca, cc, st->cq, st->cf[s], st->cpose[s], st->ctime[s], st->cu[n], st->cf[n], st->cpose[n], st->ctime[n]};
        LandmarkOp lop = {st->alive, pw, cl, st->wkp, st->wL, st->wrep, st->kp[n], st->L[n], st->rep[n]};
        if (pw <= MERGE_LDS && nL_hint + nC_hint + VO_FRAME_MAX_PENDING * o.sel.keep <= st->one_block) {
            hipLaunchKernelGGL(k_merge_par, dim3(2), dim3(1024), 0, sm, m, cop, cc, counts + 1, lop, counts + 0, st->sizes);
        } else {                                               // the all-pairs counts spread over the chip
            const int padded = st->sort_padded;
            MergeWideArgs mw = {m, st->pin, (uint8_t*)st->pos, (uint8_t*)st->pos + pw, counts + 9,
                                {st->skey, st->skey + padded}, {st->sidx, st->sidx + padded}, st->sbefore, padded};
            if (wide) hipLaunchKernelGGL(k_merge_list_w, dim3(cw_blocks(std::max(cc, cl))), dim3(64), 0, sm, mw, chain(2));
            else hipLaunchKernelGGL(k_merge_list, dim3(1), dim3(1024), 0, sm, mw);
            hipLaunchKernelGGL(k_sort_tiles, dim3(padded / SORT_TILE), dim3(1024), 0, sm, mw);
            // (the pool's size is on the device: the run length is chosen for the size the last collected frame suggests)
            const int Ld = sort_decide_run(std::min<long>(pw, (long)nL_hint + nC_hint + VO_FRAME_MAX_PENDING * o.sel.keep));
            int from = 0;
            for (int L = SORT_TILE; L < Ld && L < padded; L <<= 1, from ^= 1)
                hipLaunchKernelGGL(k_sort_merge, dim3(blocks(padded)), dim3(256), 0, sm, mw, L, from);
            hipLaunchKernelGGL(k_sort_decide, dim3(blocks(padded, 64)), dim3(64), 0, sm, mw, Ld, from);
            if (wide) {
                hipLaunchKernelGGL(k_merge_write_w, dim3(cw_blocks(std::max(cc, cl))), dim3(64), 0, sm, mw, chain(3));
                const Chain cha = chain(4), chb = chain(5);
                hipLaunchKernelGGL((k_compact2_w<CandOp, LandmarkOp>), dim3(cw_blocks(std::max(cc, pw)), 2), dim3(64), 0, sm, cop, cc, counts + 1,
                                   st->sizes + 1, cha, lop, pw, counts + 0, st->sizes + 0, cl, chb);
            } else {
                hipLaunchKernelGGL(k_merge_write, dim3(1), dim3(1024), 0, sm, mw);
                hipLaunchKernelGGL((k_compact2<CandOp, LandmarkOp>), dim3(2), dim3(1024), 0, sm, cop, cc, counts + 1, st->sizes + 1, lop, pw, counts + 0,
                                   st->sizes + 0, cl);
            }
        }
    }
    stamp(sm, 8);
    // The next frame's landmark branch may start here -- one launch later when there is a selection: its k_uniform is
    // one block of 16 waves with 127 registers, which needs a compute unit to itself, and the landmark track puts a
    // wave on every one of them.  Released at the same moment the track won that race and the block waited for the
    // track's waves to end (45 -> 60-90 us on the ctx stream's chain); released behind k_kill (5 us), the block,
    // next in its stream, is placed before the other stream's wait has even been seen.
    if (o.sel.keep <= 0) ST_CHECK(hipEventRecord(st->ev_state, sm));
    // [5b] :330-345 new candidates away from every tracked point; the number of existing points stays on the
    // device (counts[0] + counts[1]), so the host does not wait here
    // (the single cast of [keypoints; candidate_keypoints], :330-331, happens inside the filter's kernel)
    if (o.sel.keep > 0) {
        if (!detect_joined) ST_CHECK(hipStreamWaitEvent(sm, st->ev_detect[set], 0));
        stamp(sm, 9);
        if ((rc = vo_launch_select_filter(ctx, sm, st->nu, st->nv, 0, nullptr, nullptr, cl + cc, counts, &o.sel, &selw, st->su, st->sv,
                                          st->slin, nullptr, counts + 2, st->kp[n], cl, st->cu[n], cc, st->ev_state))) return rc;
    }
    stamp(sm, 10);
    const int ps = (st->pend_head + st->pend_count) % VO_FRAME_MAX_PENDING;
    // (one-wave workgroups: the launch may find the next frame's landmark track on every compute unit already, and a
    // four-wave workgroup then waits for four free slots on one of them -- 0.28 ms in the trace of a 25k + 25k frame)
    hipLaunchKernelGGL(k_append_new, dim3(blocks(std::max(o.sel.keep, 36), 64)), dim3(64), 0, sm, counts, cl, cc, st->su, st->sv, slot,
                       st->cu[n], st->cf[n], st->cpose[n], st->ctime[n], rsout, pose_now, st->h_rec + 40 * (size_t)ps, st->sizes);
    // the frame's record into the next pinned slot; whoever collects the frame waits for the slot's event only
    // (enqueued before the look-ahead's dozen launches: the record is not held up behind host work it does not need)
    ST_CHECK(hipEventRecord(st->ev_rec[ps], sm));
    stamp(sm, 11);
    st->pend_frame_no[ps] = frame_no;
    ++st->pend_count;
    st->s = n; st->cur = b; st->frame_no = frame_no;   // the frame is in: what follows is work for the next one
    // look-ahead, part 2: the announced image's copy, pyramid and detect half on the side stream.  Enqueued here,
    // after this frame's own launches (the host's launch rate is what the loop runs at: the ctx stream must not
    // wait for launches of work it does not need), it may run since the tracking launch finished -- beside
    // whatever of this frame's tail is still queued.
    if (st->next_src) {
        vo_sel_work& nw = st->selw[1 - set];
        if (o.sel.keep > 0 &&
            (rc = vo_int_sel_work_prepare(ctx, st->nu, st->nv, o.sel.keep, cl + cc, o.sel.min_distance, &nw, 1 - set))) return rc;
        if ((rc = vo_int_frame_prepare(ctx, &st->pyr[c], st->nu, st->nv, o.klt.levels))) return rc;   // may grow (first use)
        if ((rc = vo_int_side_ensure(ctx))) return rc;
        ST_CHECK(hipStreamWaitEvent(ctx->side, st->ev_fork_la, 0));   // long satisfied by the time this is enqueued
        stamp(ctx->side, 12);
        if ((rc = copy_image(ctx, st, c, st->next_src, st->next_on_device, ctx->side))) return rc;
        if ((rc = build_pyramid(ctx, st, c, ctx->side))) return rc;
        stamp(ctx->side, 13);
        ST_CHECK(hipEventRecord(st->ev_ready, ctx->side));
        if (o.sel.keep > 0) {
            if ((rc = vo_launch_select_detect(ctx, ctx->side, st->img[c], st->nu, st->nv, &o.sel, &nw))) return rc;
            ST_CHECK(hipEventRecord(st->ev_detect[1 - set], ctx->side));
        }
        stamp(ctx->side, 14);
        st->ahead_src = st->next_src; st->ahead_on_device = st->next_on_device; st->ahead_gen = ctx->sel_gen[1 - set];
        st->next_src = nullptr;
    }
    ST_CHECK(hipGetLastError());
    return VO_OK;
}

// The two decisions of the merge policy for a pool given as such (see include/vo_hip.h): the sort-based form that
// vo_frame_submit uses for large pools, on the caller's values.
int vo_dev_merge_ranks(vo_ctx* ctx, const double* d_values, int n0, int np, int cap, uint8_t* d_survives, uint8_t* d_accepted) {
    if (!ctx || n0 < 0 || np < 0 || cap < 0) return VO_ERR_BAD_ARG;
    const long P = (long)n0 + np;
    if (P == 0) return VO_OK;
    if (!d_values || !d_survives || !d_accepted || P > (1l << 26)) return VO_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    const int padded = (int)((P + SORT_TILE - 1) / SORT_TILE * SORT_TILE);
    // work: the pool in place (the caller's array), two key and two index buffers, the counts, n0 and np on the device
    const size_t bytes = (size_t)padded * (2 * 8 + 2 * 4 + 4) + 5 * 256;
    int rc = vo_buf_reserve(ctx, &ctx->scratch[19], bytes);
    if (rc) return rc;
    Bump b; b.base = (char*)ctx->scratch[19].p;
    MergeWideArgs mw = {};
    mw.key[0] = b.take<double>(padded); mw.key[1] = b.take<double>(padded);
    mw.idx[0] = b.take<int32_t>(padded); mw.idx[1] = b.take<int32_t>(padded);
    mw.before = b.take<int32_t>(padded);
    int32_t* d_n = b.take<int32_t>(2);
    const int32_t hn[2] = {n0, np};
    ST_CHECK(hipMemcpyAsync(d_n, hn, sizeof hn, hipMemcpyHostToDevice, ctx->stream));
    mw.m.n0 = d_n; mw.np_out = d_n + 1; mw.m.cap = cap;
    mw.pool = const_cast<double*>(d_values); mw.surv = d_survives; mw.acc = d_accepted; mw.padded = padded;
    hipStream_t sm = ctx->stream;
    hipLaunchKernelGGL(k_sort_tiles, dim3(padded / SORT_TILE), dim3(1024), 0, sm, mw);
    const int Ld = sort_decide_run(P);
    int from = 0;
    for (int L = SORT_TILE; L < Ld; L <<= 1, from ^= 1)
        hipLaunchKernelGGL(k_sort_merge, dim3(blocks(padded)), dim3(256), 0, sm, mw, L, from);
    hipLaunchKernelGGL(k_sort_decide, dim3(blocks(padded, 64)), dim3(64), 0, sm, mw, Ld, from);
    ST_CHECK(hipGetLastError());
    ST_CHECK(hipStreamSynchronize(sm));   // hn lives on this frame of the stack
    return VO_OK;
}

int vo_frame_collect(vo_ctx* ctx, vo_frame_state* st, double pose43[12], vo_frame_info* info) {
    if (!ctx || !st || !pose43 || !info) return VO_ERR_BAD_ARG;
    if (!st->pend_count) { ctx->last_error = "frame: nothing was submitted (vo_frame_submit)"; return VO_ERR_BAD_ARG; }
    (void)hipSetDevice(ctx->device);
    const int ps = st->pend_head;
    ST_CHECK(hipEventSynchronize(st->ev_rec[ps]));
    const double* rec = st->h_rec + 40 * (size_t)ps;
    memcpy(st->h_counts, rec + 28, 16 * sizeof(int32_t));
    int32_t ri[5];
    memcpy(ri, rec + 12, sizeof ri);
    memcpy(pose43, rec + 16, 96);
    info->status = ri[0]; info->trials_used = ri[3]; info->n_inliers = ri[4];
    info->n_landmarks = st->h_counts[0]; info->n_candidates = st->h_counts[3]; info->n_tracked = st->h_counts[6];
    info->n_promoted = st->h_counts[8]; info->n_new = st->h_counts[5]; info->overflow = st->h_counts[4];
    info->frame_no = st->pend_frame_no[ps];
    st->nL = std::min(st->h_counts[0], st->capL); st->nC = st->h_counts[3];
    st->pend_head = (ps + 1) % VO_FRAME_MAX_PENDING; --st->pend_count;
    // A device-side wait of this frame (or of one submitted behind it) gave up: an ordered compaction did not see a
    // count it needed within CW_SPINS looks, or the candidates' launch never saw the head of the landmarks'.  The
    // grid drained and nothing was written outside the arrays, but counts and order are not the reference's: say so.
    if (const int32_t f = __atomic_exchange_n(st->h_fault, 0, __ATOMIC_RELAXED)) {
        ctx->last_error = f == 2 ? "frame: the wait for the head of the landmarks' tracking launch gave up; the frame's result is not valid (load a state again)"
                                 : "frame: a compaction's wait for a workgroup's count gave up; the frame's result is not valid (load a state again)";
        return VO_ERR_HIP;
    }
    return VO_OK;
}

int vo_frame_process(vo_ctx* ctx, vo_frame_state* st, const uint8_t* img, int on_device, double pose43[12], vo_frame_info* info) {
    if (!ctx || !st || !img || !pose43 || !info) return VO_ERR_BAD_ARG;
    if (st->pend_count) { ctx->last_error = "frame: frames were submitted and not collected yet (vo_frame_collect)"; return VO_ERR_BAD_ARG; }
    const int rc = vo_frame_submit(ctx, st, img, on_device);
    return rc ? rc : vo_frame_collect(ctx, st, pose43, info);
}

} // extern "C"
