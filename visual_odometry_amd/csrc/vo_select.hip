// vo_select.hip -- candidate keypoint selection: corner response (Shi-Tomasi min-eigen or Harris),
// 8-neighbour non-max suppression with sub-pixel refinement, distance filter against existing
// points, uniform (grid) selection; plus the greedy box-suppression selector of old/selectKeypoints.m.
// Replaces detectMinEigenFeatures + pdist2 + selectUniform as called by
// modules/selectCandidateKeypoints.m:20-35, and old/harris.m:3-23 / old/selectKeypoints.m:5-14.
// Normative definitions: oracle/vo_oracle.h N6-N8.  Every decision (threshold, NMS, tie-breaks,
// distance test, grid cell, ordering) is taken on integers or on IEEE doubles/floats evaluated in
// the oracle's operation order (this file is compiled with -ffp-contract=off), so the selected
// indices are bit-exact.
//
// Bound: image-wide stencils are HBM/L2 streaming (read 1 B/px, write 8 B/px response, re-read
// 8 B/px for NMS); the distance filter is a tiled all-pairs test in LDS (VALU bound).
#include "vo_common.h"

namespace {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

constexpr int RS_TU = 64, RS_TV = 16; // response tile

struct RespArgs {
    const uint8_t* img;
    int nu, nv, r, scorer;
    double kappa;
    int32_t w[16];
    double* resp;
    unsigned long long* maxbits;
};

// zeroes the running maximum and (optionally) a word range -- the corner grid's tables -- in one launch.  A kernel of
// our own, not hipMemsetAsync: the runtime's fill kernels were seen to hold back the dispatch of kernels on the
// ctx's OTHER streams until they had finished (the frame's tail waited ~10-20 us for the side stream's memsets).
__global__ __launch_bounds__(256) void k_sel_zero(unsigned long long* maxbits, unsigned* p, long nwords) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && maxbits) *maxbits = 0ull;
    if (i < nwords) p[i] = 0u;
}

// One block = one RS_TU x RS_TV tile.  LDS: uint8 tile with halo r+1; int32 products on the tile
// with halo r; int32 horizontal sums, int64 vertical sums.  MINEIGEN uses replicate borders at every stage (N6), HARRIS
// only produces values where the whole support is inside the image (old/harris.m 'valid').
__global__ __launch_bounds__(256) void k_response(RespArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int r = a.r, nu = a.nu, nv = a.nv;
    const int SW = RS_TU + 2 * r + 2, SH = RS_TV + 2 * r + 2; // staged bytes
    const int PW = RS_TU + 2 * r, PH = RS_TV + 2 * r;         // products
    uint8_t* S = smem;
    int32_t* pa = (int32_t*)(smem + ((SW * SH + 15) & ~15));
    int32_t* pb = pa + PW * PH;
    int32_t* pc = pb + PW * PH;
    // horizontal sums fit int32: |product| <= 255^2 (min-eigen) or (4*255)^2 (Harris), weights sum to
    // 4096 (Gaussian, N6) or the patch width <= 15 (box); the vertical sums need 64 bits
    int32_t* ha = pc + PW * PH;
    int32_t* hb = ha + RS_TU * PH;
    int32_t* hc = hb + RS_TU * PH;
    const int u0 = blockIdx.x * RS_TU, v0 = blockIdx.y * RS_TV;
    const int su0 = u0 - r - 1, sv0 = v0 - r - 1;
    const int tid = threadIdx.x;
    for (int c = tid; c < SW * SH; c += 256) {
        int y = c / SW, x = c - y * SW;
        S[c] = a.img[(long)clampi(sv0 + y, 0, nv - 1) * nu + clampi(su0 + x, 0, nu - 1)];
    }
    __syncthreads();
#define PX(cu, cv) ((int)S[((cv) - sv0) * SW + ((cu) - su0)])
    for (int c = tid; c < PW * PH; c += 256) {
        int y = c / PW, x = c - y * PW;
        int ru = u0 - r + x, rv = v0 - r + y;
        int va, vb, vc;
        if (a.scorer == VO_SCORER_MINEIGEN) {
            int cu = clampi(ru, 0, nu - 1), cv = clampi(rv, 0, nv - 1);
            int du = PX(clampi(cu + 1, 0, nu - 1), cv) - PX(clampi(cu - 1, 0, nu - 1), cv);
            int dv = PX(cu, clampi(cv + 1, 0, nv - 1)) - PX(cu, clampi(cv - 1, 0, nv - 1));
            va = du * du; vb = du * dv; vc = dv * dv;
        } else {
            // old/harris.m:3-10, centre (ru, rv); zero where the 3x3 support leaves the image
            if (ru >= 1 && ru <= nu - 2 && rv >= 1 && rv <= nv - 2) {
                int ix = (PX(ru - 1, rv - 1) + 2 * PX(ru - 1, rv) + PX(ru - 1, rv + 1)) -
                         (PX(ru + 1, rv - 1) + 2 * PX(ru + 1, rv) + PX(ru + 1, rv + 1));
                int iy = (PX(ru - 1, rv - 1) + 2 * PX(ru, rv - 1) + PX(ru + 1, rv - 1)) -
                         (PX(ru - 1, rv + 1) + 2 * PX(ru, rv + 1) + PX(ru + 1, rv + 1));
                va = ix * ix; vb = ix * iy; vc = iy * iy;
            } else { va = vb = vc = 0; }
        }
        pa[c] = va; pb[c] = vb; pc[c] = vc;
    }
#undef PX
    __syncthreads();
    // horizontal pass.  MINEIGEN: index clamp(u+k) of the product arrays (replicate the PRODUCTS).
    for (int c = tid; c < RS_TU * PH; c += 256) {
        int y = c / RS_TU, x = c - y * RS_TU;
        int32_t sa = 0, sb = 0, sc = 0;
        for (int k = -r; k <= r; ++k) {
            int xx;
            if (a.scorer == VO_SCORER_MINEIGEN) xx = clampi(u0 + x + k, 0, nu - 1) - (u0 - r);
            else xx = x + r + k;
            xx = clampi(xx, 0, PW - 1);
            const int32_t w = a.w[k + r];
            sa += w * pa[y * PW + xx]; sb += w * pb[y * PW + xx]; sc += w * pc[y * PW + xx];
        }
        ha[c] = sa; hb[c] = sb; hc[c] = sc;
    }
    __syncthreads();
    double bmax = 0.0;
    for (int c = tid; c < RS_TU * RS_TV; c += 256) {
        int y = c / RS_TU, x = c - y * RS_TU;
        int u = u0 + x, v = v0 + y;
        if (u >= nu || v >= nv) continue;
        long long sa = 0, sb = 0, sc = 0;
        for (int k = -r; k <= r; ++k) {
            int yy;
            if (a.scorer == VO_SCORER_MINEIGEN) yy = clampi(v + k, 0, nv - 1) - (v0 - r);
            else yy = y + r + k;
            yy = clampi(yy, 0, PH - 1);
            const long long w = a.w[k + r];
            sa += w * ha[yy * RS_TU + x]; sb += w * hb[yy * RS_TU + x]; sc += w * hc[yy * RS_TU + x];
        }
        double out;
        if (a.scorer == VO_SCORER_MINEIGEN) {
            double A = (double)sa, B = (double)sb, C = (double)sc;
            double dif = A - C;
            double rad = dif * dif + 4.0 * (B * B);
            double lam = 0.5 * ((A + C) - sqrt(rad));
            if (lam < 0.0) lam = 0.0;
            out = lam * (1.0 / 67108864.0);
        } else {
            out = 0.0;
            if (u >= 1 + r && u < nu - 1 - r && v >= 1 + r && v < nv - 1 - r) {
                double sxx = (double)sa, sxy = (double)sb, syy = (double)sc;
                double tr = sxx + syy;
                double s = (sxx * syy - sxy * sxy) - a.kappa * (tr * tr);
                if (s < 0) s = 0;
                out = s;
            }
        }
        a.resp[(long)v * nu + u] = out;
        if (out > bmax) bmax = out;
    }
    if (a.maxbits) {
        // non-negative doubles order like their bit patterns
        unsigned long long bits = (unsigned long long)__double_as_longlong(bmax);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            unsigned long long o = __shfl_xor(bits, s, 64);
            bits = o > bits ? o : bits;
        }
        // a plain look first: the word only grows, and after the first few tiles almost no wave beats it -- two
        // thousand same-address atomics in a row were most of this kernel's time
        if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(a.maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.maxbits, bits);
    }
}

// The min-eigen response with the filter radius known at compile time (R = 2 is the reference's FilterSize 5,
// modules/VisualOdometry.m:332): the same three LDS stages and the same integer / double arithmetic as k_response,
// with every tap loop unrolled, the weights in registers, no index division, and the border clamps only in the tiles
// that touch the image border (EDGE).  Thread (tx, ty) owns column tx of rows ty, ty + 4, ...
template <int R, bool EDGE>
__device__ __forceinline__ void response_me_tile(const RespArgs& a, unsigned char* smem, double& bmax) {
    constexpr int SW = RS_TU + 2 * R + 2, SH = RS_TV + 2 * R + 2, PW = RS_TU + 2 * R, PH = RS_TV + 2 * R;
    constexpr int SWP = (SW + 3) & ~3;
    uint8_t* S = smem;
    int32_t* pa = (int32_t*)(smem + ((SWP * SH + 15) & ~15));
    int32_t* pb = pa + PW * PH;
    int32_t* pc = pb + PW * PH;
    int32_t* ha = pc + PW * PH;
    int32_t* hb = ha + RS_TU * PH;
    int32_t* hc = hb + RS_TU * PH;
    const int nu = a.nu, nv = a.nv;
    const int u0 = blockIdx.x * RS_TU, v0 = blockIdx.y * RS_TV;
    const int su0 = u0 - R - 1, sv0 = v0 - R - 1;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int32_t w[2 * R + 1];
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) w[k] = a.w[k];
    // stage: image bytes with halo R + 1 (clamped into the image in edge tiles)
    for (int y = ty; y < SH; y += 4) {
        const int gy = EDGE ? clampi(sv0 + y, 0, nv - 1) : sv0 + y;
        const uint8_t* row = a.img + (long)gy * nu;
        S[y * SWP + tx] = row[EDGE ? clampi(su0 + tx, 0, nu - 1) : su0 + tx];
        if (tx < SW - 64) S[y * SWP + 64 + tx] = row[EDGE ? clampi(su0 + 64 + tx, 0, nu - 1) : su0 + 64 + tx];
    }
    __syncthreads();
    // gradient products at (u0 - R + x, v0 - R + y); S index of image (cu, cv) is (cv - sv0, cu - su0)
    auto products = [&](int x, int y) {
        int du, dv;
        if (EDGE) {
            const int cu = clampi(u0 - R + x, 0, nu - 1), cv = clampi(v0 - R + y, 0, nv - 1);
            const int sy = cv - sv0, sx = cu - su0;
            du = (int)S[sy * SWP + (clampi(cu + 1, 0, nu - 1) - su0)] - (int)S[sy * SWP + (clampi(cu - 1, 0, nu - 1) - su0)];
            dv = (int)S[(clampi(cv + 1, 0, nv - 1) - sv0) * SWP + sx] - (int)S[(clampi(cv - 1, 0, nv - 1) - sv0) * SWP + sx];
        } else {
            const int sy = y + 1, sx = x + 1;
            du = (int)S[sy * SWP + sx + 1] - (int)S[sy * SWP + sx - 1];
            dv = (int)S[(sy + 1) * SWP + sx] - (int)S[(sy - 1) * SWP + sx];
        }
        pa[y * PW + x] = du * du; pb[y * PW + x] = du * dv; pc[y * PW + x] = dv * dv;
    };
    for (int y = ty; y < PH; y += 4) {
        products(tx, y);
        if (tx < PW - 64) products(64 + tx, y);
    }
    __syncthreads();
    // horizontal pass: the column indices do not depend on the row
    int xx[2 * R + 1];
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k)
        xx[k] = EDGE ? clampi(clampi(u0 + tx + k - R, 0, nu - 1) - (u0 - R), 0, PW - 1) : tx + k;
    for (int y = ty; y < PH; y += 4) {
        int32_t sa = 0, sb = 0, sc = 0;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            sa += w[k] * pa[y * PW + xx[k]]; sb += w[k] * pb[y * PW + xx[k]]; sc += w[k] * pc[y * PW + xx[k]];
        }
        ha[y * RS_TU + tx] = sa; hb[y * RS_TU + tx] = sb; hc[y * RS_TU + tx] = sc;
    }
    __syncthreads();
    const int u = u0 + tx;
#pragma unroll
    for (int j = 0; j < RS_TV / 4; ++j) {
        const int y = ty + 4 * j, v = v0 + y;
        if (EDGE && (u >= nu || v >= nv)) continue;
        long long sa = 0, sb = 0, sc = 0;
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const int yy = EDGE ? clampi(clampi(v + k - R, 0, nv - 1) - (v0 - R), 0, PH - 1) : y + k;
            sa += (long long)w[k] * ha[yy * RS_TU + tx]; sb += (long long)w[k] * hb[yy * RS_TU + tx];
            sc += (long long)w[k] * hc[yy * RS_TU + tx];
        }
        const double A = (double)sa, B = (double)sb, C = (double)sc;
        const double dif = A - C;
        const double rad = dif * dif + 4.0 * (B * B);
        double lam = 0.5 * ((A + C) - sqrt(rad));
        if (lam < 0.0) lam = 0.0;
        const double out = lam * (1.0 / 67108864.0);
        a.resp[(long)v * nu + u] = out;
        if (out > bmax) bmax = out;
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_response_me(RespArgs a) {
    constexpr int SW = RS_TU + 2 * R + 2, SH = RS_TV + 2 * R + 2, PW = RS_TU + 2 * R, PH = RS_TV + 2 * R;
    constexpr int SWP = (SW + 3) & ~3;
    __shared__ __align__(16) unsigned char smem[((SWP * SH + 15) & ~15) + 4 * (3 * PW * PH + 3 * RS_TU * PH)];
    const int u0 = blockIdx.x * RS_TU, v0 = blockIdx.y * RS_TV;
    const bool edge = u0 - R - 1 < 0 || v0 - R - 1 < 0 || u0 + RS_TU + R + 1 > a.nu || v0 + RS_TV + R + 1 > a.nv;
    double bmax = 0.0;
    if (edge) response_me_tile<R, true>(a, smem, bmax);
    else response_me_tile<R, false>(a, smem, bmax);
    if (a.maxbits) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(bmax);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            unsigned long long o = __shfl_xor(bits, s, 64);
            bits = o > bits ? o : bits;
        }
        // a plain look first: the word only grows, and after the first few tiles almost no wave beats it -- two
        // thousand same-address atomics in a row were most of this kernel's time
        if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(a.maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.maxbits, bits);
    }
}

// ---- N8 distance filter, the grid.  keep[i] = no existing point with d2 < thr_d2 (thr_d2 = smallest float whose
// correctly rounded sqrtf reaches min_distance, computed on the host).  The CORNERS are bucketed into a uniform grid of
// cell size cs >= min_distance + 1 (counting sort: count -- inside k_detect's write pass --, scan, fill), all of it
// in the image-only half of the selection, off the frame's critical path; once the existing points are known, one
// thread per existing point tests the corners of the 3 x 3 cells around its own and clears keep[] of those it is too
// close to (k_kill).  A pair closer than min_distance differs by less than 1 - 1/cs cells per axis, so the +-1
// neighbourhood is exhaustive even with the rounding of the float cell computation; clamping out-of-image points
// into the border cells only moves them towards the cells that must see them.  The verdict is an "any" over pairs,
// each pair evaluated as the oracle's all-pairs loop evaluates it (corner minus existing point, squares added in the
// same order): bit-exact, whatever the order of the stores.
struct GridArgs {
    float inv_cs; int gu, gv;
    int32_t *count, *offs, *cursor; float *su, *sv; int32_t* si;   // sorted corner positions and indices
};
__device__ __forceinline__ int grid_cell1(float x, float inv_cs, int g) {
    const float c = floorf(x * inv_cs);
    return !(c >= 0.0f) ? 0 : (c > (float)(g - 1) ? g - 1 : (int)c); // NaN -> cell 0
}

// ---- N7 detection.  One block = 1024 consecutive linear indices; pass 0 counts, pass 1 writes.
// One-wave workgroups without LDS for the three kernels that follow the response in the image-only half (round 3): the
// half runs a frame ahead on the side stream and, when it outlasts the tail it starts in, beside the NEXT tracking launch,
// which holds every wave slot and all but 3.6 KB of each CU's LDS.  A block of sixteen waves then waited until the
// tracking launch had drained (k_detect 0.11 ms on average, 1 ms at worst, and the selection's filter half waited for
// the end of the chain in the next tail); a one-wave block takes the first slot a finished keypoint leaves.
constexpr int DT_BLOCK = 64;

__device__ __forceinline__ bool is_corner(const double* __restrict__ resp, int nu, int nv, long i, double thr) {
    const int v = (int)(i / nu), u = (int)(i - (long)v * nu);
    if (u < 1 || u >= nu - 1 || v < 1 || v >= nv - 1) return false;
    const double r = resp[i];
    if (!(r > thr)) return false;
#pragma unroll
    for (int dv = -1; dv <= 1; ++dv)
#pragma unroll
        for (int du = -1; du <= 1; ++du) {
            if (!du && !dv) continue;
            const long j = i + (long)dv * nu + du;
            const double q = resp[j];
            if (q > r || (q == r && j < i)) return false;
        }
    return true;
}

__global__ __launch_bounds__(DT_BLOCK) void k_detect(const double* __restrict__ resp, int nu, int nv,
                                                      const unsigned long long* __restrict__ maxbits,
                                                      double min_quality, int pass, int32_t* blk_counts,
                                                      const int32_t* __restrict__ blk_offs, int cap,
                                                      int32_t* lin, float* lu, float* lv, double* metric,
                                                      GridArgs g, uint8_t* keep) {
    const long n = (long)nu * nv;
    const long i = (long)blockIdx.x * DT_BLOCK + threadIdx.x;
    const double thr = min_quality * __longlong_as_double((long long)*maxbits);
    const bool c = i < n && is_corner(resp, nu, nv, i, thr);
    const unsigned long long bal = __ballot(c);
    const int lane = threadIdx.x;
    if (pass == 0) {
        if (lane == 0) blk_counts[blockIdx.x] = __popcll(bal);
        return;
    }
    if (!c) return;
    const int pos = blk_offs[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
    if (pos >= cap) return;
    const int v = (int)(i / nu), u = (int)(i - (long)v * nu);
    const double r = resp[i];
    const double rum = resp[i - 1], rup = resp[i + 1], rvm = resp[i - nu], rvp = resp[i + nu];
    const double cuu = (rup - 2.0 * r) + rum, cvv = (rvp - 2.0 * r) + rvm;
    double ou = 0, ov = 0;
    if (cuu < 0) { ou = -0.5 * (rup - rum) / cuu; if (ou > 0.5) ou = 0.5; if (ou < -0.5) ou = -0.5; }
    if (cvv < 0) { ov = -0.5 * (rvp - rvm) / cvv; if (ov > 0.5) ov = 0.5; if (ov < -0.5) ov = -0.5; }
    const float fu = (float)((double)u + ou), fv = (float)((double)v + ov);
    lin[pos] = (int32_t)i;
    lu[pos] = fu;
    lv[pos] = fv;
    metric[pos] = r;
    keep[pos] = 1;
    atomicAdd(&g.count[grid_cell1(fv, g.inv_cs, g.gv) * g.gu + grid_cell1(fu, g.inv_cs, g.gu)], 1);
}

// inclusive scan over the wave's 64 lanes without the LDS: four shifts inside the rows of 16 lanes, then lane 15 of every
// odd row's predecessor and lane 31 into the upper half (DPP row_shr / row_bcast; a lane without a source adds the 0 it
// is given as the old value) -- six dependent vector adds instead of six trips through ds_bpermute
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_or_zero(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_scan_incl(int x) {
    x += dpp_or_zero<0x111, 0xf>(x);   // row_shr:1
    x += dpp_or_zero<0x112, 0xf>(x);   // row_shr:2
    x += dpp_or_zero<0x114, 0xf>(x);   // row_shr:4
    x += dpp_or_zero<0x118, 0xf>(x);   // row_shr:8
    x += dpp_or_zero<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}
// exclusive scan of a table of counts by ONE wave, total -> *m0 (clamped to cap): 512 entries a step (eight per lane in
// two 16-byte loads, the next step's loads in flight while this step's wave scan runs), the running total carried in a
// scalar.  15 steps for a KITTI frame's 7.3k blocks or 7.7k grid cells, ~30 instructions each: the launch sits in the
// image-only half that runs beside a tracking launch, where every instruction of a lone wave waits its turn.
__global__ __launch_bounds__(64) void k_scan_counts(const int32_t* __restrict__ counts, int n, int32_t* offs,
                                                     int32_t* m0, int cap) {
    const int lane = threadIdx.x;
    auto load8 = [&](int i, int (&c)[8]) {
        if (i + 7 < n) {
            const int4 v = *reinterpret_cast<const int4*>(counts + i), w = *reinterpret_cast<const int4*>(counts + i + 4);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; c[4] = w.x; c[5] = w.y; c[6] = w.z; c[7] = w.w;
        } else for (int k = 0; k < 8; ++k) c[k] = i + k < n ? counts[i + k] : 0;
    };
    int carry = 0;
    int nxt[8];
    load8(8 * lane, nxt);
    for (int base = 0; base < n; base += 512) {
        const int i = base + 8 * lane;
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = nxt[k];
        if (base + 512 < n) load8(i + 512, nxt);
        int e[8];   // exclusive prefix inside the lane's eight
        e[0] = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) e[k] = e[k - 1] + c[k - 1];
        const int s = e[7] + c[7];
        const int inc = wave_scan_incl(s);
        const int run = carry + inc - s;
        if (i + 7 < n) {
            *reinterpret_cast<int4*>(offs + i) = make_int4(run, run + e[1], run + e[2], run + e[3]);
            *reinterpret_cast<int4*>(offs + i + 4) = make_int4(run + e[4], run + e[5], run + e[6], run + e[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + k < n) offs[i + k] = run + e[k];
        }
        carry += __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 0) *m0 = carry < cap ? carry : cap;
}

// second half of the counting sort: corner i goes to its cell's span of the sorted arrays
__global__ __launch_bounds__(64) void k_grid_fill(const float* __restrict__ cu, const float* __restrict__ cv,
                                                   const int32_t* __restrict__ m0, GridArgs g) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= *m0) return;
    const float u = cu[i], v = cv[i];
    const int c = grid_cell1(v, g.inv_cs, g.gv) * g.gu + grid_cell1(u, g.inv_cs, g.gu);
    const int pos = g.offs[c] + atomicAdd(&g.cursor[c], 1);
    g.su[pos] = u; g.sv[pos] = v; g.si[pos] = i;
}
// one thread per existing point and grid row (blockIdx.y = the row of the 3 x 3 neighbourhood): clears keep[] of every
// corner closer than the threshold.  The launch is a chain of dependent L2 round trips (cell span, positions, index),
// so the chain is kept short: the span's two ends and the count are loaded together, the span is walked four
// corners at a time.  One-wave blocks: the launch sits on the frame's critical path right behind a tracking launch
// that may still hold most of the chip.
// The existing points: two float arrays, or -- the frame loop's state, modules/VisualOdometry.m:330-331
// `single([keypoints; candidate_keypoints])` -- two structure-of-arrays double arrays cast here (ne_dev[0] landmarks of
// pitch capL, then ne_dev[1] candidates of pitch capC) instead of by a launch of their own.
struct ExistSrc { const float *eu, *ev; const double *kpL, *cuC; int capL, capC; };
__global__ __launch_bounds__(64) void k_kill(ExistSrc src, int ne, const int32_t* __restrict__ ne_dev, GridArgs g, float thr_d2, uint8_t* keep) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= (ne_dev ? ne_dev[0] + ne_dev[1] : ne)) return;
    float pu, pv;
    if (src.kpL) {
        const int nL = ne_dev[0];
        if (i < nL) { pu = (float)src.kpL[i]; pv = (float)src.kpL[src.capL + i]; }
        else { const int c = i - nL; pu = (float)src.cuC[c]; pv = (float)src.cuC[src.capC + c]; }
    } else { pu = src.eu[i]; pv = src.ev[i]; }
    const int c0u = grid_cell1(pu, g.inv_cs, g.gu), c0v = grid_cell1(pv, g.inv_cs, g.gv);
    const int cv_ = c0v + (int)blockIdx.y - 1;
    if (cv_ < 0 || cv_ >= g.gv) return;
    const int ua = max(c0u - 1, 0), ub = min(c0u + 1, g.gu - 1);
    // the three cells of one grid row are contiguous in the sorted arrays
    const int j0 = g.offs[cv_ * g.gu + ua];
    const int j1 = g.offs[cv_ * g.gu + ub] + g.count[cv_ * g.gu + ub];
    for (int j = j0; j < j1; j += 4) {
        float du[4], dw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // the sorted arrays are padded (arena): reading past the span is harmless
            du[k] = g.su[j + k] - pu; dw[k] = g.sv[j + k] - pv;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d2 = du[k] * du[k] + dw[k] * dw[k];
            if (j + k < j1 && d2 < thr_d2) keep[g.si[j + k]] = 0;
        }
    }
}

// ---- N8 selectUniform + ordered output.  Single block; cell tables in global scratch.
struct UniArgs {
    const float *cu, *cv; const double* metric; const int32_t* lin; const uint8_t* keep; uint8_t* sel;
    const int32_t* m0; int n_keep, nu, nv, u_is_rows;
    unsigned long long* cell_key; int32_t* cell_idx;
    int32_t *klist, *kcell; // compact list of the kept corners and their cell in the current round
    float *out_u, *out_v; int32_t* out_lin; double* out_metric; int32_t* out_count;
};

constexpr int UNI_LDS_CELLS = 2048; // per-round cell tables live in LDS up to this many cells
#ifndef UNI_NT
#define UNI_NT 1024   // 512 / 256 threads with 32 / 64 corners each were measured: 22 / 80 us slower in the frame loop
#endif
constexpr int UNI_KPT = 16 * 1024 / UNI_NT;   // kept corners a thread can hold in registers across the rounds

// floor((double)c * cells / n) without a division: the product is exact in double (24 + 13 bits),
// a float estimate is corrected by exact integer-valued comparisons.  Identical to the oracle's
// floor of the rounded quotient: a non-integer quotient of such operands stays > 2^-37 (relative)
// away from any integer, far more than the 2^-53 rounding of the division.
__device__ __forceinline__ int cell_exact(float c, int cells, int n, float inv_n) {
    const double tt = (double)c * (double)cells;
    int b = (int)floorf((float)tt * inv_n);
    while ((double)(b + 1) * (double)n <= tt) ++b;
    while ((double)b * (double)n > tt) --b;
    return b;
}

// The same cell through an fp32 estimate when that cannot be wrong: the estimate (c * cells) * (1/n) is within
// 3 * 2^-24 relative of the quotient, i.e. within 4e-4 of it for up to 2048 cells, so a fractional part inside
// [1e-3, 0.999] leaves the floor certain; only estimates next to a cell border pay for the exact form.
__device__ __forceinline__ int cell_of(float c, int cells, int n, float inv_n) {
    const float e = (c * (float)cells) * inv_n;
    const float f = floorf(e);
    const float r = e - f;
    if (cells <= 2048 && r > 1e-3f && r < 0.999f) return (int)f;
    return cell_exact(c, cells, n, inv_n);
}

struct UniShared {
    unsigned long long key[UNI_LDS_CELLS];
    int idx[UNI_LDS_CELLS];
    int cnt[UNI_NT / 64];
    int remaining, selected, round_sel[3], nlist;
#ifdef UNI_TIMING
    unsigned long long t0;
#endif
};

#ifdef UNI_TIMING
#define UNI_T(k) do { if (threadIdx.x == 0 && a.out_metric) a.out_metric[200 + (k)] = (double)(wall_clock64() - sh.t0); } while (0)
#else
#define UNI_T(k) do {} while (0)
#endif

// The rounds of selectUniform with every kept corner in a register slot of one of NTA threads and the cell tables in
// LDS: the rounds touch neither L2 nor HBM.  Called by all NTA threads of the block.
template <int NTA, int KPT>   // KPT: register slots per thread
__device__ __forceinline__ void uniform_fast(const UniArgs& a, int t, int m, int nl, unsigned Hu, unsigned Wu, UniShared& sh) {
    // per corner: coordinates, metric key, and one packed word: index (bits 0-16, corners < 2^17 by
    // the capacity rule n/4+16 is checked on the host), this round's cell (17-27), selected (31)
    float fu[KPT], fv[KPT];
    unsigned long long fk[KPT];
    unsigned fp[KPT];
    constexpr unsigned NONE = 0xffffffffu, IDX = 0x1ffffu, CELL_NONE = 0x7ffu;
    const float inv_nu = 1.0f / (float)a.nu, inv_nv = 1.0f / (float)a.nv;
    const int kmax = (nl + NTA - 1) / NTA;   // slots in use (uniform)
    // two batches of independent loads (list entries, then the gathers through them): written slot by slot behind
    // a branch each, they were sixteen dependent round trips to L2
    int li[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) li[k] = a.klist[min(t + NTA * k, nl - 1)];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = li[k];
        fu[k] = a.cu[i]; fv[k] = a.cv[i];
        fk[k] = (unsigned long long)__double_as_longlong(a.metric[i]) + 1ull;
        fp[k] = (k < kmax && t + NTA * k < nl) ? ((unsigned)i | (CELL_NONE << 17)) : NONE;
    }
    UNI_T(1);
    int rounds = 0, selected = 0, remaining = nl;
    while (selected < a.n_keep && remaining > 0) {
        // the grid of the round as in k_uniform's general path, in 32-bit arithmetic: want < 2^11 and H, W < 2^19 here
        const unsigned want = (unsigned)(a.n_keep - selected);
        unsigned x = (want * Hu) / Wu, h = 0;
        while ((h + 1) * (h + 1) <= x) ++h;
        if (h < 1) h = 1;
        if (h > want) h = want;
        unsigned w = (h * Wu) / Hu; if (w < 1) w = 1; if (h * w > want) w = want / h;
        const int cells_u = (int)(a.u_is_rows ? h : w), cells_v = (int)(a.u_is_rows ? w : h);
        const int nc = cells_u * cells_v;
        for (int k = t; k < nc; k += NTA) { sh.key[k] = 0ull; sh.idx[k] = 0x7fffffff; }
        // three counters in turn: the one of the next round is cleared while this round's is summed and the last
        // round's may still be read by a slow wave
        if (t == 0) sh.round_sel[(rounds + 1) % 3] = 0;
        __syncthreads();
        if (rounds == 0) UNI_T(4);
#pragma unroll
        for (int k = 0; k < KPT; ++k) { // phase 1: per-cell maximum metric
            if (k >= kmax) continue;
            if (fp[k] == NONE || (fp[k] >> 31)) continue;
            const int bu = clampi(cell_of(fu[k], cells_u, a.nu, inv_nu), 0, cells_u - 1);
            const int bv = clampi(cell_of(fv[k], cells_v, a.nv, inv_nv), 0, cells_v - 1);
            const unsigned cell = (unsigned)(bv * cells_u + bu);
            fp[k] = (fp[k] & IDX) | (cell << 17);
            // a plain look first: the table only grows, so a key that does not beat the value seen cannot win,
            // and after the first few updates almost none does -- thousands of corners on a handful of cells
            // (late rounds) would otherwise serialise on the same LDS words
            if (fk[k] > __hip_atomic_load(&sh.key[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(&sh.key[cell], fk[k]);
        }
        __syncthreads();
        if (rounds == 0) UNI_T(5);
#pragma unroll
        for (int k = 0; k < KPT; ++k) { // phase 2: lowest index among the maxima
            if (k >= kmax) continue;
            if (fp[k] == NONE || (fp[k] >> 31)) continue;
            const unsigned cell = (fp[k] >> 17) & CELL_NONE;
            if (sh.key[cell] == fk[k]) atomicMin(&sh.idx[cell], (int)(fp[k] & IDX));
        }
        __syncthreads();
        if (rounds == 0) UNI_T(6);
        int got = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (k >= kmax) continue;
            if (fp[k] == NONE || (fp[k] >> 31)) continue;
            const unsigned cell = (fp[k] >> 17) & CELL_NONE;
            if (sh.idx[cell] == (int)(fp[k] & IDX)) { fp[k] |= 0x80000000u; ++got; }
        }
        if (got) atomicAdd(&sh.round_sel[rounds % 3], got);
        __syncthreads();
        const int rs = sh.round_sel[rounds % 3];
        selected += rs; remaining -= rs;
        if (rounds == 0) UNI_T(7);
        ++rounds;
    }
    UNI_T(2);
#ifdef UNI_TIMING
    if (t == 0 && a.out_metric) { a.out_metric[210] = rounds; a.out_metric[211] = nl; a.out_metric[212] = m; }
#endif
    // output in ascending corner index: the selected corners (<= n_keep <= UNI_LDS_CELLS) go to
    // LDS, each finds its rank by counting the smaller indices
#pragma unroll
    for (int k = 0; k < KPT; ++k)
        if (k < kmax && fp[k] != NONE && (fp[k] >> 31)) sh.idx[atomicAdd(&sh.nlist, 1)] = (int)(fp[k] & IDX);
    __syncthreads();
    const int S = sh.nlist;
    for (int e = t; e < S; e += NTA) {
        const int i = sh.idx[e];
        int pos = 0;
        for (int j = 0; j < S; ++j) pos += sh.idx[j] < i ? 1 : 0;
        a.out_u[pos] = a.cu[i]; a.out_v[pos] = a.cv[i];
        if (a.out_lin) a.out_lin[pos] = a.lin[i];
        if (a.out_metric) a.out_metric[pos] = a.metric[i];
    }
    if (t == 0) *a.out_count = S;
    UNI_T(3);
}

__global__ __launch_bounds__(UNI_NT) void k_uniform(UniArgs a) {
    __shared__ UniShared sh;
    // the frame's critical path runs through this one block while the next frame's image work (side stream) shares
    // its CU: issue priority over those waves
    __builtin_amdgcn_s_setprio(3);
    int* const s_cnt = sh.cnt; int* const s_idx = sh.idx; unsigned long long* const s_key = sh.key;
    int &s_selected = sh.selected, &s_remaining = sh.remaining, &s_round_sel = sh.round_sel[0], &s_nlist = sh.nlist;
#ifdef UNI_TIMING
    if (threadIdx.x == 0) sh.t0 = wall_clock64();
    __syncthreads();
#endif
    const int t = threadIdx.x, m = *a.m0;
    UNI_T(0);
    // The corners that survived the distance filter, as a compact list in ascending index: 16 flags per thread in one
    // load, block-wide exclusive scan of the counts.  (k_kill only clears flags: it cannot know which corners are left
    // before its last store.)
    int nl = 0;
    {
        const int lane = t & 63, wid = t >> 6;
        for (int start = 0; start < m; start += 16 * UNI_NT) {
            const int i0 = start + 16 * t;
            unsigned bits = 0;
            if (i0 < m) {   // keep[] is 256-byte aligned and padded (arena): the 16 bytes are there
                const uint4 q = *reinterpret_cast<const uint4*>(a.keep + i0);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (i0 + k < m && ((w4[k >> 2] >> (8 * (k & 3))) & 0xffu)) bits |= 1u << k;
            }
            const int c = __popc(bits);
            int inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
            if (lane == 63) s_cnt[wid] = inc;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w2 = 0; w2 < UNI_NT / 64; ++w2) { const int c2 = s_cnt[w2]; if (w2 < wid) before += c2; total += c2; }
            int pos = nl + before + inc - c;
            while (bits) { const int k = __ffs((int)bits) - 1; bits &= bits - 1u; a.klist[pos++] = i0 + k; }
            nl += total;
            __syncthreads();
        }
    }
    if (t == 0) { s_selected = 0; s_remaining = nl; s_nlist = 0; sh.round_sel[0] = 0; }
    __threadfence_block();
    __syncthreads();
    const long H = a.u_is_rows ? a.nu : a.nv, W = a.u_is_rows ? a.nv : a.nu;
    const float inv_nu = 1.0f / (float)a.nu, inv_nv = 1.0f / (float)a.nv;
    // A round has at most `want` <= n_keep cells.  When the cell tables fit LDS and a thread's share
    // of the kept corners fits its registers, the rounds touch neither L2 nor HBM.
    // asking for at least as many points as survive selects them all (bootstrap: selectUniform over every
    // detected corner, modules/bootstrap.m:36-38) -- the rounds would end there too
    const bool all = a.n_keep >= nl;
    const bool fast = !all && nl <= UNI_NT * UNI_KPT && a.n_keep <= UNI_LDS_CELLS - 1 && m <= 0x1ffff && H < (1 << 19) && W < (1 << 19);
    if (fast) {
        // (fewer waves with more slots each were measured for short lists -- 256 threads x 6 slots at 1300 corners:
        // a round is bound by the dependent instruction chain of a slot, ~500 cycles for a wave alone on its SIMD, and
        // four waves a SIMD hide most of that for each other: 2.4 us a round with 1024 threads, 2.9 us with 256)
        uniform_fast<UNI_NT, UNI_KPT>(a, t, m, nl, (unsigned)H, (unsigned)W, sh);
        return;
    }
    if (all && nl <= UNI_LDS_CELLS) {
        // every surviving corner is selected and there are few of them (the usual case beside tens of thousands of
        // existing points): the compact list goes through LDS, each entry finds its rank by counting the smaller
        // indices -- ascending corner index, as the ordered compaction below produces
        for (int e = t; e < nl; e += UNI_NT) s_idx[e] = a.klist[e];
        __syncthreads();
        for (int e = t; e < nl; e += UNI_NT) {
            const int i = s_idx[e];
            int pos = 0;
            for (int j = 0; j < nl; ++j) pos += s_idx[j] < i ? 1 : 0;
            a.out_u[pos] = a.cu[i]; a.out_v[pos] = a.cv[i];
            if (a.out_lin) a.out_lin[pos] = a.lin[i];
            if (a.out_metric) a.out_metric[pos] = a.metric[i];
        }
        if (t == 0) *a.out_count = nl;
        return;
    }
    for (int i = t; i < m; i += UNI_NT) a.sel[i] = all ? a.keep[i] : 0;
    __threadfence_block();
    __syncthreads();
    while (!all) {
        const int selected = s_selected, remaining = s_remaining;
        if (!(selected < a.n_keep && remaining > 0)) break;
        const long want = a.n_keep - selected;
        long x = (want * H) / W, h = 0;
        while ((h + 1) * (h + 1) <= x) ++h;
        if (h < 1) h = 1;
        if (h > want) h = want;
        long w = (h * W) / H; if (w < 1) w = 1; if (h * w > want) w = want / h;
        const int cells_u = (int)(a.u_is_rows ? h : w), cells_v = (int)(a.u_is_rows ? w : h);
        const int nc = cells_u * cells_v;
        const bool lds = nc <= UNI_LDS_CELLS; // wave-uniform: LDS atomics when the table fits, else L2 atomics
        for (int k = t; k < nc; k += UNI_NT) {
            if (lds) { s_key[k] = 0ull; s_idx[k] = 0x7fffffff; }
            else { a.cell_key[k] = 0ull; a.cell_idx[k] = 0x7fffffff; }
        }
        if (t == 0) s_round_sel = 0;
        __syncthreads();
        // phase 1: per-cell maximum metric (+1 so that a zero metric still differs from "empty")
        for (int e = t; e < nl; e += UNI_NT) {
            const int i = a.klist[e];
            if (a.sel[i]) { a.kcell[e] = -1; continue; }
            const int bu = clampi(cell_of(a.cu[i], cells_u, a.nu, inv_nu), 0, cells_u - 1);
            const int bv = clampi(cell_of(a.cv[i], cells_v, a.nv, inv_nv), 0, cells_v - 1);
            const int cell = bv * cells_u + bu;
            a.kcell[e] = cell;
            const unsigned long long key = (unsigned long long)__double_as_longlong(a.metric[i]) + 1ull;
            if (lds) { if (key > __hip_atomic_load(&s_key[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(&s_key[cell], key); }
            else if (key > __hip_atomic_load(&a.cell_key[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.cell_key[cell], key);
        }
        __threadfence_block();
        __syncthreads();
        // phase 2: lowest index among the maxima
        for (int e = t; e < nl; e += UNI_NT) {
            const int cell = a.kcell[e];
            if (cell < 0) continue;
            const int i = a.klist[e];
            const unsigned long long key = (unsigned long long)__double_as_longlong(a.metric[i]) + 1ull;
            if (lds) { if (s_key[cell] == key) atomicMin(&s_idx[cell], i); }
            else if (__hip_atomic_load(&a.cell_key[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key)
                atomicMin(&a.cell_idx[cell], i);
        }
        __threadfence_block();
        __syncthreads();
        int got = 0;
        for (int k = t; k < nc; k += UNI_NT) {
            const int idx = lds ? s_idx[k] : __hip_atomic_load(&a.cell_idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx != 0x7fffffff) { a.sel[idx] = 1; ++got; }
        }
        if (got) atomicAdd(&s_round_sel, got);
        __threadfence_block();
        __syncthreads();
        if (t == 0) { s_selected += s_round_sel; s_remaining -= s_round_sel; }
        __syncthreads();
    }
    // ordered compaction of sel[] (ascending corner index)
    int base = 0;
    const int lane = t & 63, wid = t >> 6;
    for (int start = 0; start < m; start += UNI_NT) {
        const int i = start + t;
        const bool s = i < m && a.sel[i];
        const unsigned long long bal = __ballot(s);
        if (lane == 0) s_cnt[wid] = __popcll(bal);
        __syncthreads();
        int pos = base, tot = 0;
        for (int w2 = 0; w2 < UNI_NT / 64; ++w2) { if (w2 < wid) pos += s_cnt[w2]; tot += s_cnt[w2]; }
        pos += __popcll(bal & ((1ull << lane) - 1ull));
        if (s) {
            a.out_u[pos] = a.cu[i]; a.out_v[pos] = a.cv[i];
            if (a.out_lin) a.out_lin[pos] = a.lin[i];
            if (a.out_metric) a.out_metric[pos] = a.metric[i];
        }
        base += tot;
        __syncthreads();
    }
    if (t == 0) *a.out_count = base;
}

// ---- old/selectKeypoints.m:5-14 greedy selection: per iteration, block partial arg-max (first
// maximum in linear order), then one block reduces the partials, records the pick and zeroes the
// (2r+1)^2 box.
constexpr int GR_BLOCK = 256, GR_PER = 16;
__global__ __launch_bounds__(GR_BLOCK) void k_greedy_partial(const double* __restrict__ sc, long n,
                                                             unsigned long long* pv, int32_t* pi) {
    __shared__ unsigned long long sv[GR_BLOCK / 64];
    __shared__ int si[GR_BLOCK / 64];
    const long base = (long)blockIdx.x * GR_BLOCK * GR_PER;
    unsigned long long bv = 0; int bi = 0x7fffffff;
    for (int k = 0; k < GR_PER; ++k) {
        const long i = base + (long)k * GR_BLOCK + threadIdx.x;
        if (i < n) {
            const double d = sc[i];
            const unsigned long long b = d > 0.0 ? (unsigned long long)__double_as_longlong(d) : 0ull;
            if (b > bv) { bv = b; bi = (int)i; } // ascending i per thread keeps the first maximum
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long ov = __shfl_xor(bv, s, 64);
        const int oi = __shfl_xor(bi, s, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < GR_BLOCK / 64; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        pv[blockIdx.x] = bv; pi[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(1024) void k_greedy_pick(double* sc, int nu, int nv, const unsigned long long* pv,
                                                      const int32_t* pi, int nblk, int r, int it, int32_t* out_uv) {
    __shared__ unsigned long long sv[16];
    __shared__ int si[16];
    __shared__ int s_best;
    unsigned long long bv = 0; int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < nblk; k += 1024) {
        const unsigned long long v = pv[k]; const int i = pi[k];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long ov = __shfl_xor(bv, s, 64);
        const int oi = __shfl_xor(bi, s, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        // all-zero image: MATLAB's max() returns the first element of the zero-padded array
        int ku, kv;
        if (bv == 0ull) { ku = -r; kv = -r; s_best = -1; }
        else { kv = bi / nu; ku = bi - kv * nu; s_best = bi; }
        out_uv[2 * it] = ku; out_uv[2 * it + 1] = kv;
    }
    __syncthreads();
    const int b = s_best;
    int ku, kv;
    if (b < 0) { ku = -r; kv = -r; } else { kv = b / nu; ku = b - kv * nu; }
    const int side = 2 * r + 1;
    for (int c = threadIdx.x; c < side * side; c += 1024) {
        const int dv = c / side - r, du = c % side - r;
        const int x = ku + du, y = kv + dv;
        if (x >= 0 && x < nu && y >= 0 && y < nv) sc[(long)y * nu + x] = 0.0;
    }
}

} // namespace

static void gauss_weights_host(int fs, int32_t* w) { // N6, identical to oracle/select.c
    int r = fs / 2;
    double s = (double)fs / 3.0, sum = 0, g[64];
    for (int k = -r; k <= r; ++k) { g[k + r] = exp(-(double)(k * k) / (2.0 * s * s)); sum += g[k + r]; }
    for (int k = 0; k < fs; ++k) w[k] = (int32_t)lround(4096.0 * g[k] / sum);
}

int vo_launch_response(vo_ctx* ctx, hipStream_t stream, const uint8_t* d_img, int nu, int nv, int scorer, int fs,
                       double kappa, double* d_resp, unsigned long long* d_maxbits, unsigned* d_zero, long zero_words) {
    if (fs < 1 || fs > 15 || !(fs & 1)) { ctx->last_error = "select: filter_size must be odd in [1,15]"; return VO_ERR_BAD_ARG; }
    if (scorer != VO_SCORER_MINEIGEN && scorer != VO_SCORER_HARRIS) { ctx->last_error = "select: unknown scorer"; return VO_ERR_BAD_ARG; }
    if (nu < 3 || nv < 3) { ctx->last_error = "select: image smaller than 3x3"; return VO_ERR_BAD_ARG; }
    RespArgs a;
    a.img = d_img; a.nu = nu; a.nv = nv; a.r = fs / 2; a.scorer = scorer; a.kappa = kappa;
    a.resp = d_resp; a.maxbits = d_maxbits;
    for (int k = 0; k < 16; ++k) a.w[k] = 1;
    if (scorer == VO_SCORER_MINEIGEN) gauss_weights_host(fs, a.w);
    const int r = a.r;
    const int SW = RS_TU + 2 * r + 2, SH = RS_TV + 2 * r + 2, PW = RS_TU + 2 * r, PH = RS_TV + 2 * r;
    size_t lds = ((size_t)(SW * SH + 15) & ~(size_t)15) + sizeof(int32_t) * (3 * (size_t)PW * PH + 3 * (size_t)RS_TU * PH);
    if (d_maxbits || zero_words > 0)
        hipLaunchKernelGGL(k_sel_zero, dim3((unsigned)((std::max(zero_words, 1L) + 255) / 256)), dim3(256), 0, stream, d_maxbits, d_zero, zero_words);
    dim3 grid((nu + RS_TU - 1) / RS_TU, (nv + RS_TV - 1) / RS_TV);
    if (!ctx->resp_attr_set) {
        VO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_response, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        ctx->resp_attr_set = true;
    }
    if (scorer == VO_SCORER_MINEIGEN && r == 2) {   // the reference's configuration: the unrolled kernel
        hipLaunchKernelGGL(k_response_me<2>, grid, dim3(256), 0, stream, a);
        VO_HIP_CHECK(ctx, hipGetLastError());
        return VO_OK;
    }
    hipLaunchKernelGGL(k_response, grid, dim3(256), lds, stream, a);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

// smallest float x with sqrtf(x) >= d (sqrtf is correctly rounded and monotone on the host)
static float d2_threshold(float d) {
    if (!(d > 0.0f)) return 0.0f;
    float x = d * d;
    while (sqrtf(x) >= d && x > 0.0f) x = nextafterf(x, 0.0f);
    while (sqrtf(x) < d) x = nextafterf(x, INFINITY);
    return x;
}

static GridArgs grid_args(const vo_sel_work* w) {
    GridArgs g;
    g.inv_cs = 1.0f / w->grid_cs; g.gu = w->grid_gu; g.gv = w->grid_gv;
    g.count = w->grid_count; g.offs = w->grid_offs; g.cursor = w->grid_cursor; g.su = w->grid_su; g.sv = w->grid_sv; g.si = w->grid_si;
    return g;
}

void vo_select_grid_dims(int nu, int nv, double min_distance, float* cs, int* gu, int* gv) {
    float c = (float)min_distance + 1.0f;
    if (!(c >= 2.0f)) c = 2.0f;
    // keep the table small for tiny distances on big images
    while ((double)(nu / c + 2) * (double)(nv / c + 2) > 4.0e6) c *= 2.0f;
    *cs = c; *gu = (int)(nu / c) + 2; *gv = (int)(nv / c) + 2;
}

int vo_launch_select_detect(vo_ctx* ctx, hipStream_t stream, const uint8_t* d_img, int nu, int nv, const vo_sel_opts* o,
                            vo_sel_work* w) {
    if (o->keep < 0) { ctx->last_error = "select: negative count"; return VO_ERR_BAD_ARG; }
    // the corner grid's tables (count + cursor) start at zero: cleared by the launch that clears the running maximum
    const long grid_words = 2 * (long)w->grid_gu * w->grid_gv;
    int rc = vo_launch_response(ctx, stream, d_img, nu, nv, o->scorer, o->filter_size, o->kappa, w->resp, w->maxbits,
                                (unsigned*)w->grid_count, grid_words);
    if (rc) return rc;
    const long n = (long)nu * nv;
    const int nblk = (int)((n + DT_BLOCK - 1) / DT_BLOCK);
    if (nblk > w->nblk) { ctx->last_error = "select: scratch too small"; return VO_ERR_BAD_ARG; }
    const GridArgs g = grid_args(w);
    hipLaunchKernelGGL(k_detect, dim3(nblk), dim3(DT_BLOCK), 0, stream, w->resp, nu, nv, w->maxbits,
                       o->min_quality, 0, w->blk_counts, w->blk_offs, w->cap, w->lin, w->lu, w->lv, w->metric, g, w->keep);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(64), 0, stream, w->blk_counts, nblk, w->blk_offs, w->m0, w->cap);
    hipLaunchKernelGGL(k_detect, dim3(nblk), dim3(DT_BLOCK), 0, stream, w->resp, nu, nv, w->maxbits,
                       o->min_quality, 1, w->blk_counts, w->blk_offs, w->cap, w->lin, w->lu, w->lv, w->metric, g, w->keep);
    // the corner grid of the distance filter: counted by the write pass above, scanned and filled here
    const int ncell = g.gu * g.gv;
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(64), 0, stream, g.count, ncell, g.offs, w->grid_total, 0x7fffffff);
    hipLaunchKernelGGL(k_grid_fill, dim3((w->cap + 63) / 64), dim3(64), 0, stream, w->lu, w->lv, w->m0, g);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_select_filter(vo_ctx* ctx, hipStream_t stream, int nu, int nv, int u_is_rows, const float* d_eu,
                            const float* d_ev, int n_exist, const int32_t* ne_dev, const vo_sel_opts* o, vo_sel_work* w,
                            float* d_out_u, float* d_out_v, int32_t* d_out_lin, double* d_out_metric,
                            int32_t* d_out_count, const double* d_kpL, int capL, const double* d_cuC, int capC, hipEvent_t after_kill) {
    if (o->keep < 0 || n_exist < 0) { ctx->last_error = "select: negative count"; return VO_ERR_BAD_ARG; }
    if (d_kpL && (!ne_dev || !d_cuC)) { ctx->last_error = "select: the state form of the existing points needs device counts"; return VO_ERR_BAD_ARG; }
    const float thr_d2 = d2_threshold((float)o->min_distance);
    if (n_exist > 0 && thr_d2 > 0.0f) {  // otherwise every corner keeps the flag the detect half gave it
        const ExistSrc src = {d_eu, d_ev, d_kpL, d_cuC, capL, capC};
        hipLaunchKernelGGL(k_kill, dim3((n_exist + 63) / 64, 3), dim3(64), 0, stream, src, n_exist, ne_dev, grid_args(w), thr_d2, w->keep);
    }
    if (after_kill) VO_HIP_CHECK(ctx, hipEventRecord(after_kill, stream));
    UniArgs a;
    a.cu = w->lu; a.cv = w->lv; a.metric = w->metric; a.lin = w->lin; a.keep = w->keep; a.sel = w->sel; a.m0 = w->m0;
    a.n_keep = o->keep; a.nu = nu; a.nv = nv; a.u_is_rows = u_is_rows;
    a.cell_key = w->cell_key; a.cell_idx = w->cell_idx; a.klist = w->klist; a.kcell = w->kcell;
    a.out_u = d_out_u; a.out_v = d_out_v; a.out_lin = d_out_lin; a.out_metric = d_out_metric; a.out_count = d_out_count;
    hipLaunchKernelGGL(k_uniform, dim3(1), dim3(UNI_NT), 0, stream, a);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_select(vo_ctx* ctx, const uint8_t* d_img, int nu, int nv, int u_is_rows, const float* d_eu,
                     const float* d_ev, int n_exist, const vo_sel_opts* o, vo_sel_work* w, float* d_out_u,
                     float* d_out_v, int32_t* d_out_lin, double* d_out_metric, int32_t* d_out_count) {
    int rc = vo_launch_select_detect(ctx, ctx->stream, d_img, nu, nv, o, w);
    if (rc) return rc;
    return vo_launch_select_filter(ctx, ctx->stream, nu, nv, u_is_rows, d_eu, d_ev, n_exist, nullptr, o, w, d_out_u, d_out_v,
                                   d_out_lin, d_out_metric, d_out_count, nullptr, 0, nullptr, 0);
}

int vo_greedy_num_blocks(long n) { return (int)((n + (long)GR_BLOCK * GR_PER - 1) / ((long)GR_BLOCK * GR_PER)); }

int vo_launch_greedy(vo_ctx* ctx, double* d_scores_work, int nu, int nv, int num, int r, int32_t* d_out_uv,
                     unsigned long long* d_tmp) {
    if (num < 0 || r < 0) { ctx->last_error = "greedy: negative argument"; return VO_ERR_BAD_ARG; }
    const long n = (long)nu * nv;
    const int nblk = vo_greedy_num_blocks(n);
    unsigned long long* pv = d_tmp;
    int32_t* pi = (int32_t*)(d_tmp + nblk);
    for (int it = 0; it < num; ++it) {
        hipLaunchKernelGGL(k_greedy_partial, dim3(nblk), dim3(GR_BLOCK), 0, ctx->stream, d_scores_work, n, pv, pi);
        hipLaunchKernelGGL(k_greedy_pick, dim3(1), dim3(1024), 0, ctx->stream, d_scores_work, nu, nv, pv, pi, nblk, r, it, d_out_uv);
    }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
