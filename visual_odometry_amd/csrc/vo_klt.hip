// vo_klt.hip -- pyramidal Lucas-Kanade with forward-backward check, one wavefront per keypoint.
// Replaces vision.PointTracker as driven by KLTTracker.track (modules/KLTTracker.m:22-41; called
// at modules/VisualOdometry.m:71,278).  Normative algorithm: oracle/vo_oracle.h N3-N5.
//
// Mapping: one 64-lane workgroup (= one wave) owns one keypoint for the forward and the backward
// pass.  The wu x wv window is dealt out lane-cyclically (pixel id = k*64 + lane); the template
// (T, Tu, Tv) lives in registers for the whole level; the search patch of the other image is
// staged once into LDS with a margin of LK_M pixels and re-staged only when the estimate walks out
// of it, so an iteration reads nothing but LDS: 4 taps per pixel with weights shared by the window.
// Bound: FP32 VALU + LDS bandwidth, HBM traffic ~ the pyramids only (L2/MALL hits).
// Algorithmic flops: set-up 24 flop/px, iteration 11 flop/px (counters[0], [2] count them).
#include "vo_common.h"

namespace {

constexpr int LK_M = 3; // LDS patch margin in pixels

struct LKArgs {
    const float *AI, *AGu, *AGv, *BI, *BGu, *BGv;
    int nus[VO_MAX_LEVELS], nvs[VO_MAX_LEVELS];
    long offs[VO_MAX_LEVELS];
    int levels, wu, wv, PW, PH, npx, max_iters, n;
    double eps2, max_bidir;
    const double *pu, *pv;
    double *qu, *qv;
    uint8_t* valid;
    float *score, *fb;
    unsigned long long* counters;
};

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

__device__ __forceinline__ void stage(float* patch, const float* __restrict__ img, int nu, int nv, int ou,
                                      int ov, int PW, int PH, int lane) {
    __syncthreads();
    for (int r = 0; r < PH; ++r) {
        const float* row = img + (long)clampi(ov + r, 0, nv - 1) * nu;
        for (int c = lane; c < PW; c += 64) patch[r * PW + c] = row[clampi(ou + c, 0, nu - 1)];
    }
    __syncthreads();
}

struct Bil {
    float w00, w10, w01, w11;
    int base;
};
__device__ __forceinline__ Bil make_bil(float fu, float fv, int fi, int fj, int PW) {
    Bil b;
    b.w00 = (1.0f - fu) * (1.0f - fv); b.w10 = fu * (1.0f - fv);
    b.w01 = (1.0f - fu) * fv; b.w11 = fu * fv;
    b.base = fj * PW + fi;
    return b;
}
__device__ __forceinline__ float tap(const float* patch, int o, int PW, const Bil& b) {
    return b.w00 * patch[o] + b.w10 * patch[o + 1] + b.w01 * patch[o + PW] + b.w11 * patch[o + PW + 1];
}

// one direction A -> B (N4).  returns flags: bit0 lost, bit1 flat at level 0
template <int K>
__device__ int lk_dir(const LKArgs& a, const float* AI, const float* AGu, const float* AGv, const float* BI,
                      double su, double sv, float* patch, const int (&off)[K], int lane, double& qu,
                      double& qv, float& ssd_out, unsigned& iters, unsigned& setups) {
    float T[K], Tu[K], Tv[K];
    const int PW = a.PW, PH = a.PH, ru = (a.wu - 1) / 2, rv = (a.wv - 1) / 2;
    const int nu0 = a.nus[0], nv0 = a.nvs[0];
    double gu = 0, gv = 0;
    int flags = 0;
    for (int l = a.levels - 1; l >= 0; --l) {
        const double s = (double)(1 << l);
        const int nu = a.nus[l], nv = a.nvs[l];
        const long lo = a.offs[l];
        const double plu = su / s, plv = sv / s;
        {
            const double flu = floor(plu), flv = floor(plv);
            const int iu = (int)flu, iv = (int)flv;
            const Bil b = make_bil((float)(plu - flu), (float)(plv - flv), LK_M, LK_M, PW);
            const int ou = iu - ru - LK_M, ov = iv - rv - LK_M;
            stage(patch, AI + lo, nu, nv, ou, ov, PW, PH, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) T[k] = tap(patch, off[k] + b.base, PW, b);
            stage(patch, AGu + lo, nu, nv, ou, ov, PW, PH, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) Tu[k] = (k * 64 + lane < a.npx) ? tap(patch, off[k] + b.base, PW, b) : 0.0f;
            stage(patch, AGv + lo, nu, nv, ou, ov, PW, PH, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) Tv[k] = (k * 64 + lane < a.npx) ? tap(patch, off[k] + b.base, PW, b) : 0.0f;
            ++setups;
        }
        float sxx = 0, sxy = 0, syy = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) { sxx += Tu[k] * Tu[k]; sxy += Tu[k] * Tv[k]; syy += Tv[k] * Tv[k]; }
        const double Guu = wave_sum((double)sxx), Guv = wave_sum((double)sxy), Gvv = wave_sum((double)syy);
        const double dif = Guu - Gvv;
        const double mineig = ((Guu + Gvv) - sqrt(dif * dif + 4.0 * (Guv * Guv))) / (2.0 * (double)a.npx);
        const double det = Guu * Gvv - Guv * Guv;
        double du = 0, dv = 0;
        const bool flat = !(mineig >= (double)VO_MIN_EIG);
        if (flat) {
            if (l == 0) flags |= 2;
        } else {
            bool staged = false;
            int pou = 0, pov = 0;
            for (int it = 0; it < a.max_iters; ++it) {
                const double cu = plu + gu + du, cv = plv + gv + dv;
                if (!(cu * s >= 0.0 && cu * s <= (double)(nu0 - 1) && cv * s >= 0.0 && cv * s <= (double)(nv0 - 1))) {
                    flags |= 1;
                    break;
                }
                const double flu = floor(cu), flv = floor(cv);
                const int iu = (int)flu, iv = (int)flv;
                int fi = iu - ru - pou, fj = iv - rv - pov;
                if (!staged || fi < 0 || fi > 2 * LK_M || fj < 0 || fj > 2 * LK_M) {
                    pou = iu - ru - LK_M; pov = iv - rv - LK_M;
                    stage(patch, BI + lo, nu, nv, pou, pov, PW, PH, lane);
                    staged = true;
                    fi = LK_M; fj = LK_M;
                }
                const Bil b = make_bil((float)(cu - flu), (float)(cv - flv), fi, fj, PW);
                float bu = 0, bv = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float e = T[k] - tap(patch, off[k] + b.base, PW, b);
                    bu += e * Tu[k];
                    bv += e * Tv[k];
                }
                const double Bu = wave_sum((double)bu), Bv = wave_sum((double)bv);
                ++iters;
                const double ddu = (Gvv * Bu - Guv * Bv) / det;
                const double ddv = (Guu * Bv - Guv * Bu) / det;
                du += ddu; dv += ddv;
                if (ddu * ddu + ddv * ddv < a.eps2) break;
            }
        }
        if (flags & 1) break;
        if (l > 0) { gu = 2.0 * (gu + du); gv = 2.0 * (gv + dv); }
        else { gu = gu + du; gv = gv + dv; }
    }
    qu = su + gu; qv = sv + gv;
    if (!(flags & 1)) {
        if (!(qu >= 0.0 && qu <= (double)(nu0 - 1) && qv >= 0.0 && qv <= (double)(nv0 - 1))) flags |= 1;
    }
    ssd_out = 0.0f;
    if (!(flags & 1)) { // N5 residual at level 0; T holds the level-0 template
        const double flu = floor(qu), flv = floor(qv);
        const int iu = (int)flu, iv = (int)flv;
        stage(patch, BI, nu0, nv0, iu - ru - LK_M, iv - rv - LK_M, PW, PH, lane);
        const Bil b = make_bil((float)(qu - flu), (float)(qv - flv), LK_M, LK_M, PW);
        float ssd = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float e = T[k] - tap(patch, off[k] + b.base, PW, b);
            ssd += (k * 64 + lane < a.npx) ? e * e : 0.0f;
        }
        ssd_out = (float)wave_sum((double)ssd);
    }
    return flags;
}

template <int K>
__global__ __launch_bounds__(64) void k_lk(LKArgs a) {
    extern __shared__ float patch[];
    const int lane = threadIdx.x, kp = blockIdx.x;
    int off[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int id = k * 64 + lane;
        const int j = id / a.wu, i = id - j * a.wu;
        off[k] = id < a.npx ? j * a.PW + i : 0;
    }
    const double p0u = a.pu[kp], p0v = a.pv[kp];
    unsigned iters = 0, setups = 0;
    double fu, fv, bu = 0, bv = 0;
    float ssd, ssd2;
    const int fl = lk_dir<K>(a, a.AI, a.AGu, a.AGv, a.BI, p0u, p0v, patch, off, lane, fu, fv, ssd, iters, setups);
    bool ok = (fl == 0);
    float fb = __builtin_nanf("");
    if (!(fl & 1) && a.max_bidir >= 0.0) {
        const int fl2 = lk_dir<K>(a, a.BI, a.BGu, a.BGv, a.AI, fu, fv, patch, off, lane, bu, bv, ssd2, iters, setups);
        if (fl2 & 1) ok = false;
        else {
            const double eu = p0u - bu, ev = p0v - bv;
            const double d = sqrt(eu * eu + ev * ev);
            fb = (float)d;
            if (!(d <= a.max_bidir)) ok = false;
        }
    }
    if (lane == 0) {
        a.qu[kp] = fu; a.qv[kp] = fv;
        a.valid[kp] = ok ? 1 : 0;
        float sc = 0.0f;
        if (!(fl & 1)) {
            sc = 1.0f - sqrtf(ssd / (float)a.npx) / 255.0f;
            sc = fminf(fmaxf(sc, 0.0f), 1.0f);
        }
        a.score[kp] = sc;
        if (a.fb) a.fb[kp] = fb;
        atomicAdd(&a.counters[0], (unsigned long long)iters);
        atomicAdd(&a.counters[1], 1ull);
        atomicAdd(&a.counters[2], (unsigned long long)setups);
    }
}

} // namespace

int vo_launch_klt(vo_ctx* ctx, const vo_pyr* A, const vo_pyr* B, const double* d_pu, const double* d_pv, int n,
                  int wu, int wv, int max_iters, double eps, double max_bidir, double* d_qu, double* d_qv,
                  uint8_t* d_valid, float* d_score, float* d_fb) {
    if (n <= 0) return VO_OK;
    if (!(wu & 1) || !(wv & 1) || wu < 3 || wv < 3 || max_iters < 1) { ctx->last_error = "klt: BlockSize must be odd >= 3"; return VO_ERR_BAD_ARG; }
    if (A->nu != B->nu || A->nv != B->nv || A->levels != B->levels) { ctx->last_error = "klt: frame shapes differ"; return VO_ERR_BAD_ARG; }
    LKArgs a;
    a.AI = A->I; a.AGu = A->Gu; a.AGv = A->Gv; a.BI = B->I; a.BGu = B->Gu; a.BGv = B->Gv;
    for (int l = 0; l < A->levels; ++l) { a.nus[l] = A->nus[l]; a.nvs[l] = A->nvs[l]; a.offs[l] = A->offs[l]; }
    a.levels = A->levels; a.wu = wu; a.wv = wv; a.PW = wu + 1 + 2 * LK_M; a.PH = wv + 1 + 2 * LK_M;
    a.npx = wu * wv; a.max_iters = max_iters; a.n = n;
    a.eps2 = eps * eps; a.max_bidir = max_bidir;
    a.pu = d_pu; a.pv = d_pv; a.qu = d_qu; a.qv = d_qv; a.valid = d_valid; a.score = d_score; a.fb = d_fb;
    a.counters = ctx->d_counters;
    const size_t lds = (size_t)a.PW * a.PH * sizeof(float);
    const int K = (a.npx + 63) / 64;
    dim3 grid(n), block(64);
    if (K <= 8) hipLaunchKernelGGL(k_lk<8>, grid, block, lds, ctx->stream, a);
    else if (K <= 16) hipLaunchKernelGGL(k_lk<16>, grid, block, lds, ctx->stream, a);
    else if (K <= 27) hipLaunchKernelGGL(k_lk<27>, grid, block, lds, ctx->stream, a);
    else if (K <= 41) hipLaunchKernelGGL(k_lk<41>, grid, block, lds, ctx->stream, a);
    else { ctx->last_error = "klt: BlockSize larger than 51x51 is not supported"; return VO_ERR_UNSUPPORTED; }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
