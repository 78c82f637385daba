// vo_image.hip -- image-wide stencil of the KLT front end: the float pyramid (oracle N1).
// Replaces the pyramid set-up inside vision.PointTracker (modules/KLTTracker.m:37-39, initialize +
// step).  Gradients are NOT stored: the tracker takes the Sobel operator of the sub-pixel-sampled
// template (vo_klt.hip), so a frame is one float plane per level.
// HBM-bound streaming kernels: LDS-tiled, coalesced along the fast (u) axis.
//
// Algorithmic bytes per image (KITTI 1241x376, 4 levels, P = 619,930 px): read 1*WH (u8) + write
// 4*P (float I) + re-read 4*(P - P_last) for decimation ~= 5.4 MB; the replicated border of VO_PYR_PAD
// pixels per level (vo_common.h) adds 4 * 268,000 bytes of writes (+ 1.1 MB).
#include "vo_common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

constexpr int PAD = VO_PYR_PAD;

// ---- level 0: uint8 -> float over the padded plane (border replicated), one row segment of 4 pixels per thread ----
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ img, int nu, int nv, float* __restrict__ I) {
    const int pitch = nu + 2 * PAD;
    const int up = (blockIdx.x * 256 + threadIdx.x) * 4, vp = blockIdx.y;   // padded coordinates
    if (up >= pitch) return;
    const uint8_t* row = img + (long)clampi(vp - PAD, 0, nv - 1) * nu;
    float* out = I + (long)vp * pitch + up;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (up + k < pitch) out[k] = (float)row[clampi(up + k - PAD, 0, nu - 1)];
}

// ---- level l >= 1: 5x5 binomial decimation of level l-1 (horizontal pass first, then vertical, N1).
// out tile 64 x 8; staged source region (2*64+3) x (2*8+3) at replicate-clamped coordinates ----
constexpr int PD_TU = 64, PD_TV = 8;
constexpr int PD_SW = 2 * PD_TU + 3, PD_SH = 2 * PD_TV + 3;
// S and I point at pixel (0, 0) of the padded planes; the borders are filled by k_pad afterwards
__global__ __launch_bounds__(256) void k_pyr_down(const float* __restrict__ S, int su, int sv, float* __restrict__ I,
                                                  int mu, int mv) {
    const int sp = su + 2 * PAD, mp = mu + 2 * PAD;   // pitches
    __shared__ float src[PD_SH][PD_SW + 1];
    __shared__ float hor[PD_SH][PD_TU + 1];
    const int u0 = blockIdx.x * PD_TU, v0 = blockIdx.y * PD_TV;
    const int sx0 = 2 * u0 - 2, sy0 = 2 * v0 - 2;
    const int tid = threadIdx.x;
    for (int c = tid; c < PD_SH * PD_SW; c += 256) {
        int b = c / PD_SW, a = c - b * PD_SW;
        src[b][a] = S[(long)clampi(sy0 + b, 0, sv - 1) * sp + clampi(sx0 + a, 0, su - 1)];
    }
    __syncthreads();
    const float w0 = 1.0f / 16, w1 = 4.0f / 16, w2 = 6.0f / 16;
    for (int c = tid; c < PD_SH * PD_TU; c += 256) {
        int b = c / PD_TU, a = c - b * PD_TU;
        const float* p = &src[b][2 * a];
        hor[b][a] = (((w0 * p[0] + w1 * p[1]) + w2 * p[2]) + w1 * p[3]) + w0 * p[4];
    }
    __syncthreads();
    for (int c = tid; c < PD_TV * PD_TU; c += 256) {
        int b = c / PD_TU, a = c - b * PD_TU;
        const int u = u0 + a, v = v0 + b;
        if (u < mu && v < mv) {
            const float val = (((w0 * hor[2 * b][a] + w1 * hor[2 * b + 1][a]) + w2 * hor[2 * b + 2][a]) +
                               w1 * hor[2 * b + 3][a]) + w0 * hor[2 * b + 4][a];
            I[(long)v * mp + u] = val;
        }
    }
}

// ---- replicated borders of the levels >= 1 (one launch for all of them; blockIdx.y = level - 1) ----
struct PadArgs { float* I; int levels; int nus[VO_MAX_LEVELS], nvs[VO_MAX_LEVELS]; long offs[VO_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void k_pad(PadArgs a) {
    const int l = blockIdx.y + 1;
    const int nu = a.nus[l], nv = a.nvs[l], pitch = nu + 2 * PAD;
    const long n = (long)pitch * (nv + 2 * PAD);
    float* P = a.I + a.offs[l];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int vp = (int)(i / pitch), up = (int)(i - (long)vp * pitch);
        const int u = up - PAD, v = vp - PAD;
        if (u >= 0 && u < nu && v >= 0 && v < nv) continue;   // interior: written by k_pyr_down
        P[i] = P[(long)(clampi(v, 0, nv - 1) + PAD) * pitch + clampi(u, 0, nu - 1) + PAD];
    }
}

} // namespace

int vo_launch_frame_build(vo_ctx* ctx, vo_pyr* f, const uint8_t* d_img, hipStream_t stream) {
    if (!stream) stream = ctx->stream;
    const int pitch0 = f->nu + 2 * PAD;
    hipLaunchKernelGGL(k_level0, dim3((unsigned)((pitch0 + 1023) / 1024), (unsigned)(f->nv + 2 * PAD)), dim3(256), 0, stream,
                       d_img, f->nu, f->nv, f->I);
    auto origin = [&](int l) { return f->I + f->offs[l] + (long)PAD * (f->nus[l] + 2 * PAD) + PAD; };   // pixel (0, 0)
    for (int l = 1; l < f->levels; ++l) {
        dim3 g((f->nus[l] + PD_TU - 1) / PD_TU, (f->nvs[l] + PD_TV - 1) / PD_TV);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, stream, origin(l - 1), f->nus[l - 1], f->nvs[l - 1], origin(l),
                           f->nus[l], f->nvs[l]);
    }
    if (f->levels > 1) {
        PadArgs pa;
        pa.I = f->I; pa.levels = f->levels;
        for (int l = 0; l < f->levels; ++l) { pa.nus[l] = f->nus[l]; pa.nvs[l] = f->nvs[l]; pa.offs[l] = f->offs[l]; }
        const long n1 = (long)(f->nus[1] + 2 * PAD) * (f->nvs[1] + 2 * PAD);
        hipLaunchKernelGGL(k_pad, dim3((unsigned)std::min<long>((n1 + 1023) / 1024, 256), (unsigned)(f->levels - 1)), dim3(256), 0,
                           stream, pa);
    }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
