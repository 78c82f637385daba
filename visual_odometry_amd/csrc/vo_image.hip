// vo_image.hip -- image-wide stencil of the KLT front end: the float pyramid (oracle N1).
// Replaces the pyramid set-up inside vision.PointTracker (modules/KLTTracker.m:37-39, initialize +
// step).  Gradients are NOT stored: the tracker takes the Sobel operator of the sub-pixel-sampled
// template (vo_klt.hip), so a frame is one float plane per level.
// HBM-bound streaming kernels: LDS-tiled, coalesced along the fast (u) axis.
//
// Algorithmic bytes per image (KITTI 1241x376, 4 levels, P = 619,930 px): read 1*WH (u8) + write
// 4*P (float I) + re-read 4*(P - P_last) for decimation ~= 5.4 MB.
#include "vo_common.h"

namespace {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- level 0: uint8 -> float, 4 pixels per thread ----
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ img, long n, float* __restrict__ I) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n && (((uintptr_t)img) & 3) == 0) {
        const uchar4 p = *reinterpret_cast<const uchar4*>(img + i4);
        *reinterpret_cast<float4*>(I + i4) = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
    } else {
        for (long i = i4; i < n && i < i4 + 4; ++i) I[i] = (float)img[i];
    }
}

// ---- level l >= 1: 5x5 binomial decimation of level l-1 (horizontal pass first, then vertical, N1).
// out tile 64 x 8; staged source region (2*64+3) x (2*8+3) at replicate-clamped coordinates ----
constexpr int PD_TU = 64, PD_TV = 8;
constexpr int PD_SW = 2 * PD_TU + 3, PD_SH = 2 * PD_TV + 3;
__global__ __launch_bounds__(256) void k_pyr_down(const float* __restrict__ S, int su, int sv, float* __restrict__ I,
                                                  int mu, int mv) {
    __shared__ float src[PD_SH][PD_SW + 1];
    __shared__ float hor[PD_SH][PD_TU + 1];
    const int u0 = blockIdx.x * PD_TU, v0 = blockIdx.y * PD_TV;
    const int sx0 = 2 * u0 - 2, sy0 = 2 * v0 - 2;
    const int tid = threadIdx.x;
    for (int c = tid; c < PD_SH * PD_SW; c += 256) {
        int b = c / PD_SW, a = c - b * PD_SW;
        src[b][a] = S[(long)clampi(sy0 + b, 0, sv - 1) * su + clampi(sx0 + a, 0, su - 1)];
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
        if (u < mu && v < mv)
            I[(long)v * mu + u] = (((w0 * hor[2 * b][a] + w1 * hor[2 * b + 1][a]) + w2 * hor[2 * b + 2][a]) +
                                   w1 * hor[2 * b + 3][a]) + w0 * hor[2 * b + 4][a];
    }
}

} // namespace

int vo_launch_frame_build(vo_ctx* ctx, vo_pyr* f, const uint8_t* d_img, hipStream_t stream) {
    if (!stream) stream = ctx->stream;
    const long n = (long)f->nu * f->nv;
    hipLaunchKernelGGL(k_level0, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, d_img, n, f->I);
    for (int l = 1; l < f->levels; ++l) {
        dim3 g((f->nus[l] + PD_TU - 1) / PD_TU, (f->nvs[l] + PD_TV - 1) / PD_TV);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, stream, f->I + f->offs[l - 1], f->nus[l - 1],
                           f->nvs[l - 1], f->I + f->offs[l], f->nus[l], f->nvs[l]);
    }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
