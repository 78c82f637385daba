// vo_image.hip -- image-wide stencils of the KLT front end: float pyramid (N1) + Sobel/8 gradients (N2).
// Replaces the pyramid/gradient set-up inside vision.PointTracker (modules/KLTTracker.m:37-39,
// initialize + step).  HBM-bound streaming kernels: LDS-tiled, coalesced along the fast (u) axis.
//
// Algorithmic bytes per image (KITTI 1241x376, 4 levels, P = 619,930 px): read 1*WH (u8) +
// write 12*P (I, Gu, Gv float) + re-read 4*(P - P_last) for decimation  ~= 10.4 MB.
#include "vo_common.h"

namespace {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- level 0: uint8 -> float + gradients.  tile 64 x 4, halo 1 ----
constexpr int L0_TU = 64, L0_TV = 4;
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ img, int nu, int nv,
                                                float* __restrict__ I, float* __restrict__ Gu,
                                                float* __restrict__ Gv) {
    __shared__ float t[L0_TV + 2][L0_TU + 2];
    const int u0 = blockIdx.x * L0_TU, v0 = blockIdx.y * L0_TV;
    const int tid = threadIdx.x;
    for (int c = tid; c < (L0_TV + 2) * (L0_TU + 2); c += 256) {
        int b = c / (L0_TU + 2), a = c - b * (L0_TU + 2);
        int u = clampi(u0 + a - 1, 0, nu - 1), v = clampi(v0 + b - 1, 0, nv - 1);
        t[b][a] = (float)img[(long)v * nu + u];
    }
    __syncthreads();
    const int a = (tid & 63) + 1, b = (tid >> 6) + 1;
    const int u = u0 + a - 1, v = v0 + b - 1;
    if (u < nu && v < nv) {
        float du = (t[b - 1][a + 1] - t[b - 1][a - 1]) + 2.0f * (t[b][a + 1] - t[b][a - 1]) + (t[b + 1][a + 1] - t[b + 1][a - 1]);
        float dv = (t[b + 1][a - 1] - t[b - 1][a - 1]) + 2.0f * (t[b + 1][a] - t[b - 1][a]) + (t[b + 1][a + 1] - t[b - 1][a + 1]);
        long o = (long)v * nu + u;
        I[o] = t[b][a];
        Gu[o] = du * 0.125f;
        Gv[o] = dv * 0.125f;
    }
}

// ---- level l >= 1: 5x5 binomial decimation of level l-1 + gradients of the result.
// out tile 32 x 8, computed with a 1-px halo at CLAMPED coordinates (replicate border of N2) ----
constexpr int PD_TU = 32, PD_TV = 8;
constexpr int PD_SW = 2 * (PD_TU + 2) + 3, PD_SH = 2 * (PD_TV + 2) + 3; // staged source region 71 x 23
__global__ __launch_bounds__(256) void k_pyr_down(const float* __restrict__ S, int su, int sv,
                                                  float* __restrict__ I, float* __restrict__ Gu,
                                                  float* __restrict__ Gv, int mu, int mv) {
    __shared__ float src[PD_SH][PD_SW + 1];
    __shared__ float out[PD_TV + 2][PD_TU + 2];
    const int u0 = blockIdx.x * PD_TU, v0 = blockIdx.y * PD_TV;
    const int sx0 = 2 * (u0 - 1) - 2, sy0 = 2 * (v0 - 1) - 2;
    const int tid = threadIdx.x;
    for (int c = tid; c < PD_SH * PD_SW; c += 256) {
        int b = c / PD_SW, a = c - b * PD_SW;
        src[b][a] = S[(long)clampi(sy0 + b, 0, sv - 1) * su + clampi(sx0 + a, 0, su - 1)];
    }
    __syncthreads();
    const float w[5] = {1.0f / 16, 4.0f / 16, 6.0f / 16, 4.0f / 16, 1.0f / 16};
    for (int c = tid; c < (PD_TV + 2) * (PD_TU + 2); c += 256) {
        int b = c / (PD_TU + 2), a = c - b * (PD_TU + 2);
        int ci = clampi(u0 + a - 1, 0, mu - 1), cj = clampi(v0 + b - 1, 0, mv - 1);
        // horizontal pass first, then vertical (N1); source coordinates are clamped per tap
        float tot = 0.0f;
#pragma unroll
        for (int bb = 0; bb < 5; ++bb) {
            int y = clampi(2 * cj + bb - 2, 0, sv - 1) - sy0;
            float s = 0.0f;
#pragma unroll
            for (int aa = 0; aa < 5; ++aa) {
                int x = clampi(2 * ci + aa - 2, 0, su - 1) - sx0;
                s += w[aa] * src[y][x];
            }
            tot += w[bb] * s;
        }
        out[b][a] = tot;
    }
    __syncthreads();
    const int a = (tid & 31) + 1, b = (tid >> 5) + 1;
    const int u = u0 + a - 1, v = v0 + b - 1;
    if (u < mu && v < mv) {
        float du = (out[b - 1][a + 1] - out[b - 1][a - 1]) + 2.0f * (out[b][a + 1] - out[b][a - 1]) + (out[b + 1][a + 1] - out[b + 1][a - 1]);
        float dv = (out[b + 1][a - 1] - out[b - 1][a - 1]) + 2.0f * (out[b + 1][a] - out[b - 1][a]) + (out[b + 1][a + 1] - out[b - 1][a + 1]);
        long o = (long)v * mu + u;
        I[o] = out[b][a];
        Gu[o] = du * 0.125f;
        Gv[o] = dv * 0.125f;
    }
}

} // namespace

int vo_launch_frame_build(vo_ctx* ctx, vo_pyr* f, const uint8_t* d_img) {
    dim3 g0((f->nu + L0_TU - 1) / L0_TU, (f->nv + L0_TV - 1) / L0_TV);
    hipLaunchKernelGGL(k_level0, g0, dim3(256), 0, ctx->stream, d_img, f->nu, f->nv, f->I, f->Gu, f->Gv);
    for (int l = 1; l < f->levels; ++l) {
        dim3 g((f->nus[l] + PD_TU - 1) / PD_TU, (f->nvs[l] + PD_TV - 1) / PD_TV);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, ctx->stream, f->I + f->offs[l - 1], f->nus[l - 1],
                           f->nvs[l - 1], f->I + f->offs[l], f->Gu + f->offs[l], f->Gv + f->offs[l], f->nus[l],
                           f->nvs[l]);
    }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
