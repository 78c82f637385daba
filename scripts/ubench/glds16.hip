// Does global_load_lds_dwordx4 accept source addresses that are only 4-byte aligned?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(1))) const float glb_float;

__global__ __launch_bounds__(64) void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    const int lane = threadIdx.x;
    lds_float* lp = reinterpret_cast<lds_float*>(reinterpret_cast<uintptr_t>(buf));
    const float* g = src + shift + lane * 7; // 4-byte aligned only
    __builtin_amdgcn_global_load_lds((glb_float*)g, lp, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = buf[lane * 4 + i];
}

int main() {
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *o;
    if (hipMalloc(&d, N * 4) != hipSuccess || hipMalloc(&o, 256 * 4) != hipSuccess) return 2;
    (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        if (hipDeviceSynchronize() != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
        std::vector<float> r(256);
        (void)hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int b = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) b += r[l * 4 + i] != (float)(shift + l * 7 + i);
        printf("shift %d: %d mismatches (lane 1 got %g %g %g %g)\n", shift, b, r[4], r[5], r[6], r[7]);
        bad += b;
    }
    return bad ? 1 : 0;
}
