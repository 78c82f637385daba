// Microbenchmark: what does v_mfma_f64_16x16x4_f64 compute, bit for bit, and how fast?  (gfx950)
//   D(16x16) = A(16x4) B(4x16) + C.  Is an element the chain fma(a3 b3, fma(a2 b2, fma(a1 b1, fma(a0 b0, c)))) --
//   one correctly rounded fused multiply-add per k, k ascending -- or something else (k descending, a tree, wider
//   accumulation)?  The pose step's projection (3 rows x [X Y Z 1] per pose x point, oracle N9) is such a chain; if the
//   matrix unit reproduces it, those nine of a test's sixteen fp64 operations can leave the vector unit.
//   Also checks the operand layout assumed below and times a stream of independent MFMAs per wave.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_chain.hip -o mfma_f64_chain && ./mfma_f64_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// layout under test: A: lane l holds A[l % 16][l / 16];  B: lane l holds B[l / 16][l % 16];
// C / D: lane l, register v holds column l % 16 and row 4 v + l / 16 (LAYOUT 0) or 4 (l / 16) + v (LAYOUT 1)
template <int LAYOUT>
__global__ __launch_bounds__(64) void k_once(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    double4_t c;
    for (int v = 0; v < 4; ++v) c[v] = C[(LAYOUT ? 4 * (l / 16) + v : 4 * v + l / 16) * 16 + l % 16];
    const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(LAYOUT ? 4 * (l / 16) + v : 4 * v + l / 16) * 16 + l % 16] = d[v];
}

__global__ __launch_bounds__(64) void k_rate(unsigned long long* stamps, double* out, int iters) {
    const int l = threadIdx.x;
    const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {   // four independent accumulators: no back-to-back dependency
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}

// does the vector unit run beside the matrix unit?  MODE 0: one MFMA a trip; 1: NF independent v_fma_f32 a trip; 2: both
template <int MODE, int NF>
__global__ __launch_bounds__(64) void k_mix(double* out, int iters) {
    const int l = threadIdx.x;
    const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double4_t c0 = {0, 0, 0, 0};
    float x[8];
    for (int k = 0; k < 8; ++k) x[k] = l + k;
    const float fa = 1.0001f, fb = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE != 1) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < NF / 8; ++r)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(fa), "v"(fb));
        }
    }
    double t = c0[0] + c0[1] + c0[2] + c0[3];
    for (int k = 0; k < 8; ++k) t += x[k];
    out[blockIdx.x * 64 + l] = t;
}
template <int MODE, int NF>
static float time_mix(int blocks, double* dO, int iters) {
    hipLaunchKernelGGL((k_mix<MODE, NF>), dim3(blocks), dim3(64), 0, 0, dO, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mix<MODE, NF>), dim3(blocks), dim3(64), 0, 0, dO, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static bool same(double x, double y) { return std::memcmp(&x, &y, 8) == 0; }

int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> mant(-1.0, 1.0);
    std::uniform_int_distribution<int> ex(-8, 8);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
  for (int layout = 0; layout < 2; ++layout) {
    long n = 0, fwd = 0, rev = 0, tree = 0, unfused = 0;
    for (int trial = 0; trial < 200; ++trial) {
        double A[64], B[64], C[256], D[256];
        for (double& x : A) x = std::ldexp(mant(rng), ex(rng));
        for (double& x : B) x = std::ldexp(mant(rng), ex(rng));
        for (double& x : C) x = std::ldexp(mant(rng), ex(rng));
        hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
        if (layout) hipLaunchKernelGGL(k_once<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        else hipLaunchKernelGGL(k_once<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double* a = A + 4 * i;
                double f = C[16 * i + j], r = C[16 * i + j], u = C[16 * i + j];
                for (int k = 0; k < 4; ++k) f = std::fma(a[k], B[16 * k + j], f);
                for (int k = 3; k >= 0; --k) r = std::fma(a[k], B[16 * k + j], r);
                for (int k = 0; k < 4; ++k) { volatile double p = a[k] * B[16 * k + j]; u = u + p; }
                const double t = std::fma(a[1], B[16 + j], a[0] * B[j]) + std::fma(a[3], B[48 + j], a[2] * B[32 + j]) + C[16 * i + j];
                ++n; fwd += same(D[16 * i + j], f); rev += same(D[16 * i + j], r); tree += same(D[16 * i + j], t);
                unfused += same(D[16 * i + j], u);
            }
    }
    printf("C/D rows %s: elements %ld: equal to the ascending-k fma chain %ld, descending-k chain %ld, pairwise tree %ld, unfused ascending %ld\n",
           layout ? "4 (l / 16) + v" : "4 v + l / 16", n, fwd, rev, tree, unfused);
  }
    // rate: W waves per SIMD on every SIMD of the chip
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* dS; double* dO;
    const int iters = 20000;
    for (int w : {1, 2, 4}) {
        const int blocks = cus * 4 * w;
        hipMalloc(&dS, blocks * 16); hipMalloc(&dO, blocks * 64 * 8);
        hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(64), 0, 0, dS, dO, 200);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(64), 0, 0, dS, dO, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> S(2 * blocks);
        hipMemcpy(S.data(), dS, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> per;
        for (int b = 0; b < blocks; ++b) per.push_back(double(S[2 * b + 1] - S[2 * b]) / (4.0 * iters));
        std::sort(per.begin(), per.end());
        const double mfmas = 4.0 * iters * blocks, flop = mfmas * 2.0 * 16 * 16 * 4;
        printf("%d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s fp64, %.1f ns per MFMA per SIMD, median s_memtime ticks per MFMA per wave %.1f\n",
               w, ms, flop / (ms * 1e-3) / 1e12, ms * 1e6 / (mfmas / (cus * 4)), per[per.size() / 2]);
        hipFree(dS); hipFree(dO);
    }
    {   // 4 waves per SIMD, 10000 trips
        const int blocks = cus * 16, it = 10000;
        hipMalloc(&dO, blocks * 64 * 8);
        const float m = time_mix<0, 8>(blocks, dO, it);
        printf("4 waves per SIMD, per trip and SIMD: one MFMA alone %.1f ns\n", m * 1e6 / (4.0 * it));
        const float f16 = time_mix<1, 16>(blocks, dO, it), b16 = time_mix<2, 16>(blocks, dO, it);
        const float f32 = time_mix<1, 32>(blocks, dO, it), b32 = time_mix<2, 32>(blocks, dO, it);
        const float f64 = time_mix<1, 64>(blocks, dO, it), b64 = time_mix<2, 64>(blocks, dO, it);
        printf("  16 v_fma_f32 alone %.1f ns, with the MFMA %.1f ns\n", f16 * 1e6 / (4.0 * it), b16 * 1e6 / (4.0 * it));
        printf("  32 v_fma_f32 alone %.1f ns, with the MFMA %.1f ns\n", f32 * 1e6 / (4.0 * it), b32 * 1e6 / (4.0 * it));
        printf("  64 v_fma_f32 alone %.1f ns, with the MFMA %.1f ns\n", f64 * 1e6 / (4.0 * it), b64 * 1e6 / (4.0 * it));
        hipFree(dO);
    }
    return 0;
}
