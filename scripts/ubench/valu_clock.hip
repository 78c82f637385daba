// Microbenchmark: VALU issue rate in CYCLES and the clock the chip holds while it runs (gfx950).
// Every wave stamps s_memtime (shader clock ticks) and s_memrealtime (100 MHz) around its loop:
//   cycles per instruction per SIMD = waves_per_SIMD * d(memtime) / instructions ... reported per wave and per SIMD
//   in-kernel clock                 = d(memtime) / d(memrealtime) * 100 MHz
// Settles the question behind DESIGN.md section 4: is the fp32 vector issue rate 2 cycles per wave64
// instruction per SIMD (MI355X_MICROARCH.md) and what does a VALU-dense kernel clock at?
//   hipcc --offload-arch=gfx950 -O3 valu_clock.hip -o valu_clock && ./valu_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* stamps, float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7, da = a, db = b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) { // one dependent chain
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(x0) : "v"(a), "v"(b));
        } else if (MODE == 2) { // 8 independent v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));
        } else if (MODE == 3) { // sub + fma pairs (the lerp of the LK set-up): 4 pairs
            asm volatile("v_sub_f32 %0, %1, %0\n v_fma_f32 %0, %0, %8, %1\n v_sub_f32 %2, %3, %2\n v_fma_f32 %2, %2, %8, %3\n"
                         "v_sub_f32 %4, %5, %4\n v_fma_f32 %4, %4, %8, %5\n v_sub_f32 %6, %7, %6\n v_fma_f32 %6, %6, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 4) { // 8 independent v_pk_fma_f32 (two fp32 FMAs per lane and instruction) on the fp64 registers' pairs
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    x0 += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char* name, unsigned long long* d_st, float* d_out, int wps, int iters) {
    const int blocks = 256 * 4 * wps; // waves per SIMD = wps when the dispatcher spreads them evenly
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // warm the clocks: ~0.3 s of the same work first
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_st, d_out, iters, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_st, d_out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * (size_t)blocks);
    (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(blocks), clk(blocks);
    for (int i = 0; i < blocks; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
    std::nth_element(cyc.begin(), cyc.begin() + blocks / 2, cyc.end());
    std::nth_element(clk.begin(), clk.begin() + blocks / 2, clk.end());
    const double inst = (double)iters * 8;
    const double per_wave = cyc[blocks / 2] / inst;             // cycles between two instructions of ONE wave
    const double per_simd = per_wave / wps;                      // cycles per instruction issued by the SIMD
    const double ns_simd = ms * 1e6 / (wps * inst);
    printf("%-18s waves/SIMD=%d  %.3f ms  in-kernel clock %.3f GHz  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD  %.3f ns/instr/SIMD\n",
           name, wps, ms, clk[blocks / 2], per_wave, per_simd, ns_simd);
}

int main() {
    unsigned long long* d_st; float* d_out;
    (void)hipMalloc(&d_st, 256 * 4 * 8 * 2 * sizeof(unsigned long long));
    (void)hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 400000;
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 indep", d_st, d_out, w, iters);
        run<1>("v_fma_f32 chain", d_st, d_out, w, iters);
        run<3>("v_sub+v_fma pairs", d_st, d_out, w, iters);
        run<2>("v_fma_f64 indep", d_st, d_out, w, iters / 2);
        run<4>("v_pk_fma_f32 indep", d_st, d_out, w, iters / 2);
    }
    return 0;
}
