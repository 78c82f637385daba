// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 at 1..8 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    float2v av = {a, a}, bv = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7, da = a, db = b;
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 8 independent scalar FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) { // 8 independent packed FMAs
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(av), "v"(bv));
        } else if (MODE == 2) { // one dependent chain of scalar FMAs
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(x0) : "v"(a), "v"(b));
        } else if (MODE == 3) { // packed add
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(av));
        } else if (MODE == 4) { // dependent chain packed fma
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(p0) : "v"(av), "v"(bv));
        } else if (MODE == 7) { // 8 independent double-precision FMAs
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));
        } else if (MODE == 5) { // 8 independent scalar-ALU adds
            asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                         "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(iters) : "scc");
        } else if (MODE == 6) { // 4 vector FMAs interleaved with 4 scalar adds
            asm volatile("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %4, %4, %8\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %5, %5, %8\n"
                         "v_fma_f32 %2, %2, %9, %10\n s_add_u32 %6, %6, %8\n v_fma_f32 %3, %3, %9, %10\n s_add_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(iters), "v"(a), "v"(b) : "scc");
        }
    }
    x0 += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    x0 += (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, float* d, int wps) {
    const int iters = 200000;
    const int blocks = 256 * 4 * wps; // waves per SIMD = wps if the dispatcher spreads evenly
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1000, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)blocks * iters * 8;
    double per_simd_ns = ms * 1e6 / ((double)wps * iters * 8); // ns per instruction per SIMD
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns/instr/SIMD (%.2f cyc @2.4GHz)  %.1f Ginstr/s\n", name, wps, ms, per_simd_ns,
           per_simd_ns * 2.4, inst / ms / 1e6);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<0>("v_fma_f32 indep", d, w);
        run<1>("v_pk_fma_f32 indep", d, w);
        run<2>("v_fma_f32 chain", d, w);
        run<4>("v_pk_fma_f32 chain", d, w);
        run<3>("v_pk_add_f32 indep", d, w);
        run<7>("v_fma_f64 indep", d, w);
        run<5>("s_add_u32 indep", d, w);
        run<6>("4 v_fma + 4 s_add", d, w);
    }
    return 0;
}
