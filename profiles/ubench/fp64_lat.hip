// Micro-benchmark: fp64 FMA dependent-issue latency and the k_meas_gleg inner loop (Chebyshev recurrence + moment FMA)
// on gfx950 for E elements per lane and W waves per SIMD.  Inner bodies are unrolled 48 x so that loop overhead is noise.
// Build: hipcc -O3 --offload-arch=gfx950 fp64_lat.hip -o fp64_lat ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void __launch_bounds__(256) k_chain(double *out, int iters, double a, double b)
{
    double v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
// E elements per lane, NL orders: per order and element T' = fma(x2, T, -Tm1); acc[l] = fma(m, T', acc[l])
template <int E, int NL>
__global__ void __launch_bounds__(64) k_cheb(double *out, int passes, double seed)
{
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = 0.0;
    double x[E], m[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { x[e] = seed * (threadIdx.x + 1) * (e + 1) * 1e-3; m[e] = 1.0 + e; }
    for (int p = 0; p < passes; ++p) {
        double tm[E], t[E], x2[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { x[e] = fma(x[e], 0.999, 1e-4); x2[e] = x[e] + x[e]; tm[e] = 1.0; t[e] = x[e]; }
#pragma unroll
        for (int l = 2; l < NL; ++l) {
#pragma unroll
            for (int e = 0; e < E; ++e) { const double n = fma(x2[e], t[e], -tm[e]); tm[e] = t[e]; t[e] = n; }
#pragma unroll
            for (int e = 0; e < E; ++e) acc[l] = fma(m[e], t[e], acc[l]);
        }
    }
    double s = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) s += acc[l];
    if (s == 12345.678) out[0] = s;
}
static float timeit(void (*launch)(int), int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(2); hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
static double *g_out; static int g_w;
template <int CH> void run_chain()
{
    const int iters = 2000;
    float ms = timeit([](int it) { hipLaunchKernelGGL(k_chain<CH>, dim3(256 * g_w), dim3(256), 0, 0, g_out, it, 1.0000001, 1e-9); }, iters);
    // per SIMD: g_w waves, each CH*48*iters FMAs
    double cyc = ms * 1e-3 * 2.4e9 / (48.0 * iters);
    printf("chain: waves/SIMD %d chains %d : %.1f cycles(2.4GHz) per unrolled step of %d FMAs per wave -> %.1f cycles per FMA per SIMD, %.1f TFLOP/s\n", g_w, CH, cyc, CH, cyc / (CH * g_w),
           2.0 * CH * 48.0 * iters * 256 * g_w * 256 / ms / 1e9);
}
template <int E, int NL> void run_cheb()
{
    const int passes = 400;
    float ms = timeit([](int it) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb<E, NL>), dim3(1024 * g_w), dim3(64), 0, 0, g_out, it, 1.0); }, passes);
    double instr = 2.0 * E * (NL - 2) * passes;       // per wave
    printf("cheb: waves/SIMD %d E %d NL %d : %.3f ms, %.2f cycles(2.4GHz) per VALU instr per SIMD, algorithmic %.1f TFLOP/s (5 flop per element and order)\n", g_w, E, NL, ms,
           ms * 1e-3 * 2.4e9 / (instr * g_w), 5.0 * E * 64 * NL * passes * 1024.0 * g_w / ms / 1e9);
}
int main()
{
    hipMalloc(&g_out, 8);
    for (int w : {1, 2, 4}) { g_w = w; run_chain<1>(); run_chain<2>(); run_chain<4>(); run_chain<8>(); }
    for (int w : {1, 2}) { g_w = w; run_cheb<1, 100>(); run_cheb<2, 100>(); run_cheb<3, 100>(); run_cheb<4, 100>(); }
    for (int w : {2, 3, 4}) { g_w = w; run_cheb<2, 50>(); run_cheb<4, 50>(); }
    return 0;
}
