// Micro-benchmark: fp64 vector FMA issue rate on gfx950 as a function of waves per SIMD and independent chains per lane.
// Build: hipcc -O3 --offload-arch=gfx950 fp64_rate.hip -o fp64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k(double *out, int iters, double a, double b)
{
    double v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
template <int CH> void run(int waves_per_simd)
{
    double *out; hipMalloc(&out, 8);
    int iters = 20000;
    int blocks = 256 * waves_per_simd;            // 256-thread blocks: 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * CH * (double)iters * blocks * 256;
    printf("chains/lane %2d waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", CH, waves_per_simd, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
