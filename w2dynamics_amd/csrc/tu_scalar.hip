// translation unit: Z-space sweep and misc estimators on the closed-form (one-state blocks) trace path, k_rebuild
#define W2DYN_TU_SCALAR
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_scalar(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (calib) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else if (C.sweep_wpe3) launch_sweep_scalar3(l, C, nsteps, count, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, false>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
void launch_meas_misc_scalar(const LaunchCfg &l, const DevConst &C) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<0>), l.grid, l.block, l.lds, l.stream, C); }
void launch_rebuild(const LaunchCfg &l, const DevConst &C) { hipLaunchKernelGGL(k_rebuild, l.grid, l.block, l.lds, l.stream, C); }

// one workgroup per (chain, flavour): the k used slots of the tau arrays and the k x k block of M move to the new strides
__global__ __launch_bounds__(256) void k_relayout(DevConst C, int kold, int knew, const double *tc, const double *ta, const double *m, double *tc2, double *ta2, double *m2)
{
    const size_t wf = (size_t)blockIdx.x * C.nflav + blockIdx.y;
    const int k = min(C.K[wf], kold);
    for (int e = threadIdx.x; e < k; e += 256) { tc2[wf * knew + e] = tc[wf * kold + e]; ta2[wf * knew + e] = ta[wf * kold + e]; }
    for (int e = threadIdx.x; e < k * k; e += 256) { const int i = e / k, j = e - i * k; m2[wf * knew * knew + (size_t)i * knew + j] = m[wf * kold * kold + (size_t)i * kold + j]; }
}
void launch_relayout(const DevConst &C, hipStream_t st, int kold, int knew, const double *tc, const double *ta, const double *m, double *tc2, double *ta2, double *m2)
{
    hipLaunchKernelGGL(k_relayout, dim3(C.nwalkers, C.nflav), dim3(256), 0, st, C, kold, knew, tc, ta, m, tc2, ta2, m2);
}
}  // namespace w2dyn
