// translation unit: Z-space sweep and misc estimators on the closed-form (one-state blocks) trace path, k_rebuild
#define W2DYN_TU_SCALAR
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_scalar(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count)
{
    if (calib) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, false>), l.grid, l.block, l.lds, l.stream, C, nsteps, count);
}
void launch_meas_misc_scalar(const LaunchCfg &l, const DevConst &C) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<0>), l.grid, l.block, l.lds, l.stream, C); }
void launch_rebuild(const LaunchCfg &l, const DevConst &C) { hipLaunchKernelGGL(k_rebuild, l.grid, l.block, l.lds, l.stream, C); }
}  // namespace w2dyn
