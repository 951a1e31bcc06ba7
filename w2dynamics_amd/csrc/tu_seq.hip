// translation unit: Z-space sweep and misc estimators on the matrix trace path, blocks of 5..16 states (sequential propagation)
#define W2DYN_WPE_MAT 1     // 407 registers when left alone: at 256 the kernel spilled 650 and still ran one wave per SIMD
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_seq(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (calib) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<9, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<9, false>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
void launch_meas_misc_seq(const LaunchCfg &l, const DevConst &C) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<9>), l.grid, l.block, l.lds, l.stream, C); }
}  // namespace w2dyn
