// translation unit: Z-space sweep and misc estimators on the matrix trace path, blocks of 2..4 states (tree product)
#include "launch.hpp"
namespace w2dyn {
template <int MAT> static void sweep_(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (calib) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<MAT, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<MAT, false>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
void launch_sweep_tree(int mat, bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (mat == 2) sweep_<2>(calib, l, C, nsteps, count, resume); else if (mat == 3) sweep_<3>(calib, l, C, nsteps, count, resume); else sweep_<4>(calib, l, C, nsteps, count, resume);
}
void launch_meas_misc_tree(int mat, const LaunchCfg &l, const DevConst &C)
{
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<2>), l.grid, l.block, l.lds, l.stream, C);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<3>), l.grid, l.block, l.lds, l.stream, C);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_misc<4>), l.grid, l.block, l.lds, l.stream, C);
}
// MeasSusz on the matrix trace path (blocks of up to 8 states); work: 4 (wcap + 1) (nsm^2 + 1) doubles per chain
void launch_meas_susz_mat(const LaunchCfg &l, const DevConst &C, double *acc, int nbs, double *work, const double *dens)
{
    if (C.nsm <= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz_mat<4>), l.grid, l.block, l.lds, l.stream, C, acc, nbs, work, dens);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz_mat<8>), l.grid, l.block, l.lds, l.stream, C, acc, nbs, work, dens);
}
}  // namespace w2dyn
