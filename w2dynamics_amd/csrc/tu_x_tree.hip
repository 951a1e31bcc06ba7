// translation unit: Z-space sweep with the moves outside the default mix on the matrix trace path, blocks of 2..4 states
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_x_tree(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<2, false, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<3, false, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<4, false, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
}  // namespace w2dyn
