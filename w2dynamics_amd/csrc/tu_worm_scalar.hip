// translation unit: mixed Z / worm-space sweep (sectors 2 and 4) on the closed-form trace path
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_worm_scalar(int nworm, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (nworm == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<2, 0>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<4, 0>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
}  // namespace w2dyn
