// translation unit: mixed Z / worm-space sweep on the matrix trace path, blocks of 5..16 states
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_worm_seq(int nworm, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (nworm == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<2, 9>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<4, 9>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
}  // namespace w2dyn
