// translation unit: mixed Z / worm-space sweep on the matrix trace path, blocks of 2..4 states
#include "launch.hpp"
namespace w2dyn {
template <int NW> static void worm_(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 2>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 3>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 4>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
void launch_sweep_worm_tree(int nworm, int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (nworm == 2) worm_<2>(mat, l, C, nsteps, resume); else worm_<4>(mat, l, C, nsteps, resume);
}
}  // namespace w2dyn
