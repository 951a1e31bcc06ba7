// translation unit: mixed Z / worm-space sweep of the worm sectors with U-matrix factors beyond sector 3 (the IE = true bodies of cthyb_worm.hip.h)
// with 8 worm operators (sectors 12, 13; both trace paths).  The context holds twelve worm operators here.
#define W2DYN_NWMAX 12
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_worm_big8(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (mat == 0) { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<8, 0, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume); return; }
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<8, 2, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<8, 3, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else if (mat == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<8, 4, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<8, 9, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
}  // namespace w2dyn
