// translation unit: mixed Z / worm-space sweep with 2 worm operators (sectors 2) together with the moves outside the default mix - two pairs at
// once, pair moves that change the outer block (Percentage4OperatorMove, PercentageOuterMove; CTQMC.F90:8441-8449, 8497-8516) - on the matrix trace path, blocks of up to four states (tree product)
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_worm_x2(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<2, 2, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<2, 3, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<2, 4, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);      // blocks of more than four states: refused by enter_sector
}
}  // namespace w2dyn
