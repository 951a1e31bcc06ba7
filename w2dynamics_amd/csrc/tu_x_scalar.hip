// translation unit: Z-space sweep with the moves outside the default mix (two pairs at once, pair moves that change the
// outer block) on the closed-form (one-state blocks) trace path
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_x_scalar(const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, false, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
}  // namespace w2dyn
