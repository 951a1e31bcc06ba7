// translation unit: the Z-space sweep kernel of the closed-form trace path held to three waves per SIMD
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_scalar3(const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<0, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
}  // namespace w2dyn
