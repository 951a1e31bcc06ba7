// translation unit: Z-space sweep with the moves outside the default mix, and with off-diagonal hybridisation, on the matrix
// trace path for blocks of 5..16 states
#define W2DYN_WPE_MAT 1     // as tu_seq.hip
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_x_seq(const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep<9, false, false, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
void launch_sweep_od_seq(const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_od<9>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
}  // namespace w2dyn
