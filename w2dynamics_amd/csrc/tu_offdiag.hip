// translation unit: off-diagonal hybridisation (offdiag = 1) - Z-space sweep for blocks of up to 4 states, the full-matrix estimator
#define W2DYN_TU_OFFDIAG
#include "launch.hpp"
namespace w2dyn {
void launch_sweep_od_tree(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int count, int resume)
{
    if (mat == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_od<2>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else if (mat == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_od<3>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_od<4>), l.grid, l.block, l.lds, l.stream, C, nsteps, count, resume);
}
void launch_meas_full(const DevConst &C, hipStream_t st, double *acc_gleg_full, double *acc_gtau_full)
{
    // the Legendre sums of a workgroup's four chains are formed in LDS when they share an accumulator bin and fit
    const size_t bytes = (size_t)C.nflav * C.nflav * C.nleg * sizeof(double);
    const int lds_leg = (C.walkers_per_bin % 4 == 0 && bytes <= 64 * 1024) ? 1 : 0;
    hipLaunchKernelGGL(k_meas_full, dim3((C.nwalkers + 3) / 4), dim3(256), lds_leg ? bytes : 0, st, C, acc_gleg_full, acc_gtau_full, lds_leg);
}
}  // namespace w2dyn
