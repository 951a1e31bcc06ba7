// translation unit: mixed Z / worm-space sweep of the worm sectors with U-matrix factors beyond sector 3 (the IE = true bodies of cthyb_worm.hip.h):
// four worm operators (14, 15, 16) on both trace paths, six (5, 10), eight (12, 13) and twelve (11) on the matrix trace path (tree product for
// blocks of up to 4 states, sequential evaluator beyond).  The context holds twelve worm operators here.
#define W2DYN_NWMAX 12
#include "launch.hpp"
namespace w2dyn {
template <int NW> static void big_(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (mat <= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 4, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 9, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
void launch_sweep_worm_big(int nworm, int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (nworm == 4) {
        if (mat == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<4, 0, true>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
        else big_<4>(mat, l, C, nsteps, resume);
    }
    else if (nworm == 6) big_<6>(mat, l, C, nsteps, resume);
    else if (nworm == 8) big_<8>(mat, l, C, nsteps, resume);
    else big_<12>(mat, l, C, nsteps, resume);
}
}  // namespace w2dyn
