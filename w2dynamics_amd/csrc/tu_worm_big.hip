// translation unit: mixed Z / worm-space sweep of the worm sectors with six (5, 10), eight (12, 13) and twelve (11) worm operators, on the
// matrix trace path (tree product for blocks of up to 4 states, sequential evaluator beyond); the context holds twelve worm operators here
#define W2DYN_NWMAX 12
#include "launch.hpp"
namespace w2dyn {
template <int NW> static void big_(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (mat <= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 4>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sweep_worm<NW, 9>), l.grid, l.block, l.lds, l.stream, C, nsteps, resume);
}
void launch_sweep_worm_big(int nworm, int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume)
{
    if (nworm == 6) big_<6>(mat, l, C, nsteps, resume);
    else if (nworm == 8) big_<8>(mat, l, C, nsteps, resume);
    else big_<12>(mat, l, C, nsteps, resume);
}
}  // namespace w2dyn
