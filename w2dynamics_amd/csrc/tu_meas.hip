// translation unit: the estimator kernels that read (M, tau) of all chains (G(l), G(tau), G(iw), two-particle estimators,
// worm estimators, autocorrelation accumulator)
#define W2DYN_TU_MEAS
#include "launch.hpp"
#include <algorithm>
namespace w2dyn {
template <int T> static void gleg_(const DevConst &C, hipStream_t st, unsigned long long *segpre, int ncu)
{
    const int lds_hist = C.ntau <= 7000;       // tau histogram in LDS (< 64 KB with the reduction tile), else global atomics
    const size_t lds = sizeof(double) * ((lds_hist ? C.ntau : 0) + GLEG_RED_ROWS * GLEG_RED_STRIDE);
    // as many one-wave workgroups as run at once: two per SIMD by registers (T > 100: one), fewer if the histograms fill the LDS
    const int per_cu = std::max(1, std::min(T > 100 ? 4 : 8, (int)((size_t)(160 * 1024) / (lds + sizeof(GlegGroup) + 64))));
    const int nseg = C.nbins * C.nflav;
    // segpre[0 .. nseg] are the prefix sums; behind them: per-workgroup totals and the ticket counter of k_gleg_segments
    unsigned long long *blocktot = segpre + nseg + 1;
    const int bpb = GLEG_SEG_BLOCK / C.nflav, nblk = (C.nbins + bpb - 1) / bpb;
    hipLaunchKernelGGL(k_gleg_segments, dim3(nblk), dim3(1024), 0, st, C, segpre, blocktot, (unsigned *)(blocktot + nblk));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_gleg<T>), dim3(ncu * per_cu), dim3(64), lds, st, C, lds_hist, segpre);
}
void launch_meas_gleg(const DevConst &C, hipStream_t st, unsigned long long *segpre, int ncu)
{
    if (C.nleg <= 16) gleg_<16>(C, st, segpre, ncu);
    else if (C.nleg <= 32) gleg_<32>(C, st, segpre, ncu);
    else if (C.nleg <= 64) gleg_<64>(C, st, segpre, ncu);
    else if (C.nleg <= 100) gleg_<100>(C, st, segpre, ncu);
    else gleg_<128>(C, st, segpre, ncu);
}
void launch_gleg_legendre(hipStream_t st, const double *in, const double *tab, int nleg, size_t nrows, double *out)
{
    hipLaunchKernelGGL(k_gleg_legendre, dim3((unsigned)((nrows * nleg + 255) / 256)), dim3(256), 0, st, in, tab, nleg, nrows, out);
}
void launch_meas_gtau_ac(const DevConst &C, hipStream_t st, double *dat, int nlev, long long n)
{
    hipLaunchKernelGGL(k_meas_gtau_ac, dim3(C.nwalkers, C.nflav), dim3(256), (size_t)C.ntau * sizeof(double), st, C, dat, nlev, n);
}
void launch_gtau_blocks(const DevConst &C, hipStream_t st, const double *dat, int nlev, long long n, double *out)
{
    hipLaunchKernelGGL(k_gtau_blocks, dim3(C.nbins, C.nflav * nlev), dim3(256), 0, st, C, dat, nlev, n, out);
}
void launch_meas_giw(const DevConst &C, hipStream_t st)
{
    hipLaunchKernelGGL(k_meas_giw, dim3(C.nbins, C.nflav, (C.niw2 + 63) / 64), dim3(64), 0, st, C);
}
void launch_meas_g4leg(const DevConst &C, hipStream_t st, double *acc_tau, double *acc_leg, int nbl, int nt, int nl, int nwb)
{
    hipLaunchKernelGGL(k_meas_g4leg, dim3(C.nwalkers, C.nflav * C.nflav), dim3(256), 0, st, C, acc_tau, acc_leg, nbl, nt, nl, nwb);
}
size_t g4leg_fact_lds(const DevConst &C, int nl) { return ((size_t)(3 * C.kmax + 3) * nl + C.kmax + 1 + 2 * 24) * sizeof(double); }
void launch_meas_g4leg_fact(const DevConst &C, hipStream_t st, double *acc_leg, int nbl, int nl, int nwb)
{
    // one (l, l') pair per thread: two pairs per thread (a single tile up to N4leg = 22) measured slower (214 ms
    // against 162 ms per pass at N4leg = 20, N4iwb = 10: the second set of accumulators costs occupancy)
    const dim3 grid(C.nwalkers, C.nflav * C.nflav);
    const size_t lds = g4leg_fact_lds(C, nl);
    if (nwb < 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_g4leg_fact<4, 1>), grid, dim3(256), lds, st, C, acc_leg, nbl, nl, nwb);
    else if (nwb < 12) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_g4leg_fact<12, 1>), grid, dim3(256), lds, st, C, acc_leg, nbl, nl, nwb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_g4leg_fact<24, 1>), grid, dim3(256), lds, st, C, acc_leg, nbl, nl, nwb);
}
void launch_meas_g2iw(const DevConst &C, hipStream_t st, int n2, double2 *G2)
{
    hipLaunchKernelGGL(k_meas_g2iw, dim3(C.nwalkers, C.nflav), dim3(256), (size_t)3 * C.kmax * G2_FC * sizeof(double2), st, C, n2, G2);
}
void launch_meas_g4iw(const DevConst &C, hipStream_t st, int nwf, int nwb, int zph, int nb4, const double2 *G2, double2 *acc4, double2 *acc4pp, double2 *acc2)
{
    const int n2 = 2 * (nwf + nwb);
    const size_t n4 = (size_t)C.nflav * C.nflav * 4 * nwf * nwf * (2 * nwb + 1), ng2 = (size_t)C.nflav * n2 * n2;
    hipLaunchKernelGGL(k_meas_g4iw, dim3((unsigned)(((n4 > ng2 ? n4 : ng2) + 255) / 256), nb4), dim3(256), 0, st, C, nwf, nwb, zph, nb4, G2, acc4, acc4pp, acc2);
}
void launch_meas_g4w(const DevConst &C, hipStream_t st, int pp, int nbw)
{
    hipLaunchKernelGGL(k_meas_g4w, dim3(nbw), dim3(256), (size_t)(4 * C.nwf + 2 * C.nwb + 1) * sizeof(double2), st, C, pp, nbw);
}
void launch_meas_gw(const DevConst &C, hipStream_t st)
{
    hipLaunchKernelGGL(k_meas_gw, dim3(C.nbins), dim3(256), (size_t)C.ntau * sizeof(double), st, C);
}

// Moments of the normalised per-bin results x_b[i] = acc[b][i] scale[b] (sqrt(2l+1), l = i mod nleg, for the Legendre
// coefficients) for the cross-rank reduction: out = [ B | S_i = sum_b x_b[i] | sum_b (x_b[i] - S_i/B)^2 | S_i^2/B ].  Every
// entry adds over ranks (parallel-axis formula, statistics.py of this package).  One thread per element, bins read with
// unit stride across the wave; the second pass over the bins finds them in L2.
__global__ void __launch_bounds__(256) k_pack_moments(const double *__restrict__ acc, int B, size_t per, const double *__restrict__ scale,
                                                      int nleg, double *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) out[0] = (double)B;
    if (i >= per) return;
    const double ef = nleg ? sqrt(2.0 * (double)(i % (size_t)nleg) + 1.0) : 1.0;
    double s1 = 0.0;
    for (int b = 0; b < B; ++b) s1 += acc[(size_t)b * per + i] * (scale[b] * ef);
    const double m = s1 / (double)B;
    double m2 = 0.0;
    for (int b = 0; b < B; ++b) { const double d = acc[(size_t)b * per + i] * (scale[b] * ef) - m; m2 += d * d; }
    out[1 + i] = s1; out[1 + per + i] = m2; out[1 + 2 * per + i] = s1 * m;
}
void launch_pack_moments(hipStream_t st, const double *acc, int B, size_t per, const double *scale, int nleg, double *out)
{
    hipLaunchKernelGGL(k_pack_moments, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, acc, B, per, scale, nleg, out);
}
}  // namespace w2dyn
