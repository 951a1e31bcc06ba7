// Launchers of the kernels of libw2dyn_hip.so.  The kernels are compiled in several translation units (tu_*.hip, built in
// parallel); each unit defines the launchers of the kernels it instantiates, the host unit (w2dyn_hip.hip) calls them.
#pragma once
#include "cthyb_kernels.hip.h"

namespace w2dyn {

struct LaunchCfg { dim3 grid, block; size_t lds; hipStream_t stream; };

// mat: 0 = one-state blocks (closed-form trace); 2, 3, 4 = matrix path, blocks padded to that size (tree product);
// 9 = matrix path, sequential propagation (blocks of 5..16 states)
void launch_sweep(int mat, bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_worm(int nworm, int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_meas_misc(int mat, const LaunchCfg &l, const DevConst &C);
void launch_rebuild(const LaunchCfg &l, const DevConst &C);
// chain arrays from capacity kold to knew > kold (tau arrays and M blocks keep their contents, row stride changes)
void launch_relayout(const DevConst &C, hipStream_t st, int kold, int knew, const double *tc, const double *ta, const double *m, double *tc2, double *ta2, double *m2);

// pieces (one per translation unit)
void launch_sweep_scalar(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_scalar3(const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);    // tu_scalar3.hip
void launch_sweep_tree(int mat, bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_seq(bool calib, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
// the same sweeps with the moves outside the default mix (tu_x_*.hip)
void launch_sweep_x_scalar(const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_x_tree(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_x_seq(const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
// off-diagonal hybridisation (tu_offdiag.hip, tu_x_seq.hip)
void launch_sweep_od_tree(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_sweep_od_seq(const LaunchCfg &l, const DevConst &C, long long nsteps, int count_sampling, int resume);
void launch_meas_full(const DevConst &C, hipStream_t st, double *acc_gleg_full, double *acc_gtau_full);
void launch_sweep_worm_scalar(int nworm, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_tree(int nworm, int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_seq(int nworm, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_meas_misc_scalar(const LaunchCfg &l, const DevConst &C);
void launch_meas_misc_tree(int mat, const LaunchCfg &l, const DevConst &C);
void launch_meas_misc_seq(const LaunchCfg &l, const DevConst &C);
void launch_meas_susz_mat(const LaunchCfg &l, const DevConst &C, double *acc, int nbs, double *work, const double *dens);   // tu_tree.hip

// estimators (tu_meas.hip); grids and LDS sizes are worked out here from the DevConst
// segpre: nbins * nflav + 1 element counts (work list, made by the launch itself); ncu: compute units of the device
void launch_meas_gleg(const DevConst &C, hipStream_t st, unsigned long long *segpre, int ncu);
// acc_gleg holds Chebyshev moments; tab[l][j]: Legendre polynomial l in Chebyshev polynomials (nleg x nleg)
void launch_gleg_legendre(hipStream_t st, const double *in, const double *tab, int nleg, size_t nrows, double *out);
void launch_meas_gtau_ac(const DevConst &C, hipStream_t st, double *dat, int nlev, long long n);
void launch_gtau_blocks(const DevConst &C, hipStream_t st, const double *dat, int nlev, long long n, double *out);
void launch_meas_giw(const DevConst &C, hipStream_t st);
// estimators of the equal-time worm sectors 6 - 9 (p2iw / p2tau / p3iw); nbe accumulator bins
void launch_meas_weq(const DevConst &C, hipStream_t st, int nbe, double *acc_iw, double *acc_tau, double *acc_p3, int n2b, int n3f, int n3b);
void launch_meas_susz(const DevConst &C, hipStream_t st, double *acc, int nbs);
void launch_meas_steps(const DevConst &C, hipStream_t st, double *ser, int nmeas, int m, int what);
void launch_meas_g4leg(const DevConst &C, hipStream_t st, double *acc_tau, double *acc_leg, int nbl, int nt, int nl, int nwb);
size_t g4leg_fact_lds(const DevConst &C, int nl);
void launch_meas_g4leg_fact(const DevConst &C, hipStream_t st, double *acc_leg, int nbl, int nl, int nwb);
void launch_meas_g2iw(const DevConst &C, hipStream_t st, int n2, int nwb, double2 *G2);     // G2: g2_stride(n2, nwb) entries per chain and flavour
void launch_meas_g4iw(const DevConst &C, hipStream_t st, int nwf, int nwb, int zph, int nb4, const double2 *G2, double2 *acc4, double2 *acc4pp, double2 *acc2);
void launch_meas_g4w(const DevConst &C, hipStream_t st, int pp, int nbw);
void launch_meas_gw(const DevConst &C, hipStream_t st);
// estimators of the worm sectors 5, 10 - 16: a complex array of n1 x n2 x n3 frequencies and / or a tau histogram, nbe accumulator bins
void launch_meas_wie(const DevConst &C, hipStream_t st, int nbe, double *acc_iw, double *acc_tau, int n1, int n2, int n3);
// mixed Z / worm-space sweep of the sectors 5 and 10 - 16 (those with six, eight and twelve worm operators: matrix trace path only)
void launch_sweep_worm_big4(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_big6(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_big8(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_big12(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
// mixed Z / worm-space sweep of the sectors 2, 3, 4, 6 - 9 with the two-pair and outer-block moves (matrix trace path)
void launch_sweep_worm_x2(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
void launch_sweep_worm_x4(int mat, const LaunchCfg &l, const DevConst &C, long long nsteps, int resume);
// moments of B normalised bins of `per` doubles for the cross-rank reduction: out[1 + 3 per] (w2dyn_reduce_pack)
void launch_pack_moments(hipStream_t st, const double *acc, int B, size_t per, const double *scale, int nleg, double *out);

}  // namespace w2dyn
