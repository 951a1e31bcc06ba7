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
    // creators in chunks of ccap operators (phase tables in LDS: 640 B per operator), one chunk in practice
    const int ccap = std::min(C.kmax, 104), niw = C.niw2 / 2, per = 256 * GIW_NP;
    const size_t lds = (size_t)(GIW_TS + ccap) * GIW_TAB * sizeof(double2);
    hipLaunchKernelGGL(k_meas_giw, dim3(C.nbins, C.nflav, (niw + per - 1) / per), dim3(256), lds, st, C, C.M, C.TC, C.TA, C.K, ccap);
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
void launch_meas_g2iw(const DevConst &C, hipStream_t st, int n2, int nwb, double2 *G2)
{
    hipLaunchKernelGGL(k_meas_g2iw, dim3(C.nwalkers, C.nflav), dim3(256), (size_t)3 * C.kmax * G2_FC * sizeof(double2), st, C, n2, nwb, G2);
}
void launch_meas_g4iw(const DevConst &C, hipStream_t st, int nwf, int nwb, int zph, int nb4, const double2 *G2, double2 *acc4, double2 *acc4pp, double2 *acc2)
{
    const int n2 = 2 * (nwf + nwb), nw = 2 * nwf, nreg = (nw + 63) / 64;
    const dim3 grid(nreg * nreg, C.nflav * C.nflav * (2 * nwb + 1), (nb4 + 7) / 8 * 8);       // z: accumulator bins in groups of eight (one per XCD)
    const int nt = (nw + 15) / 16, nch = (2 * nwb + 1 + G4X_WB - 1) / G4X_WB;              // 16 x 16 tiles x chunks of bosonic frequencies
    const unsigned gz = (unsigned)((nb4 + 7) / 8 * 8);
    if (acc4) {
        hipLaunchKernelGGL(k_meas_g4iw, grid, dim3(256), 0, st, C, nwf, nwb, zph, nb4, G2, acc4);
        hipLaunchKernelGGL(k_meas_g4iw_exch, dim3(nt * nt * nch, C.nflav, gz), dim3(256), 0, st, C, nwf, nwb, zph, nb4, G2, acc4);
    }
    if (acc4pp) hipLaunchKernelGGL(k_meas_g4iw_pp, dim3(nt * nt * nch, C.nflav * C.nflav, gz), dim3(256), 0, st, C, nwf, nwb, nb4, G2, acc4pp);
    if (acc2) {
        const size_t ng2 = (size_t)C.nflav * n2 * n2;
        const int per_b4 = (C.nbins + nb4 - 1) / nb4;                  // sample bins per accumulator bin
        hipLaunchKernelGGL(k_meas_g2acc, dim3((unsigned)((ng2 + 255) / 256), nb4, (per_b4 + G2ACC_CHUNK - 1) / G2ACC_CHUNK), dim3(256), 0, st, C, n2, nwb, nb4, G2, acc2);
    }
}
void launch_meas_weq(const DevConst &C, hipStream_t st, int nbe, double *acc_iw, double *acc_tau, double *acc_p3, int n2b, int n3f, int n3b)
{
    hipLaunchKernelGGL(k_meas_weq, dim3(nbe), dim3(256), 0, st, C, nbe, acc_iw, acc_tau, acc_p3, n2b, n3f, n3b);
}
void launch_meas_g4w(const DevConst &C, hipStream_t st, int pp, int nbw)
{
    hipLaunchKernelGGL(k_meas_g4w, dim3(nbw), dim3(256), (size_t)(4 * C.nwf + 2 * C.nwb + 1) * sizeof(double2), st, C, pp, nbw);
}
void launch_meas_wie(const DevConst &C, hipStream_t st, int nbe, double *acc_iw, double *acc_tau, int n1, int n2, int n3)
{
    hipLaunchKernelGGL(k_meas_wie, dim3(nbe), dim3(256), 0, st, C, nbe, acc_iw, acc_tau, n1, n2, n3);
}
void launch_meas_gw(const DevConst &C, hipStream_t st)
{
    hipLaunchKernelGGL(k_meas_gw, dim3(C.nbins), dim3(256), (size_t)C.ntau * sizeof(double), st, C);
}

// MeasSusz = 1 on one-state blocks: <n_1(tauref) n_2(tauref + t beta/Ntau)> (meas_ntau_n0, Trace.F90:6472-6835, where the
// densities are inserted into stored bra / ket vectors; with one state per block they are occupation numbers).  tauref is
// the upper edge of the chain's window; points beyond beta are taken at tauref - (Ntau - t) beta/Ntau as there.
// One workgroup per sample bin, thread = tau point (NT points per thread).  Chain by chain the tau arrays are staged in LDS
// and every thread counts the operators before its times: occupation masks of up to 8 chains stay in registers, then
// acc[f1][f2][t] -= cfgsign [n_f1(tauref)] [n_f2(t)] summed over the chains is added to the bin's accumulator (atomics
// only where several sample bins share an accumulator bin).
template <int NT>
__global__ void __launch_bounds__(256) k_meas_susz(DevConst C, double *__restrict__ acc, int nbs)
{
    extern __shared__ double tops[];             // [2][nflav][kmax]: creator / annihilator times of the current chain
    const int bin = blockIdx.x, ab = bin % nbs, tid = threadIdx.x, nf = C.nflav, nd = C.nd, kmax = C.kmax, ntau = C.ntau;
    const double beta = C.beta, tauD = beta / (double)ntau;
    const bool shared_bin = nbs < C.nbins;
    double *out = acc + (size_t)ab * nf * nf * ntau;
    const int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
    for (int wc = w0; wc < w1; wc += 8) {
        int m0[8], m[8][NT]; double sg[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int w = wc + c;
            m0[c] = 0; sg[c] = 0.0;
#pragma unroll
            for (int i = 0; i < NT; ++i) m[c][i] = 0;
            if (w >= w1) continue;                                   // uniform
            const WalkerScalars &ws = C.ws[w];
            if (ws.sector > 1) continue;                             // chains in worm space measure no Z-space estimator
            __syncthreads();
            for (int f = 0; f < nf; ++f) {
                const int k = C.K[(size_t)w * nf + f];
                for (int e = tid; e < k; e += 256) {
                    tops[(size_t)f * kmax + e] = C.TC[((size_t)w * nf + f) * kmax + e];
                    tops[(size_t)(nf + f) * kmax + e] = C.TA[((size_t)w * nf + f) * kmax + e];
                }
            }
            __syncthreads();
            sg[c] = -C.cfgsign[w];
            const int outer = ws.outer;
            const double tauref = ((double)(ws.wpos + 2) / (double)(C.num_windows + 1)) * beta;       // Trace.F90:1698-1699
            auto mask_at = [&](double t) {
                int mk = 0;
                for (int f = 0; f < nf; ++f) {
                    const int k = C.K[(size_t)w * nf + f];
                    int n = (outer >> ((f >> 1) + nd * (f & 1))) & 1;
                    for (int e = 0; e < k; ++e) n += (tops[(size_t)f * kmax + e] <= t ? 1 : 0) - (tops[(size_t)(nf + f) * kmax + e] <= t ? 1 : 0);
                    mk |= (n & 1) << f;
                }
                return mk;
            };
            m0[c] = mask_at(tauref);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = tid + 256 * i;
                if (t < ntau) {
                    double tauint = tauref + (double)t * tauD;
                    if (tauint > beta) tauint = tauref - (double)(ntau - t) * tauD;
                    m[c][i] = mask_at(tauint);
                }
            }
        }
        for (int f1 = 0; f1 < nf; ++f1)
            for (int f2 = 0; f2 < nf; ++f2) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int t = tid + 256 * i;
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v += (((m0[c] >> f1) & 1) && ((m[c][i] >> f2) & 1)) ? sg[c] : 0.0;
                    if (t < ntau && v != 0.0) {
                        double *o = out + ((size_t)f1 * nf + f2) * ntau + t;
                        if (shared_bin) atomicAdd(o, v); else *o += v;
                    }
                }
            }
    }
}
// Gtau_mean_step, Gtau_mid_step, sign_step (CTQMC.F90:2394-2402): per-measurement series.  The reference keeps one chain's
// series per rank; here entry m is the MEAN over the chains of this process (independent chains: the normalised
// autocorrelation of the mean is that of one chain).  One wave per (chain, flavour): sum of the elements as they go into G(tau)
// (sign of the wrap; the configuration sign enters twice, i.e. not at all), and of those with |tau_a - tau_c| / beta in
// [0.4, 0.6]; the sign entry is the configuration sign of chains that hold at least one pair (the reference sets it inside
// the pair loop).  ser = [mean: nflav x nmeas | mid: nflav x nmeas | sign: nmeas].
__global__ void __launch_bounds__(256) k_meas_steps(DevConst C, double *__restrict__ ser, int nmeas, int m, int what)
{
    const int wv = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63, nf = C.nflav;
    if (wv >= C.nwalkers * nf) return;
    const int w = wv / nf, f = wv - w * nf;
    const int k = C.K[(size_t)w * nf + f];
    if (k == 0) return;
    const size_t wf = (size_t)w * nf + f;
    const double *tc = C.TC + wf * C.kmax, *ta = C.TA + wf * C.kmax, *M = C.M + wf * C.kmax * C.kmax;
    double sm = 0.0, smid = 0.0;
    for (int e = lane; e < k * k; e += 64) {
        const int ic = e / k, ia = e - ic * k;
        double v = M[(size_t)ic * C.kmax + ia];
        const double d = ta[ia] - tc[ic];
        if (!(d > 0.0)) v = -v;
        sm += v;
        const double r = fabs(d) / C.beta;
        if (r >= 0.4 && r <= 0.6) smid += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); smid += __shfl_xor(smid, o, 64); }
    if (lane != 0) return;
    const double inv = 1.0 / (double)C.nwalkers;
    if (what & 1) atomicAdd(&ser[(size_t)f * nmeas + m], sm * inv);
    if (what & 2) atomicAdd(&ser[(size_t)(nf + f) * nmeas + m], smid * inv);
    if (what & 4) {
        bool first = true;
        for (int g = 0; g < f; ++g) first = first && C.K[(size_t)w * nf + g] == 0;
        if (first) atomicAdd(&ser[(size_t)2 * nf * nmeas + m], C.cfgsign[w] * inv);
    }
}
void launch_meas_steps(const DevConst &C, hipStream_t st, double *ser, int nmeas, int m, int what)
{
    const unsigned waves = (unsigned)C.nwalkers * (unsigned)C.nflav;
    hipLaunchKernelGGL(k_meas_steps, dim3((waves + 3) / 4), dim3(256), 0, st, C, ser, nmeas, m, what);
}

void launch_meas_susz(const DevConst &C, hipStream_t st, double *acc, int nbs)
{
    const size_t lds = (size_t)2 * C.nflav * C.kmax * sizeof(double);
    const dim3 grid(C.nbins), block(256);
    if (C.ntau <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz<1>), grid, block, lds, st, C, acc, nbs);
    else if (C.ntau <= 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz<2>), grid, block, lds, st, C, acc, nbs);
    else if (C.ntau <= 1024) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz<4>), grid, block, lds, st, C, acc, nbs);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_meas_susz<8>), grid, block, lds, st, C, acc, nbs);
}

// Moments of the normalised per-bin results x_b[i] = acc[b][i] scale[b] (sqrt(2l+1), l = i mod nleg, for the Legendre
// coefficients) for the cross-rank reduction: out = [ B | S_i = sum_b x_b[i] | sum_b (x_b[i] - S_i/B)^2 | S_i^2/B ].  Every
// entry adds over ranks (parallel-axis formula, statistics.py of this package).  One thread per element and chunk of 64
// bins (unit stride across the wave), three phases in stream order: sums, squared deviations from the mean, the rest.
constexpr int PACK_BINS = 64;
__global__ void __launch_bounds__(256) k_pack_moments(const double *__restrict__ acc, int B, size_t per, const double *__restrict__ scale,
                                                      int nleg, double *__restrict__ out, int phase)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    if (phase == 2) { const double s1 = out[1 + i]; out[1 + 2 * per + i] = s1 * (s1 / (double)B); if (i == 0) out[0] = (double)B; return; }
    const double ef = nleg ? sqrt(2.0 * (double)(i % (size_t)nleg) + 1.0) : 1.0;
    const int b0 = blockIdx.y * PACK_BINS, b1 = min(b0 + PACK_BINS, B);
    const double m = phase == 1 ? out[1 + i] / (double)B : 0.0;
    double a = 0.0;
    for (int b = b0; b < b1; ++b) { const double d = acc[(size_t)b * per + i] * (scale[b] * ef) - m; a += phase == 1 ? d * d : d; }
    atomicAdd(&out[1 + (phase == 1 ? per : 0) + i], a);
}
void launch_pack_moments(hipStream_t st, const double *acc, int B, size_t per, const double *scale, int nleg, double *out)
{
    (void)hipMemsetAsync(out, 0, (1 + 3 * per) * sizeof(double), st);
    const dim3 grid((unsigned)((per + 255) / 256), (unsigned)((B + PACK_BINS - 1) / PACK_BINS));
    hipLaunchKernelGGL(k_pack_moments, grid, dim3(256), 0, st, acc, B, per, scale, nleg, out, 0);
    hipLaunchKernelGGL(k_pack_moments, grid, dim3(256), 0, st, acc, B, per, scale, nleg, out, 1);
    hipLaunchKernelGGL(k_pack_moments, dim3(grid.x), dim3(256), 0, st, acc, B, per, scale, nleg, out, 2);
}
}  // namespace w2dyn
