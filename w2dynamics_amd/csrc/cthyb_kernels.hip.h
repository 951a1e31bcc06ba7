// Device side of libw2dyn_hip.so: batched-walker CT-HYB sampler and estimators for gfx950 (MI355X).
//
// Execution model: ONE WAVEFRONT (64 lanes) PER MARKOV CHAIN.  All control flow of a chain is wave-uniform; lanes
// parallelise over matrix elements / operators.  A 256-thread block holds four independent chains (no block-level
// barrier anywhere).  Chain state stays resident in HBM between launches (structure of arrays over walkers); the
// working set of one step (k_f^2 doubles of one M block, the tau arrays) is L2/L1 resident.
//
// Data layout per walker w (capacity KMAX per flavour):
//   K [w][f]                      expansion order of flavour f
//   TC[w][f][KMAX], TA[...]       creator / annihilator times in SLOT order (insertion order, swap-with-last removal)
//   M [w][f][KMAX*KMAX]           inverse hybridisation matrix, row = creator slot, column = annihilator slot
// Slot order replaces the reference's time-ordered rows/columns (MatrixUpdate.F90:410-482 shifts and reallocates on
// every accepted move): an insertion appends one row/column in place, a removal is a rank-one update plus moving
// the last row/column into the hole, a tau-shift is a pure sign flip.  Time-ordered ranks, where the reference
// needs them (FPos parity, Trace.F90:2828), are obtained by wave-parallel counting.
//
// Local problem: density-density, every block of H_loc holds one Fock state (reference "scalar" trace path,
// Trace.F90:5556-5727).  The local trace ratio of a pair move is evaluated in closed form from occupied-length
// integrals (no time-ordered walk), which is exact for this class of Hamiltonians.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/w2dyn_rng.h"

namespace w2dyn {

constexpr int WCAP = 256;        // max operators inside the sliding window handled by the removal proposal
constexpr int MAXFLAV = 16;

enum Counter { C_TRYADD, C_ACCADD, C_QNADD, C_TRYREM, C_ACCREM, C_QNREM, C_TRYGLOB, C_ACCGLOB, C_QNGLOB,
               C_TRYSHIFT, C_ACCSHIFT, C_SAMPLING, C_NCOUNT };

enum WalkerErr { ERR_NONE = 0, ERR_KMAX = 1, ERR_WCAP = 2, ERR_DETRAT = 4, ERR_SINGULAR = 8 };

struct WalkerScalars {
    unsigned long long ndraw;   // random numbers consumed
    int outer;                  // outer superstate = Fock state at tau = 0/beta
    int npairs;                 // removable pairs in the window, -1 = stale (Trace.F90:1634, 1670)
    int wpos;                   // window position
    int err;
    double det_sign;            // sign of det F in the reference's time order (Trace%Det%sign)
    double det_log;             // log|det F|
    unsigned long long cnt[C_NCOUNT];
    unsigned long long pad[1];
};

struct DevConst {
    int nd, nflav, nstates, nftau, ntau, nleg, niw2, kmax, nwalkers, num_windows, nslide, has_szt, maxhisto;
    int walkers_per_bin, nbins, lds_per_wave;   // lds_per_wave in bytes
    double beta, taudiff_max, p_glob, p_shift, egs;
    unsigned long long seed, first_walker;
    const double *ftab;      // [nflav][nftau]
    const double *energies;  // [nstates]
    const double *udd;       // [nflav][nflav] symmetric density-density couplings, zero diagonal
    const double *eps;       // [nflav] one-body levels (diagonal of muimp as passed, i.e. -mu)
    const double *outer_cum; // [nstates] cumulative normalised empty-trace weights (Trace.F90:655-663)
    const double *outer_w;   // [nstates] normalised empty-trace weights
    // walker state
    WalkerScalars *ws;
    int *K;
    double *TC, *TA, *M, *MB, *TS;     // MB: second matrix buffer (global moves), TS: tau scratch [w][f][2][kmax]
    double *cfgsign;                   // [nwalkers] sign as seen by the estimators (= -sign, CTQMC.F90:3574)
    // accumulators (per bin)
    double *acc_gleg;     // [nbins][nflav][nleg]
    double *acc_gtau;     // [nbins][nflav][ntau]
    double *acc_giw;      // [nbins][nflav][niw2][2]
    double *acc_histo;    // [nbins][nflav][maxhisto+1]
    double *acc_dm;       // [nbins][nstates]
    double *acc_outer;    // [nbins][nstates]
    double *acc_signouter;// [nbins][nstates]
    double *acc_sign;     // [nbins]
    unsigned long long *accpair;  // [nflav][ntau]
};

// ---------------------------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ void wave_sync()
{
    // cross-lane hand-off through LDS / global memory inside ONE wave: order the memory operations for the
    // compiler and drain outstanding stores; no s_barrier is needed (a wave executes in lockstep).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int count_pred(bool p) { return __popcll(__ballot(p)); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Trace.F90:1014-1033 get_HybrF
__device__ __forceinline__ double hybr(const double *ft, int nftau, double beta, double dtau)
{
    double sig = 1.0;
    if (dtau < 0.0) { dtau += beta; sig = -1.0; }
    double n = dtau / beta * (double)(nftau - 1);
    int i = (int)n;
    double a = ft[i], b = ft[i + 1];
    return sig * (a + (n - (double)i) * (b - a));
}

__device__ __forceinline__ int flav_bit(int f, int nd) { return (f >> 1) + nd * (f & 1); }

// ---------------------------------------------------------------------------------------------------------------
// per-wave context
// ---------------------------------------------------------------------------------------------------------------
struct Ctx {
    const DevConst &C;
    int w, lane;
    int *K;
    double *TC, *TA, *M;
    // LDS scratch of this wave
    double *Q, *R, *MQ, *RM, *wt;
    int *wm, *wc;
    // wave-uniform scalars
    unsigned long long ndraw;
    int outer, npairs, wpos, err;
    double det_sign, det_log;
    unsigned long long cnt[C_NCOUNT];
    double wmin, wmax;

    __device__ Ctx(const DevConst &c) : C(c) {}

    __device__ __forceinline__ double draw()
    {
        double u = w2dyn_uniform_at(C.seed, C.first_walker + (unsigned long long)w, ndraw);
        ndraw += 1;
        return u;
    }
    __device__ __forceinline__ int randint(int lo, int hi)
    {
        int n = hi - lo + 1;
        int r = (int)(draw() * (double)n);
        if (r >= n) r = n - 1;
        return lo + r;
    }
    __device__ __forceinline__ void window_bounds()
    {   // Trace.F90:1698-1699
        wmin = ((double)wpos / (double)(C.num_windows + 1)) * C.beta;
        wmax = ((double)(wpos + 2) / (double)(C.num_windows + 1)) * C.beta;
    }
    __device__ __forceinline__ double *tc(int f) const { return TC + (size_t)f * C.kmax; }
    __device__ __forceinline__ double *ta(int f) const { return TA + (size_t)f * C.kmax; }
    __device__ __forceinline__ double *mat(int f) const { return M + (size_t)f * C.kmax * C.kmax; }
};

// pair (x,y) of one creator and one annihilator of equal flavour counts as removable (Trace.F90:1794-1814)
__device__ __forceinline__ bool pair_ok(double x, double y, double wmin, double wmax, double tdm)
{
    double a = fmin(x, y), b = fmax(x, y);
    return (a >= wmin) && (b <= wmax) && !(b > a + tdm);
}

// Collect the operators inside the window into LDS (wt: tau, wm: f | ca<<8 | slot<<16) and count, for each of them,
// its partners at later times (wc).  Returns the total = NPairs (Trace.F90:1770-1815).
__device__ int build_window(Ctx &x)
{
    const DevConst &C = x.C;
    int nw = 0;
    for (int f = 0; f < C.nflav; ++f) {
        int k = x.K[f];
        const double *tcp = x.tc(f), *tap = x.ta(f);
        for (int base = 0; base < 2 * k; base += 64) {
            int idx = base + x.lane;
            bool valid = idx < 2 * k;
            int ca = idx < k ? 1 : 2;
            int slot = idx < k ? idx : idx - k;
            double tau = valid ? (ca == 1 ? tcp[slot] : tap[slot]) : -1.0;
            bool inw = valid && tau >= x.wmin && tau <= x.wmax;
            unsigned long long mask = __ballot(inw);
            int pos = nw + __popcll(mask & ((1ull << x.lane) - 1ull));
            if (inw && pos < WCAP) { x.wt[pos] = tau; x.wm[pos] = f | (ca << 8) | (slot << 16); }
            nw += __popcll(mask);
        }
    }
    nw = uni(nw);
    if (nw > WCAP) { x.err |= ERR_WCAP; nw = WCAP; }
    wave_sync();
    int total = 0;
    for (int e = x.lane; e < ((nw + 63) & ~63); e += 64) {
        int c = 0;
        if (e < nw) {
            double te = x.wt[e]; int me = x.wm[e] & 0xffff;
            for (int q = 0; q < nw; ++q) {
                double tq = x.wt[q]; int mq = x.wm[q] & 0xffff;
                bool partner = ((mq & 0xff) == (me & 0xff)) && ((mq >> 8) != (me >> 8)) && (tq > te) && !(tq > te + C.taudiff_max);
                c += partner ? 1 : 0;
            }
            x.wc[e] = c;
        }
        total += c;
    }
    total = wave_sum(total);
    wave_sync();
    x.wt[WCAP] = (double)nw;   // stash n_w for the pick
    return uni(total);
}

// integral over (lo,hi) of  eps_f + sum_g U_fg n_g(tau)   for the current configuration (flavour f itself excluded)
__device__ double level_integral(Ctx &x, int f, double lo, double hi)
{
    const DevConst &C = x.C;
    double part = 0.0;
    double base = C.eps[f];
    for (int g = 0; g < C.nflav; ++g) {
        if (g == f) continue;
        double u = C.udd[f * C.nflav + g];
        if (u == 0.0) continue;
        int k = x.K[g];
        if ((x.outer >> flav_bit(g, C.nd)) & 1) base += u;
        const double *tcp = x.tc(g), *tap = x.ta(g);
        for (int e = x.lane; e < k; e += 64) {
            double c = fmin(fmax(tcp[e], lo), hi), a = fmin(fmax(tap[e], lo), hi);
            part += u * ((hi - c) - (hi - a));
        }
    }
    return base * (hi - lo) + wave_sum(part);
}

__device__ __forceinline__ void accpair_add(Ctx &x, int f, double dt)
{
    const DevConst &C = x.C;
    if (x.lane == 0) {
        int bin = (int)floor(dt / C.beta * (double)C.ntau) + 1;
        if (bin > C.ntau) bin = C.ntau;
        atomicAdd(&C.accpair[(size_t)f * C.ntau + (bin - 1)], 1ull);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pair insertion: CTQMC.F90:3353-3612 (StepAdd), Trace.F90:3657-3698, 2775-2832, MatrixUpdate.F90:392-442
// ---------------------------------------------------------------------------------------------------------------
__device__ void step_add(Ctx &x)
{
    const DevConst &C = x.C;
    x.cnt[C_TRYADD] += 1;
    double width1 = x.wmax - x.wmin;
    double tau1 = x.wmin + width1 * x.draw();                 // annihilator
    int orb = x.randint(1, C.nd);
    int spin = x.randint(1, 2);
    double tmin = fmax(tau1 - C.taudiff_max, x.wmin);
    double width2 = fmin(tau1 + C.taudiff_max, x.wmax) - tmin;
    double tau2 = tmin + width2 * x.draw();                   // creator
    double tdf = width1 * width2;
    int f = 2 * (orb - 1) + (spin - 1);
    int k = x.K[f];
    const int kmax = C.kmax;
    double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
    const double *ft = C.ftab + (size_t)f * C.nftau;

    double lo = fmin(tau1, tau2), hi = fmax(tau1, tau2);
    int c_lo = 0, a_lo = 0, c_hi = 0, a_hi = 0, pc = 0, pa = 0, eq = (tau1 == tau2);
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        c_lo += count_pred(tce < lo); a_lo += count_pred(tae < lo);
        c_hi += count_pred(tce < hi); a_hi += count_pred(tae < hi);
        pc += count_pred(tce < tau2); pa += count_pred(tae < tau1);
        eq |= __ballot(v && (tce == tau1 || tce == tau2 || tae == tau1 || tae == tau2)) != 0ull;
        if (v) {   // Q_j = F(tau_a_j - tau_c_new), R_i = F(tau_a_new - tau_c_i)   (Trace.F90:2814-2817)
            x.Q[e] = hybr(ft, C.nftau, C.beta, tae - tau2);
            x.R[e] = hybr(ft, C.nftau, C.beta, tau1 - tce);
        }
    }
    int between = (c_hi + a_hi) - (c_lo + a_lo);
    int occ_lo = ((x.outer >> flav_bit(f, C.nd)) & 1) + c_lo - a_lo;
    bool creator_first = tau2 < tau1;
    bool valid = !eq && between == 0 && occ_lo == (creator_first ? 0 : 1);
    if (!valid) { x.cnt[C_QNADD] += 1; return; }
    if (k >= kmax) { x.err |= ERR_KMAX; x.cnt[C_QNADD] += 1; return; }

    wave_sync();
    double acc = 0.0;
    for (int e = x.lane; e < k * k; e += 64) {
        int i = e / k, j = e - i * k;
        acc += x.R[i] * Mf[i * kmax + j] * x.Q[j];
    }
    double S = hybr(ft, C.nftau, C.beta, tau1 - tau2) - wave_sum(acc);
    double sig = ((pc - pa) & 1) ? -1.0 : 1.0;

    double I = level_integral(x, f, lo, hi);
    double dlogtr = creator_first ? -I : I;
    double rnd = x.draw();

    if (x.npairs < 0) x.npairs = build_window(x);
    int dnp = 0;
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        dnp += count_pred(v && pair_ok(tau2, tap[e], x.wmin, x.wmax, C.taudiff_max));
        dnp += count_pred(v && pair_ok(tcp[e], tau1, x.wmin, x.wmax, C.taudiff_max));
    }
    dnp += pair_ok(tau2, tau1, x.wmin, x.wmax, C.taudiff_max) ? 1 : 0;
    int np_new = x.npairs + dnp;

    double prob = fabs(tdf * (double)(2 * C.nd) / (double)np_new * exp(dlogtr + log(fabs(S))));
    if (!(rnd < prob)) return;

    // accepted: M' = [[M + MQ RM^T / S, -MQ/S], [-RM^T/S, 1/S]] with the new row/column appended at slot k
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        if (e < k) {
            double rm = 0.0, mq = 0.0;
            for (int i = 0; i < k; ++i) rm += x.R[i] * Mf[i * kmax + e];
            for (int j = 0; j < k; ++j) mq += Mf[e * kmax + j] * x.Q[j];
            x.RM[e] = rm; x.MQ[e] = mq;
        }
    }
    wave_sync();
    double invS = 1.0 / S;
    for (int e = x.lane; e < k * k; e += 64) {
        int i = e / k, j = e - i * k;
        Mf[i * kmax + j] += S * (-x.MQ[i] * invS) * (-x.RM[j] * invS);
    }
    for (int e = x.lane; e < k; e += 64) {
        Mf[e * kmax + k] = -x.MQ[e] * invS;
        Mf[k * kmax + e] = -x.RM[e] * invS;
    }
    if (x.lane == 0) { Mf[k * kmax + k] = invS; tcp[k] = tau2; tap[k] = tau1; x.K[f] = k + 1; }
    wave_sync();
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = np_new;
    x.cnt[C_ACCADD] += 1;
    accpair_add(x, f, hi - lo);
}

// ---------------------------------------------------------------------------------------------------------------
// pair removal: CTQMC.F90:3892-4150 (StepRem), Trace.F90:4851-4950, 2937-2963, MatrixUpdate.F90:445-482
// ---------------------------------------------------------------------------------------------------------------
__device__ void step_rem(Ctx &x)
{
    const DevConst &C = x.C;
    x.cnt[C_TRYREM] += 1;
    // the window list is needed for the pick in any case
    int np = build_window(x);
    x.npairs = np;
    if (np <= 0) { x.cnt[C_QNREM] += 1; return; }
    int pick = x.randint(1, np);
    int nw = (int)x.wt[WCAP];

    // first operator of the pick-th pair in the reference's enumeration (time order of the earlier operator)
    int estar = -1, rnk = 0;
    for (int base = 0; base < nw; base += 64) {
        int e = base + x.lane;
        int P = 0; bool hit = false;
        if (e < nw) {
            double te = x.wt[e];
            for (int q = 0; q < nw; ++q) P += (x.wt[q] < te) ? x.wc[q] : 0;
            hit = (P < pick) && (pick <= P + x.wc[e]);
        }
        unsigned long long m = __ballot(hit);
        if (m) { int src = __ffsll((long long)m) - 1; estar = base + src; rnk = pick - __shfl(P, src, 64); }
    }
    estar = uni(estar); rnk = uni(rnk);
    double t1 = x.wt[estar]; int m1 = x.wm[estar];
    int qstar = -1;
    for (int base = 0; base < nw; base += 64) {
        int q = base + x.lane;
        bool hit = false;
        if (q < nw) {
            double tq = x.wt[q]; int mq = x.wm[q];
            bool partner = ((mq & 0xff) == (m1 & 0xff)) && (((mq >> 8) & 0xff) != ((m1 >> 8) & 0xff)) && (tq > t1) && !(tq > t1 + C.taudiff_max);
            if (partner) {
                int r = 0;
                for (int p = 0; p < nw; ++p) {
                    double tp = x.wt[p]; int mp = x.wm[p];
                    bool pp = ((mp & 0xff) == (m1 & 0xff)) && (((mp >> 8) & 0xff) != ((m1 >> 8) & 0xff)) && (tp > t1) && !(tp > t1 + C.taudiff_max);
                    r += (pp && tp < tq) ? 1 : 0;
                }
                hit = (r == rnk - 1);
            }
        }
        unsigned long long m = __ballot(hit);
        if (m) qstar = base + __ffsll((long long)m) - 1;
    }
    qstar = uni(qstar);
    double t2 = x.wt[qstar]; int m2 = x.wm[qstar];
    int f = m1 & 0xff;
    int ca1 = (m1 >> 8) & 0xff;
    int ic = (ca1 == 1) ? (m1 >> 16) : (m2 >> 16);
    int ia = (ca1 == 2) ? (m1 >> 16) : (m2 >> 16);
    double tau_c = (ca1 == 1) ? t1 : t2, tau_a = (ca1 == 2) ? t1 : t2;
    double hybf = (double)np / ((x.wmax - x.wmin) *
                                (fmin(tau_a + C.taudiff_max, x.wmax) - fmax(tau_a - C.taudiff_max, x.wmin)));

    int k = x.K[f];
    const int kmax = C.kmax;
    double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
    int between = 0, pc = 0, pa = 0;
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        between += count_pred(v && tce > t1 && tce < t2) + count_pred(v && tae > t1 && tae < t2);
        pc += count_pred(tce < tau_c); pa += count_pred(tae < tau_a);
    }
    if (between != 0) { x.cnt[C_QNREM] += 1; return; }

    double S = Mf[ic * kmax + ia];
    double sig = ((pc - pa) & 1) ? -1.0 : 1.0;
    bool creator_first = tau_c < tau_a;
    // the pair's own operators sit on the boundary of (t1,t2) and belong to flavour f, which level_integral skips
    double I = level_integral(x, f, t1, t2);
    double dlogtr = creator_first ? I : -I;
    double prob = fabs(hybf / (double)(2 * C.nd) * exp(dlogtr + log(fabs(S))));
    double rnd = x.draw();
    if (!(rnd < prob)) return;
    if (fabs(S) < 1e-12) { x.err |= ERR_DETRAT; return; }   // CTQMC.F90:4062 stops here

    for (int e = x.lane; e < k; e += 64) { x.RM[e] = Mf[ic * kmax + e]; x.MQ[e] = Mf[e * kmax + ia]; }
    wave_sync();
    const int last = k - 1;
    for (int e = x.lane; e < k * k; e += 64) {
        int i = e / k, j = e - i * k;
        if (i == ic || j == ia) continue;
        double v = Mf[i * kmax + j] - x.MQ[i] * x.RM[j] / S;
        int id = (i == last) ? ic : i, jd = (j == last) ? ia : j;
        Mf[id * kmax + jd] = v;
    }
    if (x.lane == 0) { tcp[ic] = tcp[last]; tap[ia] = tap[last]; x.K[f] = last; }
    wave_sync();
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = -1;
    x.cnt[C_ACCREM] += 1;
    accpair_add(x, f, t2 - t1);
}

// ---------------------------------------------------------------------------------------------------------------
// tau-shift: CTQMC.F90:4557-4835, Trace.F90:2355-2457; empty configuration: StepChangeOuter 4570-4638
// ---------------------------------------------------------------------------------------------------------------
__device__ void step_change_outer(Ctx &x)
{
    const DevConst &C = x.C;
    double r = x.draw();
    int sel = C.nstates - 1;
    bool found = false;
    for (int base = 0; base < C.nstates && !found; base += 64) {
        int s = base + x.lane;
        bool hit = s < C.nstates && C.outer_cum[s] >= r && C.outer_w[s] != 0.0;
        unsigned long long m = __ballot(hit);
        if (m) { sel = base + __ffsll((long long)m) - 1; found = true; }
    }
    x.outer = uni(sel);
}

__device__ void step_shift(Ctx &x)
{
    const DevConst &C = x.C;
    int nops = 0;
    for (int f = 0; f < C.nflav; ++f) nops += 2 * x.K[f];
    if (nops == 0) { step_change_outer(x); return; }
    double dt = x.draw() * C.beta;
    if (dt == 0.0 || dt == C.beta) return;
    x.cnt[C_TRYSHIFT] += 1;
    const int kmax = C.kmax;
    int outer = x.outer;
    double detrat = 1.0;
    for (int f = 0; f < C.nflav; ++f) {
        int k = x.K[f];
        if (k == 0) continue;
        double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
        int wcnt = 0, wacnt = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane;
            bool v = e < k;
            wcnt += count_pred(v && tcp[e] < dt); wacnt += count_pred(v && tap[e] < dt);
        }
        if ((wcnt + wacnt) & 1) outer ^= 1 << flav_bit(f, C.nd);
        if (!((wcnt == 0 && wacnt == 0) || (wcnt == k && wacnt == k))) {
            for (int e = x.lane; e < k * k; e += 64) {
                int i = e / k, j = e - i * k;
                double s = ((tcp[i] < dt) ? -1.0 : 1.0) * ((tap[j] < dt) ? -1.0 : 1.0);
                if (s < 0.0) Mf[i * kmax + j] = -Mf[i * kmax + j];
            }
        }
        if ((k & 1) && ((wcnt + wacnt) & 1)) detrat = -detrat;
        wave_sync();
        for (int e = x.lane; e < k; e += 64) {
            double a = tcp[e] - dt; if (a < 0.0) a += C.beta; tcp[e] = a;
            double b = tap[e] - dt; if (b < 0.0) b += C.beta; tap[e] = b;
        }
    }
    wave_sync();
    (void)x.draw();    // acceptance draw; |tr'/tr| = 1 for a cyclic shift (CTQMC.F90:4769-4775)
    x.outer = outer;
    x.det_sign *= detrat;
    {   // CTQMC.F90:4744-4746
        int nw = C.num_windows;
        int a = (int)floor((dt / C.beta) * (double)(nw + 1)); a = ((a % nw) + nw) % nw;
        x.wpos = (((x.wpos - a) % nw) + nw) % nw;
        x.window_bounds();
    }
    x.npairs = -1;
    x.cnt[C_ACCSHIFT] += 1;
}

// ---------------------------------------------------------------------------------------------------------------
// global moves: CTQMC.F90:4433-4554, Trace.F90:1083-1198, 1306-1433, 2126-2188, 2279-2333
// ---------------------------------------------------------------------------------------------------------------

// log of the scalar local trace of the current configuration, up to the constant Egs*beta:
//   -int_0^beta E(n(tau)) dtau  with  E(n) = sum_f eps_f n_f + sum_{f<g} U_fg n_f n_g
// n_f(tau) = n_f(0) + #creators before tau - #annihilators before tau.  Pair sums over operators of different
// flavours are distributed over the lanes.
__device__ double full_logtrace(Ctx &x, const int *Kv, const double *TCb, const double *TAb, int outer)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    const double beta = C.beta;
    double part = 0.0, uni_part = 0.0;
    // per-flavour occupied length L_f = n0 beta + sum_c (beta - tc) - sum_a (beta - ta)
    for (int f = 0; f < C.nflav; ++f) {
        int k = Kv[f];
        double n0 = (double)((outer >> flav_bit(f, C.nd)) & 1);
        uni_part += C.eps[f] * n0 * beta;
        for (int e = x.lane; e < k; e += 64)
            part += C.eps[f] * ((beta - TCb[(size_t)f * kmax + e]) - (beta - TAb[(size_t)f * kmax + e]));
    }
    // overlaps O_fg = int n_f n_g
    for (int f = 0; f < C.nflav; ++f) {
        for (int g = f + 1; g < C.nflav; ++g) {
            double u = C.udd[f * C.nflav + g];
            if (u == 0.0) continue;
            int kf = Kv[f], kg = Kv[g];
            double nf0 = (double)((outer >> flav_bit(f, C.nd)) & 1), ng0 = (double)((outer >> flav_bit(g, C.nd)) & 1);
            uni_part += u * nf0 * ng0 * beta;
            const double *cf = TCb + (size_t)f * kmax, *af = TAb + (size_t)f * kmax;
            const double *cg = TCb + (size_t)g * kmax, *ag = TAb + (size_t)g * kmax;
            // n_f0 * sum_y s_y (beta - tau_y) and n_g0 * sum_x s_x (beta - tau_x)
            for (int e = x.lane; e < kg; e += 64) part += u * nf0 * ((beta - cg[e]) - (beta - ag[e]));
            for (int e = x.lane; e < kf; e += 64) part += u * ng0 * ((beta - cf[e]) - (beta - af[e]));
            // sum over operator pairs s_x s_y (beta - max(tau_x, tau_y)); x in {c_f, a_f}, y in {c_g, a_g}
            int nx = 2 * kf, ny = 2 * kg;
            for (int e = x.lane; e < nx * ny; e += 64) {
                int ix = e / ny, iy = e - ix * ny;
                double tx = ix < kf ? cf[ix] : af[ix - kf]; double sx = ix < kf ? 1.0 : -1.0;
                double ty = iy < kg ? cg[iy] : ag[iy - kg]; double sy = iy < kg ? 1.0 : -1.0;
                part += u * sx * sy * (beta - fmax(tx, ty));
            }
        }
    }
    return -(uni_part + wave_sum(part));
}

// Build F(tau_a_i - tau_c_j) of one flavour in `A` (row = annihilator slot, column = creator slot, stride kmax) and
// invert it in place by Gauss-Jordan with partial pivoting; the result read as [creator][annihilator] is M.
// Returns log|det F| and its sign in slot order.  What the reference obtains from dgetrf/dgetri
// (Trace.F90:2279-2333, MatrixUpdate.F90:41-99).
__device__ void build_and_invert(Ctx &x, int f, int k, const double *tcp, const double *tap, double *A,
                                 double &logdet, double &sgn)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    const double *ft = C.ftab + (size_t)f * C.nftau;
    logdet = 0.0; sgn = 1.0;
    if (k == 0) return;
    for (int e = x.lane; e < k * k; e += 64) {
        int a = e / k, c = e - a * k;
        A[a * kmax + c] = hybr(ft, C.nftau, C.beta, tap[a] - tcp[c]);
    }
    wave_sync();
    int *piv = x.wm;                 // k <= kmax <= WCAP entries of LDS
    for (int p = 0; p < k; ++p) {
        // pivot search in column p, rows p..k-1
        double best = -1.0; int bi = p;
        for (int r = p + x.lane; r < k; r += 64) { double v = fabs(A[r * kmax + p]); if (v > best) { best = v; bi = r; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        bi = uni(bi);
        if (!(best > 0.0)) { x.err |= ERR_SINGULAR; return; }
        if (x.lane == 0) piv[p] = bi;
        if (bi != p) {
            for (int j = x.lane; j < k; j += 64) { double t = A[p * kmax + j]; A[p * kmax + j] = A[bi * kmax + j]; A[bi * kmax + j] = t; }
            sgn = -sgn;
            wave_sync();
        }
        double d = A[p * kmax + p];
        logdet += log(fabs(d)); if (d < 0.0) sgn = -sgn;
        double id = 1.0 / d;
        // scaled pivot row (with the inverse's entry in column p) -> LDS; pivot column -> LDS
        for (int j = x.lane; j < k; j += 64) { x.RM[j] = (j == p) ? id : A[p * kmax + j] * id; x.MQ[j] = A[j * kmax + p]; }
        wave_sync();
        for (int e = x.lane; e < k * k; e += 64) {
            int r = e / k, j = e - r * k;
            double v;
            if (r == p) v = x.RM[j];
            else v = ((j == p) ? 0.0 : A[r * kmax + j]) - x.MQ[r] * x.RM[j];
            A[r * kmax + j] = v;
        }
        wave_sync();
    }
    for (int p = k - 1; p >= 0; --p) {    // undo the row interchanges as column interchanges
        int bi = piv[p];
        if (bi != p) {
            for (int r = x.lane; r < k; r += 64) { double t = A[r * kmax + p]; A[r * kmax + p] = A[r * kmax + bi]; A[r * kmax + bi] = t; }
            wave_sync();
        }
    }
}

// parity (+1/-1) of the permutation that sorts `t[0..k)` (number of inversions)
__device__ double sort_parity(Ctx &x, const double *t, int k)
{
    int inv = 0;
    for (int e = x.lane; e < k * k; e += 64) { int i = e / k, j = e - i * k; inv += (i < j && t[i] > t[j]) ? 1 : 0; }
    return (wave_sum(inv) & 1) ? -1.0 : 1.0;
}

__device__ void step_glob(Ctx &x)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd, kmax = C.kmax;
    x.cnt[C_TRYGLOB] += 1;
    int nops = 0;
    for (int f = 0; f < nf; ++f) nops += 2 * x.K[f];
    if (nops == 0) return;
    int gu[MAXFLAV + 1];
    double r = x.draw();
    bool spinflip = false;
    if (r < 0.3) { spinflip = true; gu[0] = 0; for (int i = 1; i <= nd; ++i) gu[i] = i + nd; for (int i = nd + 1; i <= 2 * nd; ++i) gu[i] = i - nd; }
    else if (r < 0.6) { gu[0] = 1; for (int i = 1; i <= nf; ++i) gu[i] = i; }
    else {
        for (int i = 1; i <= nf; ++i) gu[i] = i;
        for (int j = 1; j <= nf; ++j) { int i = x.randint(j, nf); int t = gu[j]; gu[j] = gu[i]; gu[i] = t; }
        gu[0] = x.randint(0, 1);
    }
    // relabelled copy of the tau arrays in scratch: TS[g][0] = creators, TS[g][1] = annihilators of new flavour g
    double *TS = C.TS + (size_t)x.w * nf * 2 * kmax;
    int Kn[MAXFLAV];
    for (int f = 0; f < nf; ++f) {
        int num = (f >> 1) + nd * (f & 1) + 1;
        int t = gu[num];
        int g = 2 * ((t - 1) % nd) + (t - 1) / nd;
        int k = x.K[f];
        Kn[g] = k;
        const double *srcc = gu[0] ? x.ta(f) : x.tc(f), *srca = gu[0] ? x.tc(f) : x.ta(f);
        for (int e = x.lane; e < k; e += 64) { TS[(size_t)(2 * g) * kmax + e] = srcc[e]; TS[(size_t)(2 * g + 1) * kmax + e] = srca[e]; }
    }
    wave_sync();
    // quantum numbers: occupancy of flavour g at tau=0 is fixed by its first operator, free if it has none
    int fixed = 0, freemask = 0;
    for (int g = 0; g < nf; ++g) {
        int k = Kn[g];
        if (k == 0) { freemask |= 1 << flav_bit(g, nd); continue; }
        double mc = 1e300, ma = 1e300;
        for (int e = x.lane; e < k; e += 64) { mc = fmin(mc, TS[(size_t)(2 * g) * kmax + e]); ma = fmin(ma, TS[(size_t)(2 * g + 1) * kmax + e]); }
        mc = wave_min(mc); ma = wave_min(ma);
        if (ma < mc) fixed |= 1 << flav_bit(g, nd);
    }
    int outer_new;
    double sst_factor = 1.0;
    if (spinflip && C.has_szt) {
        int down = x.outer & ((1 << nd) - 1), up = (x.outer >> nd) & ((1 << nd) - 1);
        outer_new = up | (down << nd);
        if ((outer_new & ~freemask) != fixed) sst_factor = 0.0;
    } else {
        // possible_ssts_old = possible_ssts_new = 2^(#flavours without operators): factor 1 (Trace.F90:1381)
        int nfree = __popc(freemask);
        int kk = x.randint(1, 1 << nfree) - 1;
        outer_new = fixed;
        for (int b = 0, used = 0; b < 2 * nd; ++b) if ((freemask >> b) & 1) { if ((kk >> used) & 1) outer_new |= 1 << b; ++used; }
    }
    if (sst_factor == 0.0) { x.cnt[C_QNGLOB] += 1; return; }

    // new inverse matrices into the second buffer, determinant in the reference's time order
    double *MBw = C.MB + (size_t)x.w * nf * kmax * kmax;
    double logdet_new = 0.0, sign_new = 1.0;
    for (int g = 0; g < nf; ++g) {
        double ld, sg;
        const double *tcg = TS + (size_t)(2 * g) * kmax, *tag = TS + (size_t)(2 * g + 1) * kmax;
        build_and_invert(x, g, Kn[g], tcg, tag, MBw + (size_t)g * kmax * kmax, ld, sg);
        if (Kn[g] > 0) sg *= sort_parity(x, tcg, Kn[g]) * sort_parity(x, tag, Kn[g]);
        logdet_new += ld; sign_new *= sg;
    }
    int Ko[MAXFLAV];
    for (int f = 0; f < nf; ++f) Ko[f] = x.K[f];
    double ltr_old = full_logtrace(x, Ko, x.TC, x.TA, x.outer);
    // TS is laid out [g][2][kmax]; full_logtrace wants separate creator/annihilator bases with flavour stride kmax:
    // pass TS with stride trick: creators at TS + 2g*kmax -> use a strided view through temporary pointers
    double ltr_new;
    {
        // copy the relabelled taus into TC/TA only on acceptance; evaluate the new trace from TS via a local view
        // (flavour stride 2*kmax): done by a second variant of the integrals below
        const double beta = C.beta;
        double part = 0.0, up = 0.0;
        for (int f = 0; f < nf; ++f) {
            int k = Kn[f];
            double n0 = (double)((outer_new >> flav_bit(f, nd)) & 1);
            up += C.eps[f] * n0 * beta;
            const double *cf = TS + (size_t)(2 * f) * kmax, *af = TS + (size_t)(2 * f + 1) * kmax;
            for (int e = x.lane; e < k; e += 64) part += C.eps[f] * ((beta - cf[e]) - (beta - af[e]));
        }
        for (int f = 0; f < nf; ++f) for (int g = f + 1; g < nf; ++g) {
            double u = C.udd[f * nf + g];
            if (u == 0.0) continue;
            int kf = Kn[f], kg = Kn[g];
            double nf0 = (double)((outer_new >> flav_bit(f, nd)) & 1), ng0 = (double)((outer_new >> flav_bit(g, nd)) & 1);
            up += u * nf0 * ng0 * beta;
            const double *cf = TS + (size_t)(2 * f) * kmax, *af = TS + (size_t)(2 * f + 1) * kmax;
            const double *cg = TS + (size_t)(2 * g) * kmax, *ag = TS + (size_t)(2 * g + 1) * kmax;
            for (int e = x.lane; e < kg; e += 64) part += u * nf0 * ((beta - cg[e]) - (beta - ag[e]));
            for (int e = x.lane; e < kf; e += 64) part += u * ng0 * ((beta - cf[e]) - (beta - af[e]));
            int nx = 2 * kf, ny = 2 * kg;
            for (int e = x.lane; e < nx * ny; e += 64) {
                int ix = e / ny, iy = e - ix * ny;
                double tx = ix < kf ? cf[ix] : af[ix - kf]; double sx = ix < kf ? 1.0 : -1.0;
                double ty = iy < kg ? cg[iy] : ag[iy - kg]; double sy = iy < kg ? 1.0 : -1.0;
                part += u * sx * sy * (beta - fmax(tx, ty));
            }
        }
        ltr_new = -(up + wave_sum(part));
    }
    double rnd = x.draw();
    if (x.err & ERR_SINGULAR) return;
    double ratio = fabs(exp((ltr_new - ltr_old) + (logdet_new - x.det_log)) * sst_factor);
    if (!(rnd < ratio)) return;
    // accept: install relabelled taus and new matrices
    for (int g = 0; g < nf; ++g) {
        int k = Kn[g];
        double *tcp = x.tc(g), *tap = x.ta(g), *Mg = x.mat(g);
        const double *Mn = MBw + (size_t)g * kmax * kmax;
        for (int e = x.lane; e < k; e += 64) { tcp[e] = TS[(size_t)(2 * g) * kmax + e]; tap[e] = TS[(size_t)(2 * g + 1) * kmax + e]; }
        for (int e = x.lane; e < k * k; e += 64) { int i = e / k, j = e - i * k; Mg[i * kmax + j] = Mn[i * kmax + j]; }
        if (x.lane == 0) x.K[g] = k;
    }
    wave_sync();
    x.outer = outer_new;
    x.det_log = logdet_new; x.det_sign = sign_new;
    x.npairs = -1;
    x.cnt[C_ACCGLOB] += 1;
}

// ---------------------------------------------------------------------------------------------------------------
// sweep kernel: nsteps iterations of the Z-space loop (CTQMC.F90:8213-8257 / 8428-8533) for every chain
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ctx_setup(Ctx &x, unsigned char *smem)
{
    const DevConst &C = x.C;
    int wib = threadIdx.x >> 6;
    x.lane = threadIdx.x & 63;
    x.w = uni((int)(blockIdx.x * (blockDim.x >> 6) + wib));
    unsigned char *base = smem + (size_t)wib * C.lds_per_wave;
    x.Q = (double *)base; x.R = x.Q + C.kmax; x.MQ = x.R + C.kmax; x.RM = x.MQ + C.kmax;
    x.wt = x.RM + C.kmax;                 // WCAP + 1 doubles
    x.wm = (int *)(x.wt + WCAP + 2); x.wc = x.wm + WCAP;
}

__device__ __forceinline__ void ctx_load(Ctx &x)
{
    const DevConst &C = x.C;
    int w = x.w;
    x.K = C.K + (size_t)w * C.nflav;
    x.TC = C.TC + (size_t)w * C.nflav * C.kmax;
    x.TA = C.TA + (size_t)w * C.nflav * C.kmax;
    x.M = C.M + (size_t)w * C.nflav * C.kmax * C.kmax;
    const WalkerScalars &s = C.ws[w];
    x.ndraw = s.ndraw; x.outer = s.outer; x.npairs = s.npairs; x.wpos = s.wpos; x.err = s.err;
    x.det_sign = s.det_sign; x.det_log = s.det_log;
    for (int i = 0; i < C_NCOUNT; ++i) x.cnt[i] = s.cnt[i];
    x.window_bounds();
}

__device__ __forceinline__ void ctx_store(Ctx &x)
{
    if (x.lane == 0) {
        WalkerScalars &s = x.C.ws[x.w];
        s.ndraw = x.ndraw; s.outer = x.outer; s.npairs = x.npairs; s.wpos = x.wpos; s.err = x.err;
        s.det_sign = x.det_sign; s.det_log = x.det_log;
        for (int i = 0; i < C_NCOUNT; ++i) s.cnt[i] = x.cnt[i];
    }
}

__global__ __launch_bounds__(256) void k_sweep(DevConst C, long long nsteps, int count_sampling)
{
    extern __shared__ unsigned char smem[];
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load(x);
    const double pg = C.p_glob, ps = C.p_shift;
    for (long long it = 0; it < nsteps; ++it) {
        if (count_sampling) x.cnt[C_SAMPLING] += 1;
        double r = x.draw();
        if (r < (1.0 - ps - pg) / 2.0) { (void)x.draw(); (void)x.draw(); step_add(x); }
        else if (r <= 1.0 - ps - pg) { (void)x.draw(); (void)x.draw(); step_rem(x); }
        else if (r <= 1.0 - ps) step_glob(x);
        else step_shift(x);
        unsigned long long tries = x.cnt[C_TRYADD] + x.cnt[C_TRYREM];
        if (tries % (unsigned long long)C.nslide == 0ull) {       // shift_window, Trace.F90:1652-1678
            if (C.num_windows > 1) {
                x.wpos += 1; if (x.wpos == C.num_windows) x.wpos = 0;
                x.window_bounds(); x.npairs = -1;
            } else x.wpos = 0;
        }
    }
    ctx_store(x);
}

// rebuild M, det and NPairs of every chain from its tau arrays (set_mc_config, Trace.F90:6972-6999)
__global__ __launch_bounds__(256) void k_rebuild(DevConst C)
{
    extern __shared__ unsigned char smem[];
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load(x);
    double ld = 0.0, sg = 1.0;
    for (int f = 0; f < C.nflav; ++f) {
        double l, s;
        int k = x.K[f];
        build_and_invert(x, f, k, x.tc(f), x.ta(f), x.mat(f), l, s);
        if (k > 0) s *= sort_parity(x, x.tc(f), k) * sort_parity(x, x.ta(f), k);
        ld += l; sg *= s;
    }
    x.det_log = ld; x.det_sign = sg;
    x.npairs = build_window(x);
    ctx_store(x);
}

// ---------------------------------------------------------------------------------------------------------------
// estimators
// ---------------------------------------------------------------------------------------------------------------

// Sign, expansion-order histogram, outer-superstate histograms and density matrix at beta/2: one wave per chain.
// CTQMC.F90:3343-3350 (MeasSign), 3264-3282 (MeasExpOrder), 3285-3339 (MeasOuterHistograms), 3088-3151.
// Sign = Trace%sign * Det%sign * BlockSign * permaltca (Trace.F90:3479-3618), each factor expressed as the parity of
// a count over operator pairs so that no time-ordered list is needed:
//   trace sign   : pairs (i,j) with tau_j < tau_i and Fock bit(f_j) < bit(f_i)          (Operator.F90:158-171)
//   block sign   : pairs (i,j) with tau_j > tau_i and flavour index f_j < f_i             (Trace.F90:3594-3618)
//   permaltca    : for every creator, operators of its flavour at later times; +1 if N/2 is odd (3554-3588)
__global__ __launch_bounds__(256) void k_meas_misc(DevConst C)
{
    extern __shared__ unsigned char smem[];
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load(x);
    const int nf = C.nflav, nd = C.nd, kmax = C.kmax;
    // flat operator index space: op = (f, ca, slot); offsets in LDS ints
    int off[MAXFLAV + 1]; off[0] = 0;
    for (int f = 0; f < nf; ++f) off[f + 1] = off[f] + 2 * x.K[f];
    int N = off[nf];
    int par = 0;
    // pair loop: every lane takes operators i = lane, lane+64, ... and scans all j
    for (int i = x.lane; i < N; i += 64) {
        int fi = 0; while (i >= off[fi + 1]) ++fi;
        int li = i - off[fi]; int ki = x.K[fi];
        bool ci = li < ki;
        double ti = ci ? x.TC[(size_t)fi * kmax + li] : x.TA[(size_t)fi * kmax + (li - ki)];
        int biti = flav_bit(fi, nd);
        for (int fj = 0; fj < nf; ++fj) {
            int kj = x.K[fj];
            int bitj = flav_bit(fj, nd);
            bool wtr = bitj < biti, wbl = fj < fi, wsame = (fj == fi) && ci;
            if (!(wtr || wbl || wsame)) continue;
            for (int lj = 0; lj < 2 * kj; ++lj) {
                double tj = lj < kj ? x.TC[(size_t)fj * kmax + lj] : x.TA[(size_t)fj * kmax + (lj - kj)];
                if (wtr && tj < ti) par ^= 1;
                if (wbl && tj > ti) par ^= 1;
                if (wsame && tj > ti) par ^= 1;
            }
        }
    }
    par = wave_sum(par) & 1;
    if ((N / 2) & 1) par ^= 1;
    double sign = x.det_sign * (par ? -1.0 : 1.0);
    double cfgsign = -sign;
    // state alive at beta/2 (Trace.F90:6334: operators with tau <= beta/2 have acted)
    int sb2 = x.outer;
    for (int f = 0; f < nf; ++f) {
        int k = x.K[f], c = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane; bool v = e < k;
            c += count_pred(v && x.TC[(size_t)f * kmax + e] <= 0.5 * C.beta) + count_pred(v && x.TA[(size_t)f * kmax + e] <= 0.5 * C.beta);
        }
        if (c & 1) sb2 ^= 1 << flav_bit(f, nd);
    }
    if (x.lane == 0) {
        int bin = x.w / C.walkers_per_bin;
        C.cfgsign[x.w] = cfgsign;
        atomicAdd(&C.acc_sign[bin], cfgsign);
        atomicAdd(&C.acc_outer[(size_t)bin * C.nstates + x.outer], 1.0);
        atomicAdd(&C.acc_signouter[(size_t)bin * C.nstates + x.outer], cfgsign);
        atomicAdd(&C.acc_dm[(size_t)bin * C.nstates + sb2], -cfgsign);
        for (int f = 0; f < nf; ++f) {
            int k = x.K[f]; int h = k < C.maxhisto ? k : C.maxhisto;
            atomicAdd(&C.acc_histo[((size_t)bin * nf + f) * (C.maxhisto + 1) + h], 1.0);
        }
    }
}

// Legendre and tau-bin estimators (MeasGtauRem, CTQMC.F90:2252-2440): THE data-parallel kernel.
// One wave per (bin, flavour).  The wave walks the chains of its bin; lanes take (creator, annihilator) pairs; every
// lane runs the Legendre three-term recursion for its pair and accumulates m*P_l into NLEG register accumulators,
// which are reduced across the wave once per launch and added to the bin's accumulator (no atomics: one owner).
template <int NLEG_TILE>
__global__ __launch_bounds__(64) void k_meas_gleg(DevConst C)
{
    const int bin = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const int kmax = C.kmax, nleg = C.nleg;
    const double beta = C.beta;
    double acc[NLEG_TILE];
#pragma unroll
    for (int l = 0; l < NLEG_TILE; ++l) acc[l] = 0.0;
    double *gt = C.acc_gtau + ((size_t)bin * C.nflav + f) * C.ntau;
    int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
    for (int w = w0; w < w1; ++w) {
        int k = C.K[(size_t)w * C.nflav + f];
        if (k == 0) continue;
        const double *tcp = C.TC + ((size_t)w * C.nflav + f) * kmax, *tap = C.TA + ((size_t)w * C.nflav + f) * kmax;
        const double *Mf = C.M + ((size_t)w * C.nflav + f) * kmax * kmax;
        double cs = C.cfgsign[w];
        for (int e = lane; e < k * k; e += 64) {
            int ic = e / k, ia = e - ic * k;
            double m = Mf[ic * kmax + ia];
            double d = tap[ia] - tcp[ic];
            if (!(d > 0.0)) { d = beta + tap[ia] - tcp[ic]; m = -m; }
            m *= cs;
            int idt = (int)(d / beta * (double)C.ntau) + 1;
            if (idt > C.ntau) idt = C.ntau;
            atomicAdd(&gt[idt - 1], m);
            double xx = 2.0 / beta * d - 1.0;
            double plm1 = 1.0, pl = xx;
            acc[0] += m;
            if (NLEG_TILE > 1) acc[1] += m * xx;
#pragma unroll
            for (int il = 2; il < NLEG_TILE; ++il) {
                const int l = il - 1;
                const double a = (2.0 * (double)l + 1.0) / (double)(l + 1), b = (double)l / (double)(l + 1);
                double pn = a * xx * pl - b * plm1;
                if (il < nleg) acc[il] += m * pn;
                plm1 = pl; pl = pn;
            }
        }
    }
    double *gl = C.acc_gleg + ((size_t)bin * C.nflav + f) * nleg;
#pragma unroll
    for (int l = 0; l < NLEG_TILE; ++l) {
        double v = wave_sum(acc[l]);
        if (lane == 0 && l < nleg) gl[l] += v;
    }
}

// Matsubara estimator (measure_giw_nfft + naive get_giw, CTQMC.F90:1015-1056, Nfft_z.F90:376-414):
// giw(n) += sum_ij M(c_j,a_i) exp(-i w_n tau_c_j) exp(+i w_n tau_a_i), no configuration sign.
// One wave per (bin, flavour, frequency chunk of 64): lane = frequency; the pair loop is wave-uniform, so M and the
// taus are broadcast loads.  u_i = sum_c M(c,a) e^{-i w tau_c} first (k^2 complex MACs), then sum_a u_a e^{i w tau_a}.
__global__ __launch_bounds__(64) void k_meas_giw(DevConst C)
{
    const int bin = blockIdx.x, f = blockIdx.y, ifr = blockIdx.z * 64 + threadIdx.x;   // 0-based frequency index
    if (ifr >= C.niw2) return;
    const int kmax = C.kmax;
    const double freq = M_PI / C.beta * (double)(2 * (ifr + 1) - C.niw2 - 1);
    double re = 0.0, im = 0.0;
    int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
    for (int w = w0; w < w1; ++w) {
        int k = C.K[(size_t)w * C.nflav + f];
        const double *tcp = C.TC + ((size_t)w * C.nflav + f) * kmax, *tap = C.TA + ((size_t)w * C.nflav + f) * kmax;
        const double *Mf = C.M + ((size_t)w * C.nflav + f) * kmax * kmax;
        for (int ic = 0; ic < k; ++ic) {
            double sc, cc; sincos(-freq * tcp[ic], &sc, &cc);
            double ur = 0.0, ui = 0.0;
            for (int ia = 0; ia < k; ++ia) {
                double sa, ca; sincos(freq * tap[ia], &sa, &ca);
                double m = Mf[ic * kmax + ia];
                ur += m * ca; ui += m * sa;
            }
            re += cc * ur - sc * ui;
            im += cc * ui + sc * ur;
        }
    }
    double *g = C.acc_giw + (((size_t)bin * C.nflav + f) * C.niw2 + ifr) * 2;
    g[0] += re; g[1] += im;
}

}  // namespace w2dyn
