// Worm sampling in sector 4 (two-particle Green's function, NW = 4 worm operators c c+ c c+), sector 2 (one-particle
// Green's function, NW = 2 worm operators c c+; NOperWorm, CTQMC.F90:5282-5285) and the sectors with equal-time pairs 6 / 7
// (P2 ph / pp: two random times) and 8 / 9 (P3 ph / pp: three random times; NW = 4, the sector is DevConst::isector):
// StepWormAdd / StepWormRem (CTQMC.F90:4992-5249), StepWormReplace with propose_/do_WormReplace (CTQMC.F90:5809-6297,
// Trace.F90:3054-3478), the mixed Z / worm-space loop (CTQMC.F90:8428-8533) and the G4iw worm estimator with the exact
// transform (CTQMC.F90:1618-1701, Nfft_worm.F90:94-137).  Included by cthyb_kernels.hip.h inside namespace w2dyn.
//
// The four worm operators c c+ c c+ (wormContainer order, flavours from the component index) carry no hybridisation
// lines: they enter the local trace and the quantum-number checks but not M.  They are kept as eight wave-uniform
// scalars (x.wtau, x.wfl); the slot-ordered arrays and M hold hybridising operators only.
//
// Trace ratio of inserting the worm into a configuration of hybridising operators (outer state unchanged):
//   dn_f(tau)  = (# worm creators of flavour f before tau) - (# worm annihilators of f before tau)
//   -Delta     = sum_f int dn_f (eps_f + sum_{g != f} U_fg n_g) + sum_{f<g} U_fg int dn_f dn_g
// with n_g the occupations from the hybridising operators alone; the first term is a sum of level integrals over the
// (at most three) intervals between consecutive worm operators, the second involves the worm times only.
#pragma once

// the two-pair moves (cthyb_xmoves.hip.h, included after this file) that the worm kernels with XM make
template <int MAT> __device__ W2DYN_INLINE void step_add4(Ctx &x);
template <int MAT> __device__ W2DYN_INLINE void step_rem4(Ctx &x);

// sort of the NW worm operators by time: ord[m] = container index of the m-th operator in time
template <int NW>
__device__ __forceinline__ void worm_order(const double (&t)[NWMAX], int (&ord)[4])
{
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) r += (t[j] < t[i] || (t[j] == t[i] && j < i)) ? 1 : 0;
#pragma unroll
        for (int m = 0; m < NW; ++m) if (r == m) ord[m] = i;
    }
}

// number of hybridising creators / annihilators of flavour f before time t, and whether any lies inside (lo, hi)
__device__ __forceinline__ void worm_counts(Ctx &x, int f, double t, int &nc, int &na)
{
    const int k = x.order(f);
    const double *tcp = x.tc(f), *tap = x.ta(f);
    nc = 0; na = 0;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        nc += count_pred(v && tcp[e] < t); na += count_pred(v && tap[e] < t);
    }
}
__device__ __forceinline__ bool worm_any_between(Ctx &x, int f, double lo, double hi)
{
    const int k = x.order(f);
    const double *tcp = x.tc(f), *tap = x.ta(f);
    bool any = false;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        any |= __ballot(v && ((tcp[e] > lo && tcp[e] < hi) || (tap[e] > lo && tap[e] < hi))) != 0ull;
    }
    return any;
}

// Quantum-number check and log trace ratio (with worm) / (without worm) for worm operators at times t[] with
// flavours fl[] on top of the present hybridising operators.  At most two distinct flavours can occur in a valid
// worm (every flavour needs as many worm creators as annihilators).
// The same for worms of more than four operators (sectors 5 and 10: six): up to NW / 2 distinct flavours, each with its own running
// occupation change d; -Delta = sum_f int d_f (eps_f + sum_{g != f} U_fg n_g) + sum_{f < g} U_fg int d_f d_g over the intervals
// between consecutive worm operators.
template <int NW>
__device__ W2DYN_INLINE bool worm_delta_multi(Ctx &x, const double (&t)[NWMAX], const int (&fl)[NWMAX], double &dlog)
{
    const DevConst &C = x.C;
    constexpr int NF = NW / 2;
    int rk[NW];                                            // rank in time of container operator i
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) r += (t[j] < t[i] || (t[j] == t[i] && j < i)) ? 1 : 0;
        rk[i] = r;
    }
    int fs[NF], d[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) { fs[q] = -1; d[q] = 0; }
    dlog = 0.0;
    bool valid = true;
#pragma unroll
    for (int m = 0; m < NW; ++m) {
        if (NW <= 8 && !valid) continue;                   // wave-uniform: nothing after the first violation is looked at (dlog is not used then); twelve operators: the loop would no longer unroll
        int i = 0; double tm = 0.0, tn = 0.0; int fm = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) { if (rk[q] == m) { i = q; tm = t[q]; fm = fl[q]; } if (rk[q] == m + 1) tn = t[q]; }
        const bool creator = (i & 1) != 0;                 // container order c, c+, c, c+, ...
        // the flavour's slot: the first one that holds it or is free (used slots come first)
        int slot = -1;
#pragma unroll
        for (int q = 0; q < NF; ++q) if (slot < 0 && (fs[q] < 0 || fs[q] == fm)) { slot = q; fs[q] = fm; }
        if (slot < 0) valid = false;                       // more flavours than can be balanced
        int nc, na;
        worm_counts(x, fm, tm, nc, na);
        int dcur = 0;
#pragma unroll
        for (int q = 0; q < NF; ++q) if (q == slot) dcur = d[q];
        const int occ = ((x.outer >> flav_bit(fm, C.nd)) & 1) + nc - na + dcur;
        if (occ != (creator ? 0 : 1)) valid = false;
#pragma unroll
        for (int q = 0; q < NF; ++q) if (q == slot) d[q] += creator ? 1 : -1;
        if (m < NW - 1) {
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                if (valid && d[q] != 0) {
                    if (worm_any_between(x, fs[q], tm, tn)) valid = false;
                    else dlog -= (double)d[q] * level_integral<0>(x, fs[q], tm, tn);
                }
            }
#pragma unroll
            for (int q = 0; q < NF; ++q)
#pragma unroll
                for (int r = q + 1; r < NF; ++r)
                    if (valid && d[q] != 0 && d[r] != 0) dlog -= C.udd[fs[q] * C.nflav + fs[r]] * (double)(d[q] * d[r]) * (tn - tm);
        }
    }
#pragma unroll
    for (int q = 0; q < NF; ++q) if (d[q] != 0) valid = false;
    return valid;
}

template <int NW>
__device__ W2DYN_INLINE bool worm_delta(Ctx &x, const double (&t)[NWMAX], const int (&fl)[NWMAX], double &dlog)
{
    if constexpr (NW > 4) return worm_delta_multi<NW>(x, t, fl, dlog);
    const DevConst &C = x.C;
    int ord[4] = {0, 0, 0, 0};
    worm_order<NW>(t, ord);
    dlog = 0.0;
    int fa = -1, fb = -1, da = 0, db = 0;
    bool valid = true;
#pragma unroll
    for (int m = 0; m < NW; ++m) {
        if (!valid) continue;                              // wave-uniform; dlog is not used after a violation
        int i = 0; double tm = 0.0; int fm = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) if (ord[m] == q) { i = q; tm = t[q]; fm = fl[q]; }
        const bool creator = (i & 1) != 0;                 // container order c, c+, c, c+
        if (fa < 0 || fm == fa) fa = fm;
        else if (fb < 0 || fm == fb) fb = fm;
        else valid = false;                                // a third flavour cannot be balanced
        int nc, na;
        worm_counts(x, fm, tm, nc, na);
        const int dcur = (fm == fa) ? da : db;
        const int occ = ((x.outer >> flav_bit(fm, C.nd)) & 1) + nc - na + dcur;
        if (occ != (creator ? 0 : 1)) valid = false;
        if (fm == fa) da += creator ? 1 : -1; else db += creator ? 1 : -1;
        if (m < NW - 1) {
            double tn = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) if (ord[m + 1] == q) tn = t[q];
            if (valid && da != 0) { if (worm_any_between(x, fa, tm, tn)) valid = false; else dlog -= (double)da * level_integral<0>(x, fa, tm, tn); }
            if (valid && db != 0) { if (worm_any_between(x, fb, tm, tn)) valid = false; else dlog -= (double)db * level_integral<0>(x, fb, tm, tn); }
            if (valid && da != 0 && db != 0) dlog -= C.udd[fa * C.nflav + fb] * (double)(da * db) * (tn - tm);
        }
    }
    if (da != 0 || db != 0) valid = false;
    return valid;
}

// get_Proposal (CTQMC.F90:5655-5798): SectorG / SectorG4 (beta Nd)^NW eta; P2 / P2pp beta^2 Nd^4 eta; P3 / P3pp beta^3 Nd^4 eta; the sectors
// with U-matrix factors: 3 / 5 beta^(N-2) Nd^(N+1) eta, 10 beta^2 Nd^8 eta, 11 beta^4 Nd^12 eta, 12 / 13 beta^3 Nd^8 eta, 14 / 15 / 16 beta^2 Nd^4 eta,
// each times the factor below.
// U-matrix factor of a worm: g[i] = flavour of worm operator i + 1, u_i = i-th flavour index of the component (u_g, set_IEOperFlavor,
// Trace.F90:3835-3883: two digits in sectors 3 and 10, four otherwise); UA1 / UA2 / UAB = the U matrix antisymmetrised in its first pair, its second
// pair, both (CTQMC.F90:7926-7948).  meas: the expression whose sign enters the estimator, which differs from the proposal's in sector 13
// (CTQMC.F90:1919-1921 against 5764-5766).
__device__ __forceinline__ double ie_ufac(const DevConst &C, int sec, const int *g, bool meas)
{
    const int nf = C.nflav;
    int u1, u2, u3 = 0, u4 = 0;
    {
        int rest = C.icomponent - 1;
        if (sec == 3 || sec == 10) { u2 = rest % nf; u1 = (rest / nf) % nf; }
        else { u4 = rest % nf; rest /= nf; u3 = rest % nf; rest /= nf; u2 = rest % nf; u1 = (rest / nf) % nf; }
    }
    auto UM = [&](int a, int b, int c, int d) { return C.umat[(((size_t)a * nf + b) * nf + c) * nf + d]; };
    auto UA1 = [&](int a, int b, int c, int d) { return 0.5 * (UM(a, b, c, d) - UM(b, a, c, d)); };
    auto UA2 = [&](int a, int b, int c, int d) { return 0.5 * (UM(a, b, c, d) - UM(a, b, d, c)); };
    auto UAB = [&](int a, int b, int c, int d) { return 0.5 * (UA1(a, b, c, d) - UA1(a, b, d, c)); };
    auto q1 = [&](int u, int o) { return 0.5 * (UM(u, g[o + 1], g[o], g[o + 2]) - UM(g[o + 1], u, g[o], g[o + 2])); };        // operators o+1, o+2, o+3 (1-based) = c c+ c
    auto q2 = [&](int u, int o) { return 0.5 * (UM(g[o + 2], g[o], u, g[o + 1]) - UM(g[o + 2], g[o], g[o + 1], u)); };        // operators o+1, o+2, o+3 = c+ c c+
    switch (sec) {
    case 3: case 5: return q1(u1, 0);
    case 10: return q1(u1, 0) * q2(u2, 3);
    case 11: return q1(u1, 0) * q2(u2, 3) * q1(u3, 6) * q2(u4, 9);
    case 12: return UA1(u1, g[1], g[0], g[2]) * UA2(g[3], g[5], u2, g[4]) * UAB(u3, g[7], u4, g[6]);
    case 13: return meas ? UA1(u1, g[1], g[0], g[2]) * UA1(u3, g[3], g[4], g[6]) * UA2(g[5], g[7], u2, u4)
                         : UA1(u1, g[1], g[0], g[2]) * UA1(u3, g[3], g[5], g[4]) * UA2(g[7], g[6], u2, u4);
    case 14: return UAB(u1, g[1], u2, g[0]) * UAB(u3, g[3], u4, g[2]);
    case 15: return UA1(u1, u3, g[0], g[2]) * UA2(g[1], g[3], u2, u4);       // the proposal carries a minus sign on top (CTQMC.F90:5791)
    case 16: return UA1(u1, g[1], g[0], g[2]) * UAB(u3, g[3], u2, u4);
    default: return 1.0;
    }
}
__device__ __forceinline__ bool worm_has_ufac(int sec) { return sec == 3 || sec == 5 || (sec >= 10 && sec <= 16); }
// sector 3 (its sign enters the estimator, CTQMC.F90:1138)
__device__ __forceinline__ double gsigma_ufac(const DevConst &C, const int *fl) { return ie_ufac(C, 3, fl, false); }

template <int NW>
__device__ __forceinline__ double worm_prefact_ie(const DevConst &C, const int (&fl)[NWMAX])
{
    const double nd = (double)C.nd, nd4 = nd * nd * nd * nd, b2 = C.beta * C.beta;
    const int sec = C.isector;
    if constexpr (NW == 4) {
        if (sec == 3) return ie_ufac(C, 3, fl, false) * b2 * (nd4 * nd) * C.worm_eta;        // beta^(N-2) Nd^(N+1) eta times the U-matrix factor, N = 4 (CTQMC.F90:5696)
        if (sec >= 6 && sec <= 9) return (sec >= 8 ? b2 * C.beta : b2) * nd4 * C.worm_eta;
        if (sec == 14) return ie_ufac(C, 14, fl, false) * C.worm_eta * b2 * nd4;             // 5778-5782
        if (sec == 15) return -ie_ufac(C, 15, fl, false) * C.worm_eta * b2 * nd4;            // 5791-5795
        if (sec == 16) return ie_ufac(C, 16, fl, false) * b2 * nd4 * C.worm_eta;             // 5706-5709
    }
    if constexpr (NW == 6) {
        if (sec == 5) return ie_ufac(C, 5, fl, false) * (b2 * b2) * (nd4 * nd * nd * nd) * C.worm_eta;
        return ie_ufac(C, 10, fl, false) * b2 * (nd4 * nd4) * C.worm_eta;                    // 5731
    }
    if constexpr (NW == 8) return ie_ufac(C, sec, fl, false) * C.worm_eta * (b2 * C.beta) * (nd4 * nd4);    // 5757, 5768
    if constexpr (NW == 12) return ie_ufac(C, 11, fl, false) * (b2 * b2) * (nd4 * nd4 * nd4) * C.worm_eta;  // 5745
    const double b = C.beta * nd, bb = b * b;
    return (NW == 4 ? bb * bb : bb) * C.worm_eta;
}

// set_OperTime (Trace.F90:3942-4020): independent times in sectors 2 and 4; sectors 6 / 7: tau_1 and tau_4 random, the partners
// an offset beta 1e-14 below tau_1 and above tau_4 (P2: 2 with 1, 3 with 4; P2pp: 3 with 1, 2 with 4); sectors 8 / 9: a third
// random time for operator 2 (P3) or 3 (P3pp), the other one an offset above tau_4.  check_EqualTime (an operator between the
// two of a pair: needs two times within 1e-14 beta of each other) is not looked for.
// Sectors with U-matrix factors (Trace.F90:3971-3983, 4022-4158): sectors 3, 5, 10, 11 draw a time for EVERY operator and then put the
// three-operator objects (1, 3, 2), (5, 4, 6), (7, 9, 8), (11, 10, 12) an offset apart below their last operator's time; 12 and 13 draw three,
// 14 - 16 two times.
template <int NW>
__device__ __forceinline__ void worm_draw_times_ie(Ctx &x, double (&t)[NWMAX])
{
    const DevConst &C = x.C;
    const double beta = C.beta, off = beta * 1e-14;
    const int sec = C.isector;
    double *const tt = &t[0];            // operators beyond the fourth (only where NWMAX = 12)
    if constexpr (NW >= 4) {
        if (sec == 3 || sec == 5 || sec == 10 || sec == 11) {
#pragma unroll
            for (int i = 0; i < NW; ++i) t[i] = x.draw() * beta;
            if (t[1] < 2.0 * off) t[1] = 2.0 * off;
            tt[2] = tt[1] - off; tt[0] = tt[1] - 2.0 * off;
            if constexpr (NW >= 6) if (sec >= 10) {
                if (tt[5] < 2.0 * off) tt[5] = 2.0 * off;
                tt[3] = tt[5] - off; tt[4] = tt[5] - 2.0 * off;
            }
            if constexpr (NW == 12) {
                if (tt[7] < 2.0 * off) tt[7] = 2.0 * off;
                if (tt[11] < 2.0 * off) tt[11] = 2.0 * off;
                tt[8] = tt[7] - off; tt[6] = tt[7] - 2.0 * off; tt[9] = tt[11] - off; tt[10] = tt[11] - 2.0 * off;
            }
            return;
        }
    }
    if constexpr (NW == 8) {
        tt[1] = x.draw() * beta; tt[3] = x.draw() * beta;
        double third = x.draw() * beta;
        if (tt[1] < 2.0 * off) tt[1] = 2.0 * off;
        if (tt[3] < 2.0 * off) tt[3] = 2.0 * off;
        if (third < off) third = off;
        tt[2] = tt[1] - off; tt[0] = tt[1] - 2.0 * off;
        if (sec == 12) { tt[7] = third; tt[5] = tt[3] - off; tt[4] = tt[3] - 2.0 * off; tt[6] = tt[7] - off; }
        else { tt[5] = third; tt[6] = tt[3] - off; tt[4] = tt[3] - 2.0 * off; tt[7] = tt[5] - off; }
        return;
    }
    if constexpr (NW == 4) {
        if (sec >= 14) {
            t[1] = x.draw() * beta;
            double second = x.draw() * beta;
            if (second < off) second = off;
            if (sec == 16) {
                if (t[1] < 2.0 * off) t[1] = 2.0 * off;
                t[3] = second; t[2] = t[1] - off; t[0] = t[1] - 2.0 * off;
            } else {
                if (t[1] < off) t[1] = off;
                if (sec == 14) { t[3] = second; t[0] = t[1] - off; t[2] = t[3] - off; } else { t[2] = second; t[0] = t[2] - off; t[3] = t[1] - off; }
            }
            return;
        }
        if (sec >= 6) {
            t[0] = x.draw() * beta;
            t[3] = x.draw() * beta;
            if (sec <= 7) {
                if (t[0] < off) t[0] = off;
                if (t[3] > beta - off) t[3] = beta - off;
                if (sec == 6) { t[1] = t[0] - off; t[2] = t[3] + off; } else { t[1] = t[3] + off; t[2] = t[0] - off; }
            } else {
                if (t[3] > beta - off) t[3] = beta - off;
                if (sec == 8) { t[1] = x.draw() * beta; t[2] = t[3] + off; } else { t[1] = t[3] + off; t[2] = x.draw() * beta; }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        t[i] = x.draw() * C.beta;
        if (i > 0 && t[i - 1] == t[i]) t[i] += C.beta * 1e-14;
    }
}

// flavours of the worm operators: digits of the component (set_OperFlavor, Trace.F90:3804-3826); sector 3: the three operators of the
// equal-time object get random flavours until the U-matrix factor of the proposal does not vanish, operator 4 carries the component's
// second flavour (set_IEOperFlavor, Trace.F90:3835-3919; CTQMC.F90:5047-5054)
template <int NW>
__device__ __forceinline__ void worm_draw_flavours_ie(Ctx &x, int (&fl)[NWMAX])
{
    const DevConst &C = x.C;
    const int nf = C.nflav, sec = C.isector;
    if constexpr (NW >= 4) {
        if (worm_has_ufac(sec)) {
            // sectors 3 and 5: the operators after the third carry the component's flavours after the first; every other operator gets a random
            // orbital and a random spin, again and again until the proposal's factor does not vanish (the host has checked that some choice has one)
            int d2 = 0, d3 = 0, d4 = 0;
            { int rest = C.icomponent - 1; d4 = rest % nf; rest /= nf; d3 = rest % nf; rest /= nf; d2 = rest % nf; }
            for (int guard = 0; guard < (1 << 20); ++guard) {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    if ((sec == 3 || sec == 5) && i >= 3) fl[i] = sec == 3 ? d4 : (i == 3 ? d2 : (i == 4 ? d3 : d4));
                    else { const int orb = x.randint(1, C.nd), spin = x.randint(1, 2); fl[i] = 2 * (orb - 1) + (spin - 1); }
                }
                if (worm_prefact_ie<NW>(C, fl) != 0.0) break;
            }
            return;
        }
    }
    const int ncomp = NW == 4 ? nf * nf * nf * nf : nf * nf;
    int comp = C.icomponent;
    if (comp == 0) comp = x.randint(1, ncomp);
    int rest = comp - 1, base = ncomp / nf;
#pragma unroll
    for (int i = 0; i < NW; ++i) { const int bs = rest / base; rest -= bs * base; base /= nf; fl[i] = bs; }
}

// The kernels of the sectors 2, 3, 4 and 6 - 9 (IE = false) keep the code they had before the sectors with U-matrix factors beyond sector 3 came
// (IE = true: 5, 10 - 16): the run-time branches of the new sectors cost the NW = 4 kernels 2.3 % (C5: 4.65 -> 4.75 ms per launch) when both
// shared one body, so each set has its own instantiation.
template <int NW>
__device__ __forceinline__ double worm_prefact_base(const DevConst &C, const int (&fl)[NWMAX])
{
    if constexpr (NW == 4) {
        if (C.isector == 3) {                          // beta^(N-2) Nd^(N+1) eta times the U-matrix factor, N = 4 (CTQMC.F90:5696)
            const double nd = (double)C.nd, nd4 = nd * nd * nd * nd;
            return gsigma_ufac(C, fl) * (C.beta * C.beta) * (nd4 * nd) * C.worm_eta;
        }
        if (C.isector >= 6) {
            const double nd = (double)C.nd, nd4 = nd * nd * nd * nd, bb = C.beta * C.beta;
            return (C.isector >= 8 ? bb * C.beta : bb) * nd4 * C.worm_eta;
        }
    }
    const double b = C.beta * (double)C.nd, b2 = b * b;
    return (NW == 4 ? b2 * b2 : b2) * C.worm_eta;
}

template <int NW>
__device__ __forceinline__ void worm_draw_times_base(Ctx &x, double (&t)[NWMAX])
{
    const DevConst &C = x.C;
    if constexpr (NW == 4) {
        if (C.isector == 3) {                          // four draws, then the equal-time object (1, 3, 2) below tau_2 (Trace.F90:3971-3983)
            const double beta = C.beta, off = beta * 1e-14;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = x.draw() * beta;
            if (t[1] < 2.0 * off) t[1] = 2.0 * off;
            t[2] = t[1] - off; t[0] = t[1] - 2.0 * off;
            return;
        }
        if (C.isector >= 6) {
            const double beta = C.beta, off = beta * 1e-14;
            t[0] = x.draw() * beta;
            t[3] = x.draw() * beta;
            if (C.isector <= 7) {
                if (t[0] < off) t[0] = off;
                if (t[3] > beta - off) t[3] = beta - off;
                if (C.isector == 6) { t[1] = t[0] - off; t[2] = t[3] + off; } else { t[1] = t[3] + off; t[2] = t[0] - off; }
            } else {
                if (t[3] > beta - off) t[3] = beta - off;
                if (C.isector == 8) { t[1] = x.draw() * beta; t[2] = t[3] + off; } else { t[1] = t[3] + off; t[2] = x.draw() * beta; }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        t[i] = x.draw() * C.beta;
        if (i > 0 && t[i - 1] == t[i]) t[i] += C.beta * 1e-14;
    }
}

template <int NW>
__device__ __forceinline__ void worm_draw_flavours_base(Ctx &x, int (&fl)[NWMAX])
{
    const DevConst &C = x.C;
    const int nf = C.nflav;
    if constexpr (NW == 4) {
        if (C.isector == 3) {
            fl[3] = (C.icomponent - 1) % nf;
            for (int guard = 0; guard < (1 << 20); ++guard) {      // the host has checked that some choice has a non-vanishing factor
#pragma unroll
                for (int i = 0; i < 3; ++i) { const int orb = x.randint(1, C.nd), spin = x.randint(1, 2); fl[i] = 2 * (orb - 1) + (spin - 1); }
                if (gsigma_ufac(C, fl) != 0.0) break;
            }
            return;
        }
    }
    const int ncomp = NW == 4 ? nf * nf * nf * nf : nf * nf;
    int comp = C.icomponent;
    if (comp == 0) comp = x.randint(1, ncomp);
    int rest = comp - 1, base = ncomp / nf;
#pragma unroll
    for (int i = 0; i < NW; ++i) { const int bs = rest / base; rest -= bs * base; base /= nf; fl[i] = bs; }
}

template <int NW, bool IE>
__device__ __forceinline__ double worm_prefact(const DevConst &C, const int (&fl)[NWMAX])
{
    if constexpr (IE) return worm_prefact_ie<NW>(C, fl); else return worm_prefact_base<NW>(C, fl);
}
template <int NW, bool IE>
__device__ __forceinline__ void worm_draw_times(Ctx &x, double (&t)[NWMAX])
{
    if constexpr (IE) worm_draw_times_ie<NW>(x, t); else worm_draw_times_base<NW>(x, t);
}
template <int NW, bool IE>
__device__ __forceinline__ void worm_draw_flavours(Ctx &x, int (&fl)[NWMAX])
{
    if constexpr (IE) worm_draw_flavours_ie<NW>(x, fl); else worm_draw_flavours_base<NW>(x, fl);
}

template <int NW, bool IE = false>
__device__ W2DYN_INLINE void step_worm_add(Ctx &x)
{
    const DevConst &C = x.C;
    x.count(C_TRYWADD);
    double t[NWMAX] = {}; int fl[NWMAX] = {};
    worm_draw_flavours<NW, IE>(x, fl);
    worm_draw_times<NW, IE>(x, t);
    double dlog;
    const bool valid = worm_delta<NW>(x, t, fl, dlog);
    const double prefact = worm_prefact<NW, IE>(C, fl);
    if (prefact == 0.0 || !valid) { x.count(C_QNWADD); x.sector = 1; return; }
    const double rnd = x.draw();
    if (rnd < fabs(prefact * exp(dlog))) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { x.wtau[i] = t[i]; x.wfl[i] = fl[i]; }
        x.sector = C.isector;
        x.count(C_ACCWADD);
    } else x.sector = 1;
}

template <int NW, bool IE = false>
__device__ W2DYN_INLINE void step_worm_rem(Ctx &x)
{
    const DevConst &C = x.C;
    x.count(C_TRYWREM);
    double dlog;
    const bool valid = worm_delta<NW>(x, x.wtau, x.wfl, dlog);
    if (!valid) { x.count(C_QNWREM); return; }
    const double rnd = x.draw();
    if (rnd < fabs(exp(-dlog) / worm_prefact<NW, IE>(C, x.wfl))) { x.sector = 1; x.count(C_ACCWREM); }
}

// One-state blocks: log ratio of the local trace when the two worm operators p1, p2 (container indices) move from their present
// times to tn1, tn2 while every other operator stays (replacement in the three-operator object of sector 3).  The change of the
// occupations is that of a virtual worm V on top of the PRESENT configuration: the operators at their new times plus, at the old
// times, operators of the opposite type that undo the old ones.  With D_g(tau) the occupation change of flavour g (at most two
// flavours), the new sequence is allowed iff no other operator of g lies where D_g != 0 and the occupation in front of each moved
// operator's new place is what its type needs; log ratio = - sum_g int D_g (eps_g + sum_g' U_gg' n_g') - U_gg' int D_g D_g' with
// n from all present operators, worm operators included (level_integral<4>).
template <int NW = 4>
__device__ W2DYN_INLINE bool worm_shift_delta(Ctx &x, int p1, int p2, double tn1, double tn2, double &dlog)
{
    const DevConst &C = x.C;
    double vt[4]; int vf[4], vs[4]; bool vreal[4];          // time, flavour, +1 creator / -1 annihilator, real (new place) or undoing (old place)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = q == 0 ? p1 : p2; const double tn = q == 0 ? tn1 : tn2;
        double told = 0.0; int fp = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) if (i == p) { told = x.wtau[i]; fp = x.wfl[i]; }
        const int sgn = (p & 1) ? 1 : -1;
        vt[2 * q] = tn; vf[2 * q] = fp; vs[2 * q] = sgn; vreal[2 * q] = true;
        vt[2 * q + 1] = told; vf[2 * q + 1] = fp; vs[2 * q + 1] = -sgn; vreal[2 * q + 1] = false;
    }
    int ord[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) r += (vt[j] < vt[i] || (vt[j] == vt[i] && j < i)) ? 1 : 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (r == m) ord[m] = i;
    }
    const int fa = vf[0], fb = vf[2];                        // flavours of the two operators (may coincide)
    int da = 0, db = 0;
    dlog = 0.0;
    bool valid = true;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double tm = 0.0; int fm = 0, sm = 0; bool rm = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ord[m] == q) { tm = vt[q]; fm = vf[q]; sm = vs[q]; rm = vreal[q]; }
        if (rm) {                                            // occupation in front of the new place: all present operators, then the change so far
            int nc, na;
            worm_counts(x, fm, tm, nc, na);
#pragma unroll
            for (int i = 0; i < NW; ++i) if (x.wfl[i] == fm && x.wtau[i] < tm) { if (i & 1) nc += 1; else na += 1; }
            const int occ = ((x.outer >> flav_bit(fm, C.nd)) & 1) + nc - na + (fm == fa ? da : db);
            if (occ != (sm > 0 ? 0 : 1)) valid = false;
        }
        if (fm == fa) da += sm; else db += sm;
        if (m < 3) {
            double tn = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (ord[m + 1] == q) tn = vt[q];
            const int dA = da, dB = (fb == fa) ? 0 : db;       // equal flavours: everything is in da
            if (valid && dA != 0) {
                bool blocked = worm_any_between(x, fa, tm, tn);
#pragma unroll
                for (int i = 0; i < NW; ++i) if (i != p1 && i != p2 && x.wfl[i] == fa && x.wtau[i] > tm && x.wtau[i] < tn) blocked = true;
                if (blocked) valid = false; else dlog -= (double)dA * level_integral<NW>(x, fa, tm, tn);
            }
            if (valid && dB != 0) {
                bool blocked = worm_any_between(x, fb, tm, tn);
#pragma unroll
                for (int i = 0; i < NW; ++i) if (i != p1 && i != p2 && x.wfl[i] == fb && x.wtau[i] > tm && x.wtau[i] < tn) blocked = true;
                if (blocked) valid = false; else dlog -= (double)dB * level_integral<NW>(x, fb, tm, tn);
            }
            if (valid && dA != 0 && dB != 0) dlog -= C.udd[fa * C.nflav + fb] * (double)(dA * dB) * (tn - tm);
        }
    }
    return valid;
}

// a worm operator and a hybridising operator of the same flavour and type exchange their roles: one row (creator) or
// column (annihilator) of the hybridisation matrix changes; Sherman-Morrison on the slot-ordered M, no reordering.
//
// Sectors 6 - 9: if the worm operator belongs to an equal-time pair, its partner moves along to the hybridising operator's
// time (CTQMC.F90:6068-6127, 6183-6249): the pair then sits at (tau_h, tau_h + offset) in the order it had, the old worm operator
// stays where it is (as a hybridising operator from now on), and the acceptance ratio gets the local-trace ratio of moving
// ONE operator, the partner (flavour g, from t to t'), with everything else in place:
//   one-state blocks: valid iff no other operator of flavour g lies between t and t' (the former hybridising operator at its
//   new time included), log ratio = -dn int_(t,t') (eps_g + sum_{g'} U_gg' n_g')  with dn = -1 (+1) for a creator moved to a
//   later (earlier) time and the reverse for an annihilator, n_g' from all other operators (level_integral<NW>);
//   larger blocks: the partner's list entry is moved, block sequence and trace are evaluated like for a worm insertion.
// A move that fails the quantum-number check consumes no random number (goto 99 there), a vanishing trace does.
template <int NW, int MAT, bool IE = false>
__device__ W2DYN_INLINE void step_worm_replace(Ctx &x)
{
    const DevConst &C = x.C;
    x.count(C_TRYWREP);
    int r = 0, partner = -1, pos_partner = 0, partner2 = -1, pos_partner2 = 0, pos_new = 0;
    if constexpr (!IE) {
            r = x.randint(1, NW) - 1;
        // container indices of the partners that move along (-1: none) and their places in the new equal-time object (1: at tau_h, 2: one
        // offset above, 3: two); the former hybridising operator takes the place that is left (qrepind / qtimeind, CTQMC.F90:6068-6127, 6164-6178)
        if constexpr (NW == 4) {
            const int sec = C.isector;
            if (sec == 3) {
                if (r == 0) { partner = 2; pos_partner = 2; partner2 = 1; pos_partner2 = 3; pos_new = 1; }
                else if (r == 1) { partner = 0; pos_partner = 1; partner2 = 2; pos_partner2 = 2; pos_new = 3; }
                else if (r == 2) { partner = 0; pos_partner = 1; partner2 = 1; pos_partner2 = 3; pos_new = 2; }
            }
            else if (sec == 6) { partner = r ^ 1; pos_partner = (r & 1) ? 2 : 1; }
            else if (sec == 7) { partner = r ^ 2; pos_partner = r < 2 ? 1 : 2; }
            else if (sec == 8) { if (r >= 2) { partner = r ^ 1; pos_partner = r == 2 ? 1 : 2; } }
            else if (sec == 9) { if (r & 1) { partner = r ^ 2; pos_partner = r == 1 ? 1 : 2; } }
            if (partner >= 0 && partner2 < 0) pos_new = 3 - pos_partner;
        }
    } else {
        // which worm operator becomes a hybridising one (CTQMC.F90:5834-5946): uniform over the container in sectors 2 - 4, 6 - 9, 14, 15; only
        // the operators outside the equal-time object in sectors 5 and 16; sectors 10 - 12: a three-operator object that holds a density
        // (operators 2, 3 or 4, 5 ... of equal flavour) moves the density; no replacement at all in sector 13
        const int sec = C.isector;
        if constexpr (NW == 8) { if (sec == 13) return; }
        if (NW >= 4 && (sec == 5 || sec == 16)) r = x.randint(4, NW) - 1;
        else if (NW == 6 && sec == 10) {
            const bool d1 = x.wfl[1] == x.wfl[2], d2 = x.wfl[NW >= 6 ? 3 : 0] == x.wfl[NW >= 6 ? 4 : 0];
            if (d1) r = d2 ? (x.randint(1, 2) - 1) * 5 : 0; else r = d2 ? 5 : x.randint(1, 6) - 1;
        } else if (NW == 8 && sec == 12) {
            const bool d1 = x.wfl[1] == x.wfl[2], d2 = x.wfl[NW >= 8 ? 5 : 0] == x.wfl[NW >= 8 ? 4 : 0];
            if (d1) r = d2 ? (x.randint(1, 2) - 1) * 3 : 0; else r = d2 ? 3 : x.randint(1, 6) - 1;
        } else if (NW == 12) {
            const bool dd[4] = {x.wfl[1] == x.wfl[2], x.wfl[NW == 12 ? 3 : 0] == x.wfl[NW == 12 ? 4 : 0], x.wfl[NW == 12 ? 7 : 0] == x.wfl[NW == 12 ? 8 : 0],
                                x.wfl[NW == 12 ? 9 : 0] == x.wfl[NW == 12 ? 10 : 0]};
            const int nden = (int)dd[0] + (int)dd[1] + (int)dd[2] + (int)dd[3];
            if (nden > 0) {
                int pick = x.randint(1, nden); r = 0;
                const int stay[4] = {0, 5, 6, 11};               // the operator that stays when object q gives its density away
    #pragma unroll
                for (int q = 3; q >= 0; --q) { int before = 0; for (int p = 0; p < q; ++p) before += (int)dd[p]; if (dd[q] && before == pick - 1) r = stay[q]; }
            } else r = x.randint(1, 12) - 1;
        } else r = x.randint(1, NW) - 1;
        // container indices of the partners that move along (-1: none) and their places in the new equal-time object (1: at tau_h, 2: one
        // offset above, 3: two); the former hybridising operator takes the place that is left (qrepind / qtimeind, CTQMC.F90:5997-6178)
        if constexpr (NW >= 4) {
            // three-operator objects: row = the operator that stays (six per pair of objects), {partner, place, partner, place, place of the new one}
            bool three = false; int row = 0, shift = 0; bool nqq = false;
            if (sec == 3) { three = r < 3; row = r; }
            else if (sec == 10 || sec == 11) { three = true; row = r % 6; shift = 6 * (r / 6); }
            else if (sec == 12) { three = true; row = r; nqq = r >= 3; }
            if (three) {
                const int q3[6][5] = {{2, 2, 1, 3, 1}, {0, 1, 2, 2, 3}, {0, 1, 1, 3, 2}, {4, 1, 5, 3, 2}, {3, 2, 5, 3, 1}, {4, 1, 3, 2, 3}};
                const int q3n[3][5] = {{4, 1, 5, 2, 3}, {5, 2, 3, 3, 1}, {4, 1, 3, 3, 2}};                      // second object of sector 12
    #pragma unroll
                for (int q = 0; q < 6; ++q) if (q == row) {
                    const int *e = (nqq && q >= 3) ? q3n[q - 3] : q3[q];
                    partner = e[0] + shift; pos_partner = e[1]; partner2 = e[2] + shift; pos_partner2 = e[3]; pos_new = e[4];
                }
            }
            else if (sec == 6) { partner = r ^ 1; pos_partner = (r & 1) ? 2 : 1; }
            else if (sec == 7) { partner = r ^ 2; pos_partner = r < 2 ? 1 : 2; }
            else if (sec == 8) { if (r >= 2) { partner = r ^ 1; pos_partner = r == 2 ? 1 : 2; } }
            else if (sec == 9) { if (r & 1) { partner = r ^ 2; pos_partner = r == 1 ? 1 : 2; } }
            else if (sec == 14) { partner = r ^ 1; pos_partner = (r & 1) ? 1 : 2; }                             // 6151-6162
            else if (sec == 15) { partner = r ^ 2; pos_partner = (r == 0 || r == 3) ? 2 : 1; }                  // 6133-6144
            if (partner >= 0 && partner2 < 0) pos_new = 3 - pos_partner;
        }
    }
    double tw = 0.0; int f = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) if (i == r) { tw = x.wtau[i]; f = x.wfl[i]; }
    const bool creator = (r & 1) != 0;
    const int k = x.order(f), kmax = C.kmax;
    if (k == 0) return;
    const int pos = x.randint(1, k);
    double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
    const double *own = creator ? tcp : tap;
    // slot of the pos-th operator of that flavour and type in time order
    int slot = -1;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane;
        int rank = 0;
        if (e < k) { const double te = own[e]; for (int j = 0; j < k; ++j) rank += own[j] < te ? 1 : 0; }
        const unsigned long long m = __ballot(e < k && rank == pos - 1);
        if (m) slot = base + __ffsll((long long)m) - 1;
    }
    slot = uni(slot);
    const double th = uni(own[slot]);
    int ncnt = 0;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane;
        ncnt += count_pred(e < k && own[e] > fmin(th, tw) && own[e] < fmax(th, tw));
    }
    const double *ft = C.ftab + (size_t)f * C.nftau;
    // creator exchanged: u over annihilators, X = M u (per creator), Y = row `slot` of M; detRat = 1 + X[slot]
    // annihilator exchanged: v over creators, X = column `slot` of M, Y = M^T v; detRat = 1 + v . X
    for (int e = x.lane; e < k; e += 64) {
        if (creator) x.Q[e] = -hybr(ft, C.nftau, C.beta, tap[e] - th) + hybr(ft, C.nftau, C.beta, tap[e] - tw);
        else x.Q[e] = -hybr(ft, C.nftau, C.beta, th - tcp[e]) + hybr(ft, C.nftau, C.beta, tw - tcp[e]);
    }
    wave_sync();
    double part = 0.0;
    for (int e = x.lane; e < k; e += 64) {
        double xs = 0.0, ys = 0.0;
        if (creator) {
            for (int a = 0; a < k; ++a) xs += Mf[(size_t)e * kmax + a] * x.Q[a];
            ys = Mf[(size_t)slot * kmax + e];
        } else {
            xs = Mf[(size_t)e * kmax + slot];
            for (int c = 0; c < k; ++c) ys += x.Q[c] * Mf[(size_t)c * kmax + e];
            part += x.Q[e] * xs;
        }
        x.MQ[e] = xs; x.RM[e] = ys;
    }
    wave_sync();
    const double detrat = creator ? 1.0 + uni(x.MQ[slot]) : 1.0 + wave_sum(part);
    double prob = fabs(detrat);
    double tp = 0.0, tpn = 0.0, tp2 = 0.0, tpn2 = 0.0, thn = th, tl = 0.0, ts = 1.0; int fp = 0, fp2 = 0;
    bool slot_moved = false;
    if constexpr (NW >= 4) {
        if (partner >= 0) {
            const double off = C.beta * 1e-14;
            tpn = th + (double)(pos_partner - 1) * off; tpn2 = th + (double)(pos_partner2 - 1) * off; thn = th + (double)(pos_new - 1) * off;
#pragma unroll
            for (int i = 0; i < NW; ++i) { if (i == partner) { tp = x.wtau[i]; fp = x.wfl[i]; } if (i == partner2) { tp2 = x.wtau[i]; fp2 = x.wfl[i]; } }
            const int pca = (partner & 1) ? 1 : 2, pca2 = (partner2 & 1) ? 1 : 2;
            if constexpr (MAT == 0) {
                if (partner2 < 0) {
                    const double lo = fmin(tp, tpn), hi = fmax(tp, tpn);
                    bool ok = !worm_any_between(x, fp, lo, hi);
                    if (fp == f && thn > lo && thn < hi) ok = false;
#pragma unroll
                    for (int i = 0; i < NW; ++i) if (i != partner && x.wfl[i] == fp && x.wtau[i] > lo && x.wtau[i] < hi) ok = false;
                    if (!ok) return;
                    const double dn = ((pca == 1) == (tpn > tp)) ? -1.0 : 1.0;
                    prob = fabs(detrat * exp(-dn * level_integral<NW>(x, fp, lo, hi)));
                } else {
                    // two partners (three-operator object): with the former hybridising operator at its place in the new object
                    double dl;
                    if (thn != th) { wave_sync(); if (x.lane == 0) { if (creator) tcp[slot] = thn; else tap[slot] = thn; } wave_sync(); slot_moved = true; }
                    const bool ok = worm_shift_delta<NW>(x, partner, partner2, tpn, tpn2, dl);
                    if (!ok) { if (slot_moved) { if (x.lane == 0) { if (creator) tcp[slot] = th; else tap[slot] = th; } wave_sync(); } return; }
                    prob = fabs(detrat * exp(dl));
                }
            } else {
                mat_remove(x, mat_lower(x, tp));
                wave_sync();
                if (partner2 >= 0) { mat_remove(x, mat_lower(x, tp2)); wave_sync(); }
                if (thn != th) { const int ih = mat_lower(x, th); if (x.lane == 0) x.ot[ih] = thn; wave_sync(); }
                // the new object in time order, so that every insertion index is found with the others in place
#pragma unroll
                for (int place = 1; place <= 3; ++place) {
                    if (pos_partner == place) mat_insert(x, mat_lower(x, tpn), tpn, mat_code(fp, pca));
                    if (partner2 >= 0 && pos_partner2 == place) mat_insert(x, mat_lower(x, tpn2), tpn2, mat_code(fp2, pca2));
                }
                if (!mat_walk_into(x, x.outer, x.tsst)) ts = 0.0;
                else {
                    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl, ts);
                    prob = ts != 0.0 ? fabs(detrat * exp(tl - x.tr_log)) : 0.0;
                    ts = ts != 0.0 ? ts : 2.0;                  // 2: the sequence is allowed, the trace vanishes (a random number is drawn)
                }
            }
        }
    }
    bool accept = false;
    if (ts != 0.0) { const double rnd = x.draw(); accept = rnd < prob; }
    if (!accept) {
        if constexpr (NW >= 4 && MAT != 0) {
            if (partner >= 0) {                                // back to the old list
                mat_remove(x, mat_lower(x, tpn));
                wave_sync();
                if (partner2 >= 0) { mat_remove(x, mat_lower(x, tpn2)); wave_sync(); }
                if (thn != th) { const int ih = mat_lower(x, thn); if (x.lane == 0) x.ot[ih] = th; wave_sync(); }
                mat_insert(x, mat_lower(x, tp), tp, mat_code(fp, (partner & 1) ? 1 : 2));
                if (partner2 >= 0) mat_insert(x, mat_lower(x, tp2), tp2, mat_code(fp2, (partner2 & 1) ? 1 : 2));
            }
        }
        if (slot_moved) { if (x.lane == 0) { if (creator) tcp[slot] = th; else tap[slot] = th; } wave_sync(); }
        return;
    }
    const int kk = k * k;
    for (int e = x.lane; e < kk; e += 64) {
        const int c = e / k, a = e - c * k;
        Mf[(size_t)c * kmax + a] -= x.MQ[c] * x.RM[a] / detrat;
    }
    if (x.lane == 0) { if (creator) tcp[slot] = tw; else tap[slot] = tw; }
#pragma unroll
    for (int i = 0; i < NW; ++i) { if (i == r) x.wtau[i] = thn; if (NW >= 4 && i == partner) x.wtau[i] = tpn; if (NW >= 4 && i == partner2) x.wtau[i] = tpn2; }
    if constexpr (NW >= 4 && MAT != 0) {
        if (partner >= 0) { x.tr_log = tl; x.tr_sign = ts; mat_commit_sst(x); }
    }
    wave_sync();
    x.det_log += log(fabs(detrat));
    x.det_sign *= (detrat < 0.0 ? -1.0 : 1.0) * ((ncnt & 1) ? -1.0 : 1.0);
    x.npairs = -1; x.wvalid = 0;
    x.count(C_ACCWREP);
}

// Matrix trace path: the worm operators are entries of the time-ordered operator list (they take part in the block
// sequence and in the trace like any operator, but not in M).  Insertion: put the NW operators into the list, walk the
// block sequence from the outer block (this IS the quantum-number check, check_sst_sequence), evaluate the trace.
template <int NW, int MAT, bool IE = false>
__device__ W2DYN_INLINE void step_worm_add_mat(Ctx &x)
{
    const DevConst &C = x.C;
    x.count(C_TRYWADD);
    double t[NWMAX] = {}; int fl[NWMAX] = {};
    worm_draw_flavours<NW, IE>(x, fl);
    worm_draw_times<NW, IE>(x, t);
    if (x.nops + NW > C.wcap) { x.err |= ERR_WCAP; x.count(C_QNWADD); x.sector = 1; return; }
#pragma unroll
    for (int i = 0; i < NW; ++i) mat_insert(x, mat_lower(x, t[i]), t[i], mat_code(fl[i], (i & 1) ? 1 : 2));
    auto take_out = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) mat_remove(x, mat_lower(x, t[i]));
        wave_sync();
    };
    const bool valid = mat_walk_into(x, x.outer, x.tsst);
    const double prefact = worm_prefact<NW, IE>(C, fl);
    if (prefact == 0.0 || !valid) { take_out(); x.count(C_QNWADD); x.sector = 1; return; }
    double tl, ts;
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl, ts);
    const double rnd = x.draw();
    if (ts != 0.0 && rnd < fabs(prefact * exp(tl - x.tr_log))) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { x.wtau[i] = t[i]; x.wfl[i] = fl[i]; }
        x.tr_log = tl; x.tr_sign = ts; mat_commit_sst(x);
        x.sector = C.isector;
        x.count(C_ACCWADD);
    } else { take_out(); x.sector = 1; }
}

template <int NW, int MAT, bool IE = false>
__device__ W2DYN_INLINE void step_worm_rem_mat(Ctx &x)
{
    const DevConst &C = x.C;
    x.count(C_TRYWREM);
#pragma unroll
    for (int i = 0; i < NW; ++i) mat_remove(x, mat_lower(x, x.wtau[i]));
    wave_sync();
    auto put_back = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) mat_insert(x, mat_lower(x, x.wtau[i]), x.wtau[i], mat_code(x.wfl[i], (i & 1) ? 1 : 2));
    };
    const bool valid = mat_walk_into(x, x.outer, x.tsst);
    if (!valid) { put_back(); x.count(C_QNWREM); return; }
    double tl, ts;
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl, ts);
    const double rnd = x.draw();
    if (ts != 0.0 && rnd < fabs(exp(tl - x.tr_log) / worm_prefact<NW, IE>(C, x.wfl))) {
        x.tr_log = tl; x.tr_sign = ts; mat_commit_sst(x);
        x.sector = 1; x.count(C_ACCWREM);
    } else put_back();
}

// sweep kernel with worm sampling: Z space CTQMC.F90:8431-8488, worm space 8489-8528.  MAT = 0: scalar trace path,
// otherwise the matrix trace path instantiation (2, 3, 4, 9 as in k_sweep)
// XM: with the two-pair moves and the pair moves that change the outer block (matrix trace path only; CTQMC.F90:8441-8449, 8497-8516)
template <int NW, int MAT, bool IE = false, bool XM = false>
__global__ __launch_bounds__(256, W2DYN_WPE_WORM) void k_sweep_worm(DevConst C, long long nsteps_launch, int resume)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    for (bool first = true;; first = false) {          // as in k_sweep: one pass (scalar path) or chains from the launch's queue (matrix path)
    if constexpr (MAT == 0) { if (!first) break; ctx_setup(x, smem); } else ctx_setup(x, smem, next_chain(C));
    if (x.w >= C.nwalkers) break;
    const long long nsteps = launch_steps(C, x.w, x.lane, nsteps_launch, resume);
    if (nsteps == 0) continue;
    ctx_load(x);
    if constexpr (XM) flag_full(x, 1); else flag_full(x);
    if constexpr (MAT != 0) mat_trace_current<MAT>(x);
    const double pg = C.p_glob, ps = C.p_shift, pw = C.p_wins, pr = C.p_wrep;
    unsigned int phase = (unsigned int)((x.cnt[C_TRYADD] + x.cnt[C_TRYREM] + x.cnt[C_W_TRYADD] + x.cnt[C_W_TRYREM]) % (unsigned long long)C.nslide);
    phase = (unsigned int)uni((int)phase);
    // XM: pair insertion / removal of a step is one pair (with the outer block left alone or not) or two pairs at once; the kernels without
    // these moves keep their four call sites as they were
    constexpr bool XMK = XM && MAT != 0;
    // (always inlined: left out of line, the four call sites made a second call level above the sequential evaluator's - the constellation
    // that hung the GPU in round 2 - and this kernel hung the same way on blocks of nine states)
    auto pair_step = [&](bool add) __attribute__((always_inline)) {
        if constexpr (XMK) {
            if (x.draw() >= C.p_4op) {
                const bool outer = x.draw() < C.p_outer;
                if (add) step_add<MAT, NW, false, true>(x, outer); else step_rem<MAT, NW, false, true>(x, outer);
                phase += 1;
            } else if (add) step_add4<MAT>(x); else step_rem4<MAT>(x);
        }
    };
    long long it = 0;
    for (; it < nsteps && !(x.err & ERR_KMAX); ++it) {
#ifdef W2DYN_STAMPS_WORM      // diagnostic build: shader-clock cycles per move type
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); int kind_;
#define WORM_KIND(c) kind_ = (c)
#else
#define WORM_KIND(c)
#endif
        if (x.sector == 1) {
            x.count(C_SAMPLING);
            const double r = x.draw();
            if (r < (1.0 - ps - pg - pw) / 2.0) { if constexpr (XMK) pair_step(true); else { x.skip(2); step_add<MAT, NW>(x); phase += 1; } WORM_KIND(C_CYC_ADD); }
            else if (r <= 1.0 - ps - pg - pw) { if constexpr (XMK) pair_step(false); else { x.skip(2); step_rem<MAT, NW>(x); phase += 1; } WORM_KIND(C_CYC_REM); }
            else if (r <= 1.0 - ps - pg) { if constexpr (MAT != 0) step_worm_add_mat<NW, MAT, IE>(x); else step_worm_add<NW, IE>(x); WORM_KIND(C_CYC_GLOB); }
            else if (r <= 1.0 - ps) { if constexpr (MAT != 0) step_glob_mat<MAT>(x); else step_glob(x); WORM_KIND(C_CYC_SHIFT); }
            else { step_shift<MAT>(x); WORM_KIND(C_CYC_SHIFT); }
        } else {
            x.count(C_W_SAMPLING);
            const double r = x.draw();
            if (r < (1.0 - pr - pw) / 2.0) { if constexpr (XMK) pair_step(true); else { x.skip(2); step_add<MAT, NW>(x); phase += 1; } WORM_KIND(C_DBG5); }
            else if (r < 1.0 - pr - pw) { if constexpr (XMK) pair_step(false); else { x.skip(2); step_rem<MAT, NW>(x); phase += 1; } WORM_KIND(C_DBG5); }
            else if (r < 1.0 - pw) { step_worm_replace<NW, MAT, IE>(x); WORM_KIND(C_DBG6); }
            else { if constexpr (MAT != 0) step_worm_rem_mat<NW, MAT, IE>(x); else step_worm_rem<NW, IE>(x); WORM_KIND(C_DBG7); }
        }
#ifdef W2DYN_STAMPS_WORM
        if (x.lane == 0) x.cnt[kind_] += __builtin_amdgcn_s_memtime() - t0_;
#endif
        if (phase == (unsigned int)C.nslide) phase = 0;
        if (phase == 0) {
            if (C.num_windows > 1) {
                x.wpos += 1; if (x.wpos == C.num_windows) x.wpos = 0;
                x.window_bounds(); x.npairs = -1; x.wvalid = 0;
            } else x.wpos = 0;
        }
    }
    ctx_store(x, nsteps - it);
    }
}

#ifdef W2DYN_TU_MEAS
// G4iw worm estimator: every chain that is in worm space adds cfgsign/eta * exp(i[nu1 x1 + nu2 x2 + omega x3]) to its
// bin, ph: x = (t1-t2, t3-t4, t1-t4 | t2-t3), pp: x = (t1-t3, t2-t4, t3-t2).  One workgroup per bin loops over the
// bin's chains; the three one-dimensional phase tables of a chain go through LDS, accumulators stay in registers
// (no HBM traffic per sample), one read-modify-write of the bin's array per launch.
constexpr int G4W_PER_THREAD = 8;
__global__ __launch_bounds__(256) void k_meas_g4w(DevConst C, int pp, int nbw)
{
    extern __shared__ double2 ph[];              // [2nwf] [2nwf] [2nwb+1]
    const int nw = 2 * C.nwf, nb = 2 * C.nwb + 1, n = nw * nw * nb, bin = blockIdx.x, tid = threadIdx.x;
    double *acc = (pp ? C.acc_g4wpp : C.acc_g4w) + (size_t)bin * n * 2;
    const double pib = M_PI / C.beta;
    for (int e0 = 0; e0 < n; e0 += 256 * G4W_PER_THREAD) {
        double re[G4W_PER_THREAD], im[G4W_PER_THREAD];
#pragma unroll
        for (int q = 0; q < G4W_PER_THREAD; ++q) { re[q] = 0.0; im[q] = 0.0; }
        for (int cb = bin; cb < C.nbins; cb += nbw)            // chain bins folded onto this accumulator bin
        for (int w = cb * C.walkers_per_bin; w < min((cb + 1) * C.walkers_per_bin, C.nwalkers); ++w) {
            const WalkerScalars &s = C.ws[w];
            if (s.sector != 4) continue;                     // uniform over the workgroup
            const double t1 = s.wtau[0], t2 = s.wtau[1], t3 = s.wtau[2], t4 = s.wtau[3];
            double x1, x2, x3;
            if (!pp) { x1 = t1 - t2; x2 = t3 - t4; x3 = C.wormphconv ? t2 - t3 : t1 - t4; }
            else { x1 = t1 - t3; x2 = t2 - t4; x3 = t3 - t2; }
            const double val = C.cfgsign[w] / C.worm_eta;
            __syncthreads();
            for (int i = tid; i < 2 * nw + nb; i += 256) {
                double a;
                if (i < nw) a = pib * (double)(2 * (i - C.nwf) + 1) * x1;
                else if (i < 2 * nw) a = pib * (double)(2 * (i - nw - C.nwf) + 1) * x2;
                else a = pib * (double)(2 * (i - 2 * nw - C.nwb)) * x3;
                double sn, cs; sincos(a, &sn, &cs);
                ph[i] = make_double2(cs, sn);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < G4W_PER_THREAD; ++q) {
                const int e = e0 + q * 256 + tid;
                if (e < n) {
                    const int i3 = e % nb, i2 = (e / nb) % nw, i1 = e / (nb * nw);
                    const double2 a = ph[i1], b = ph[nw + i2], c = ph[2 * nw + i3];
                    const double pr = a.x * b.x - a.y * b.y, pi = a.x * b.y + a.y * b.x;
                    re[q] += val * (pr * c.x - pi * c.y); im[q] += val * (pr * c.y + pi * c.x);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < G4W_PER_THREAD; ++q) {
            const int e = e0 + q * 256 + tid;
            if (e < n) { acc[2 * e] += re[q]; acc[2 * e + 1] += im[q]; }
        }
    }
}

// One-particle worm estimators (sector 2), measure_giw_worm / measure_gtau_worm (CTQMC.F90:1061-1120): every chain in worm
// space adds cfgsign/eta * exp(i w_n (t1 - t2)), n = -Niw..Niw-1, to giw_worm of its bin and +-cfgsign/eta to the bin of
// t1 - t2 (+ beta with sign -1) of gtau_worm.  One workgroup per accumulator bin loops over the bin's chains; thread t
// owns the frequencies t, t+256, ... and steps from one to the next by a rotation with exp(i 512 pi (t1-t2)/beta) (two
// sincos per chain and thread); accumulators in registers, the tau histogram in LDS, one read-modify-write per launch.
constexpr int GW_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_meas_gw(DevConst C)
{
    extern __shared__ double gth[];              // [ntau]
    const int bin = blockIdx.x, tid = threadIdx.x, nf2 = C.niw2, niw = C.niw2 / 2;
    const double pib = M_PI / C.beta;
    if (C.acc_gtauw) { for (int i = tid; i < C.ntau; i += 256) gth[i] = 0.0; }
    __syncthreads();
    const int w0 = bin * C.walkers_per_bin, w1 = min((bin + 1) * C.walkers_per_bin, C.nwalkers);
    for (int e0 = 0; e0 < nf2 || e0 == 0; e0 += 256 * GW_PER_THREAD) {
        double re[GW_PER_THREAD], im[GW_PER_THREAD];
#pragma unroll
        for (int q = 0; q < GW_PER_THREAD; ++q) { re[q] = 0.0; im[q] = 0.0; }
        for (int w = w0; w < w1; ++w) {
            const WalkerScalars &s = C.ws[w];
            if (s.sector != C.isector) continue;             // uniform over the workgroup
            // sector 2: t1 - t2; sector 3 (GSigmaiw_worm, measure_ie_sigma_worm, CTQMC.F90:1123-1164): t1 - t4, times the sign of the U-matrix factor
            double dt = s.wtau[0] - s.wtau[1], val = C.cfgsign[w] / C.worm_eta;
            if (C.isector == 3) {
                const int fl[4] = {s.wflav[0], s.wflav[1], s.wflav[2], s.wflav[3]};
                dt = s.wtau[0] - s.wtau[3];
                if (gsigma_ufac(C, fl) < 0.0) val = -val;
            }
            if (C.acc_giww) {
                double sn, cs, sr, cr;
                sincos(pib * (double)(2 * (e0 + tid - niw) + 1) * dt, &sn, &cs);
                sincos(pib * 512.0 * dt, &sr, &cr);
#pragma unroll
                for (int q = 0; q < GW_PER_THREAD; ++q) {
                    re[q] += val * cs; im[q] += val * sn;
                    const double c2 = cs * cr - sn * sr; sn = cs * sr + sn * cr; cs = c2;
                }
            }
            if (C.acc_gtauw && e0 == 0 && tid == 0) {
                double tau = dt, sg = 1.0;
                if (tau < 0.0) { tau += C.beta; sg = -1.0; }
                int it = (int)((double)C.ntau * tau / C.beta) + 1;
                if (it > C.ntau) it = C.ntau;
                gth[it - 1] += sg * val;
            }
        }
        if (C.acc_giww) {
            double *acc = C.acc_giww + (size_t)bin * nf2 * 2;
#pragma unroll
            for (int q = 0; q < GW_PER_THREAD; ++q) {
                const int e = e0 + q * 256 + tid;
                if (e < nf2) { acc[2 * e] += re[q]; acc[2 * e + 1] += im[q]; }
            }
        }
    }
    __syncthreads();
    if (C.acc_gtauw) { double *acc = C.acc_gtauw + (size_t)bin * C.ntau; for (int i = tid; i < C.ntau; i += 256) acc[i] += gth[i]; }
}

// Estimators of the equal-time sectors (measure_p2iw_worm, p2tau, p2iwpp, p2taupp, p3iw, p3iwpp, CTQMC.F90:1297-1507, with the
// exact transform, Nfft_worm.F90:94-137): every chain in worm space adds cfgsign/eta times
//   sectors 6 / 7: exp(i w_j dt), w_j = 2 pi j / beta, j = -N2iwb..N2iwb, and one count in the bin of dt (+ beta if negative, no
//                  sign change), dt = t1 - t3 (P2) or t1 - t2 (P2pp);
//   sectors 8 / 9: exp(i [nu_i x1 + w_j x2]), nu_i = (2 i + 1) pi / beta, i = -N3iwf..N3iwf-1, j = -N3iwb..N3iwb,
//                  (x1, x2) = (t1 - t2, t2 - t3) (P3) or (t1 - t3, t3 - t2) (P3pp).
// One workgroup per accumulator bin (sample bins folded onto nbe of them), the bin's chains looped, thread = element.
__global__ __launch_bounds__(256) void k_meas_weq(DevConst C, int nbe, double *acc_iw, double *acc_tau, double *acc_p3, int n2b, int n3f, int n3b)
{
    const int bin = blockIdx.x, tid = threadIdx.x, sec = C.isector;
    const double pib = M_PI / C.beta;
    for (int cb = bin; cb < C.nbins; cb += nbe)
    for (int w = cb * C.walkers_per_bin; w < min((cb + 1) * C.walkers_per_bin, C.nwalkers); ++w) {
        const WalkerScalars &s = C.ws[w];
        if (s.sector < 6) continue;                          // uniform over the workgroup
        const double t1 = s.wtau[0], t2 = s.wtau[1], t3 = s.wtau[2];
        const double val = C.cfgsign[w] / C.worm_eta;
        if (sec <= 7) {
            const double dt = sec == 6 ? t1 - t3 : t1 - t2;
            if (acc_iw) {
                double *a = acc_iw + (size_t)bin * (2 * n2b + 1) * 2;
                for (int e = tid; e < 2 * n2b + 1; e += 256) {
                    double sn, cs; sincos(pib * (double)(2 * (e - n2b)) * dt, &sn, &cs);
                    a[2 * e] += val * cs; a[2 * e + 1] += val * sn;
                }
            }
            if (acc_tau && tid == 0) {
                double tau = dt; if (tau < 0.0) tau += C.beta;
                int it = (int)((double)C.ntau * tau / C.beta) + 1;
                if (it > C.ntau) it = C.ntau;
                acc_tau[(size_t)bin * C.ntau + it - 1] += val;
            }
        } else if (acc_p3) {
            const double x1 = sec == 8 ? t1 - t2 : t1 - t3, x2 = sec == 8 ? t2 - t3 : t3 - t2;
            const int nb = 2 * n3b + 1, n = 2 * n3f * nb;
            double *a = acc_p3 + (size_t)bin * n * 2;
            for (int e = tid; e < n; e += 256) {
                const int i = e / nb - n3f, j = e - (e / nb) * nb - n3b;
                double sn, cs; sincos(pib * ((double)(2 * i + 1) * x1 + (double)(2 * j) * x2), &sn, &cs);
                a[2 * e] += val * cs; a[2 * e + 1] += val * sn;
            }
        }
    }
}
// Estimators of the worm sectors 5 and 10 - 16 (measure_ie_chi_worm, measure_qq_worm / qqtau, measure_qqqq_worm, measure_nqqdag_worm,
// measure_qqdd_worm, measure_ucaca_worm / ucacatau, measure_uccaa_worm / uccaatau, measure_quddag_worm; CTQMC.F90:1166-1294, 1708-2128, with
// the exact transform, Nfft_worm.F90:94-137): every chain in worm space adds cfgsign/eta times the sign of the proposal's U-matrix
// factor times exp(i [m1 x1 + m2 x2 + m3 x3] pi / beta) to a complex array of up to three frequency axes (fermionic m = 2 n + 1 or bosonic
// m = 2 n), and one count to the tau bin of a time difference:
//   5  h4iw (F, F, B):  t1 - t4, t5 - t6, t1 - t6 (WormPHConvention: t4 - t5)       10 qq (F): -(t1 - t4);  qqtau: t2 - t6, antiperiodic
//   11 qqqq (F, F, B):  t2 - t6, t8 - t12, t2 - t12 (t6 - t8)                        12 nqqdag (F, B): t2 - t4, t4 - t8
//   13 qqdd (F, F):     t2 - t6, t4 - t6                                             14 ucaca (B), ucacatau: t1 - t3, periodic
//   15 uccaa (B), uccaatau: t3 - t2, periodic                                        16 quddag (F): t1 - t4
// One workgroup per accumulator bin (sample bins folded onto nbe of them), the bin's chains looped, thread = element.
__global__ __launch_bounds__(256) void k_meas_wie(DevConst C, int nbe, double *acc_iw, double *acc_tau, int n1, int n2, int n3)
{
    const int bin = blockIdx.x, tid = threadIdx.x, sec = C.isector;
    const double pib = M_PI / C.beta;
    const bool f1 = sec != 14 && sec != 15, f2 = true, f3 = false;        // fermionic axes (the second axis of sector 12 is bosonic, below)
    const bool b2 = sec == 12;
    const int n = n1 * n2 * n3;
    for (int cb = bin; cb < C.nbins; cb += nbe)
    for (int w = cb * C.walkers_per_bin; w < min((cb + 1) * C.walkers_per_bin, C.nwalkers); ++w) {
        const WalkerScalars &s = C.ws[w];
        if (s.sector != sec) continue;                       // uniform over the workgroup
        double t[NW_STORE]; int g[NW_STORE];
#pragma unroll
        for (int i = 0; i < NW_STORE; ++i) { t[i] = s.wtau[i]; g[i] = s.wflav[i]; }
        double val = C.cfgsign[w] / C.worm_eta;
        if (ie_ufac(C, sec, g, true) < 0.0) val = -val;
        double x1 = 0.0, x2 = 0.0, x3 = 0.0, dtau = 0.0; bool anti = false;
        switch (sec) {
        case 5: x1 = t[0] - t[3]; x2 = t[4] - t[5]; x3 = C.wormphconv ? t[3] - t[4] : t[0] - t[5]; break;
        case 10: x1 = t[0] - t[3]; dtau = t[1] - t[5]; anti = true; break;
        case 11: x1 = t[1] - t[5]; x2 = t[7] - t[11]; x3 = C.wormphconv ? t[5] - t[7] : t[1] - t[11]; break;
        case 12: x1 = t[1] - t[3]; x2 = t[3] - t[7]; break;
        case 13: x1 = t[1] - t[5]; x2 = t[3] - t[5]; break;
        case 14: x1 = t[0] - t[2]; dtau = x1; break;
        case 15: x1 = t[2] - t[1]; dtau = x1; break;
        default: x1 = t[0] - t[3]; break;
        }
        if (acc_iw) {
            const double viw = sec == 10 ? -val : val;       // measure_qq_worm carries a minus sign (CTQMC.F90:1230)
            double *a = acc_iw + (size_t)bin * n * 2;
            for (int e = tid; e < n; e += 256) {
                const int i3 = e % n3, i2 = (e / n3) % n2, i1 = e / (n3 * n2);
                const double m1 = f1 ? (double)(2 * (i1 - n1 / 2) + 1) : (double)(2 * (i1 - (n1 - 1) / 2));
                const double m2 = n2 == 1 ? 0.0 : (b2 ? (double)(2 * (i2 - (n2 - 1) / 2)) : (double)(2 * (i2 - n2 / 2) + 1));
                const double m3 = n3 == 1 ? 0.0 : (double)(2 * (i3 - (n3 - 1) / 2));
                double sn, cs; sincos(pib * (m1 * x1 + m2 * x2 + m3 * x3), &sn, &cs);
                a[2 * e] += viw * cs; a[2 * e + 1] += viw * sn;
            }
        }
        if (acc_tau && tid == 0) {
            double tau = dtau, v = val;
            if (tau < 0.0) { tau += C.beta; if (anti) v = -v; }
            int it = (int)((double)C.ntau * tau / C.beta) + 1;
            if (it > C.ntau) it = C.ntau;
            acc_tau[(size_t)bin * C.ntau + it - 1] += v;
        }
    }
    (void)f2; (void)f3;
}
#endif  // W2DYN_TU_MEAS
