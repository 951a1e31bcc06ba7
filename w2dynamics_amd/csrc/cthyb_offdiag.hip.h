// Off-diagonal hybridisation (offdiag = 1, CTQMC.F90:461-467): the bath couples unequal flavours, so a configuration has ONE
// hybridisation matrix over all flavours (MInv_full, Trace.F90:2690-2770; rows = creators, columns = annihilators) and a pair
// move draws the flavours of its two operators independently.  Included by cthyb_kernels.hip.h inside namespace w2dyn.
//
// Storage.  The chain's arrays of flavour 0 hold ALL hybridising operators (K[w][0] = number of pairs, the other orders stay 0):
// creator / annihilator times in slot order, M[w][0] of order kt with row = creator slot, column = annihilator slot, and the
// flavour of every slot in the tag arrays FC / FA.  In LDS the tags take the slices of flavour 1's times, which nothing else
// uses in this mode.  Slot order makes the rank-one updates those of the diagonal case; the reference's ordering of rows and
// columns (by flavour, then time) matters for signs only and is obtained by counting.
//
// Always the matrix trace path: the time-ordered operator list carries the true flavours (od_build_list), quantum numbers are
// checked by walking the block sequence, the trial trace is evaluated without the window cache.  First version: correctness
// (every chain retraces the oracle, which retraces the reference), not speed - the pair bookkeeping recounts from the list.
#pragma once

__device__ __forceinline__ int *od_fc(Ctx &x) { return (int *)(x.tc_ + x.C.kmax); }
__device__ __forceinline__ int *od_fa(Ctx &x) { return (int *)(x.ta_ + x.C.kmax); }

// get_HybrF_full(tau, b1, s1, b2, s2), Trace.F90:1038-1057: table index (f1, f2)
__device__ __forceinline__ double hybr_full(const DevConst &C, int f1, int f2, double dtau)
{
    return hybr(C.ffull + ((size_t)f1 * C.nflav + f2) * C.nftau, C.nftau, C.beta, dtau);
}

// time-ordered list (times, codes with the true flavours) from the slot arrays and tags
__device__ W2DYN_INLINE void od_build_list(Ctx &x)
{
    const DevConst &C = x.C;
    const int kt = x.order(0);
    x.nops = 2 * kt;
    if (x.nops > C.wcap) { x.err |= ERR_WCAP; x.nops = 0; return; }
    wave_sync();
    const double *tcp = x.tc(0), *tap = x.ta(0); const int *fc = od_fc(x), *fa = od_fa(x);
    for (int e = x.lane; e < 2 * kt; e += 64) {
        const bool cre = e < kt; const int sl = cre ? e : e - kt;
        const double tau = cre ? tcp[sl] : tap[sl];
        int rank = 0;
        for (int j = 0; j < kt; ++j) rank += (tcp[j] < tau ? 1 : 0) + (tap[j] < tau ? 1 : 0);
        x.ot[rank] = tau; x.om[rank] = mat_code(cre ? fc[sl] : fa[sl], cre ? 1 : 2);
    }
    wave_sync();
}

__device__ W2DYN_INLINE void od_load(Ctx &x)
{
    const DevConst &C = x.C;
    const int kt = x.order(0);
    int *fc = od_fc(x), *fa = od_fa(x);
    for (int e = x.lane; e < kt; e += 64) { fc[e] = C.FC[(size_t)x.w * C.kmax + e]; fa[e] = C.FA[(size_t)x.w * C.kmax + e]; }
    wave_sync();
    od_build_list(x);
    mat_walk_sst(x);
}

__device__ W2DYN_INLINE void od_store(Ctx &x)
{
    const DevConst &C = x.C;
    const int kt = x.order(0);
    const int *fc = od_fc(x), *fa = od_fa(x);
    for (int e = x.lane; e < kt; e += 64) { C.FC[(size_t)x.w * C.kmax + e] = fc[e]; C.FA[(size_t)x.w * C.kmax + e] = fa[e]; }
}

// row (creators) or column (annihilators) an operator of flavour f at time t has in the reference's matrix: operators of its
// kind with a smaller flavour, or the same flavour and an earlier time (add_ftau_offset, Trace.F90:3002-3040)
__device__ W2DYN_INLINE int od_rank(Ctx &x, const double *t, const int *tag, int k, int f, double tau)
{
    int r = 0;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        const int fe = v ? tag[e] : 0; const double te = v ? t[e] : 0.0;
        r += count_pred(v && (fe < f || (fe == f && te < tau)));
    }
    return r;
}

// parity (+1 / -1) of the permutation that sorts slots by (flavour, time)
__device__ W2DYN_INLINE double od_sort_parity(Ctx &x, const double *t, const int *tag, int k)
{
    int inv = 0;
    for (int j = x.lane; j < k; j += 64) {
        const double tj = t[j]; const int fj = tag[j];
        for (int i = 0; i < j; ++i) inv += (tag[i] > fj || (tag[i] == fj && t[i] > tj)) ? 1 : 0;
    }
    return (wave_sum(inv) & 1) ? -1.0 : 1.0;
}

// Removable pairs of the time-ordered list (ensure_valid_NPairs_offdiag, Trace.F90:1820-1865): a creator and an annihilator
// of ANY flavours, both inside the window, at most taudiff_max apart.  pick > 0: the pick-th pair in the reference's enumeration
// (gen_OperRemove, Trace.F90:4879-4917: earlier operator by time, then the later one by time) comes back in (i1, i2).  Uses x.wc.
__device__ W2DYN_INLINE int od_pairs(Ctx &x, int pick, int &i1, int &i2)
{
    const int n = x.nops; const double tm = x.C.taudiff_max;
    int total = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + x.lane;
        int c = 0;
        if (e < n) {
            const double te = x.ot[e]; const int ce = x.om[e] & 1;
            if (te >= x.wmin && te <= x.wmax)
                for (int q = e + 1; q < n; ++q) {
                    const double tq = x.ot[q];
                    if (tq > x.wmax || tq > te + tm) break;
                    c += ((x.om[q] & 1) != ce) ? 1 : 0;
                }
            x.wc[e] = c;
        }
        total += c;
    }
    total = wave_sum(total);
    wave_sync();
    if (pick <= 0) return total;
    int estar = -1, rnk = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + x.lane;
        int P = 0; bool hit = false;
        if (e < n) { for (int q = 0; q < e; ++q) P += x.wc[q]; hit = (P < pick) && (pick <= P + x.wc[e]); }
        const unsigned long long m = __ballot(hit);
        if (m) { const int src = __ffsll((long long)m) - 1; estar = base + src; rnk = pick - __shfl(P, src, 64); }
    }
    estar = uni(estar); rnk = uni(rnk);
    const int ce = uni(x.om[estar]) & 1;
    int q = estar;
    while (rnk > 0) { ++q; if ((uni(x.om[q]) & 1) != ce) --rnk; }
    i1 = estar; i2 = q;
    return total;
}

// configuration sign factor of recalc_sign (Trace.F90:3500-3545): parity of the permutation that takes the hybridising operators
// from time order to the order (creators first; flavour, time descending), plus 1 + 2 + ... + (n/2 - 1).  From the list.
__device__ W2DYN_INLINE int od_sign_parity(Ctx &x)
{
    const int n = x.nops;
    int inv = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + x.lane;
        if (j < n) {
            const int cj = x.om[j];
            for (int i = 0; i < j; ++i) {
                const int ci = x.om[i];
                // the earlier operator has the smaller key: annihilator before creator, or same kind and flavour not larger
                inv += ((ci & 1) > (cj & 1) || ((ci & 1) == (cj & 1) && (ci >> 1) <= (cj >> 1))) ? 1 : 0;
            }
        }
    }
    const int kt = n / 2;
    return (wave_sum(inv) + (kt > 0 ? kt * (kt - 1) / 2 : 0)) & 1;
}

// F(tau_a_r - tau_c_c; flavour of a, flavour of c) over all slots, inverted (get_FullHybr_offdiag + get_MatLogDetFull,
// Trace.F90:2690-2770): build_and_invert with the two-flavour table
__device__ W2DYN_INLINE void od_build_and_invert(Ctx &x, int k, const double *tcp, const double *tap, const int *fcp, const int *fap, double *A, double &logdet, double &sgn)
{
    build_and_invert<true>(x, 0, k, tcp, tap, A, logdet, sgn, fcp, fap);
}

// StepAdd with offdiag (CTQMC.F90:3353-3612; pair_OperAdd with force_diagonal = false, get_LogDetRatPairAdd_full, Trace.F90:2835-2932)
template <int MAT>
__device__ W2DYN_INLINE void step_add_od(Ctx &x)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    x.count(C_TRYADD);
    const double width1 = x.wmax - x.wmin;
    const double tau1 = x.wmin + width1 * x.draw();                 // annihilator
    const int orb1 = x.randint(1, C.nd), spin1 = x.randint(1, 2);
    const double tmin = fmax(tau1 - C.taudiff_max, x.wmin);
    const double width2 = fmin(tau1 + C.taudiff_max, x.wmax) - tmin;
    const double tau2 = tmin + width2 * x.draw();                   // creator
    const int orb2 = x.randint(1, C.nd), spin2 = x.randint(1, 2);
    const double tdf = width1 * width2;
    const int fa = 2 * (orb1 - 1) + (spin1 - 1), fc = 2 * (orb2 - 1) + (spin2 - 1);
    const int kt = x.order(0);
    double *tcp = x.tc(0), *tap = x.ta(0); int *fcs = od_fc(x), *fas = od_fa(x);
    bool eq = (tau1 == tau2);
    for (int base = 0; base < x.nops; base += 64) {
        const int e = base + x.lane; const double t = e < x.nops ? x.ot[e] : -1.0;
        eq |= __ballot(t == tau1 || t == tau2) != 0ull;
    }
    if (eq) { x.count(C_QNADD); return; }
    if (x.nops + 2 > C.wcap) { x.err |= ERR_WCAP; return; }
    if (kt >= kmax) { x.err |= ERR_KMAX; x.count(C_QNADD); return; }
    const double lo = fmin(tau1, tau2), hi = fmax(tau1, tau2);
    const int code_c = mat_code(fc, 1), code_a = mat_code(fa, 2);
    const int code_lo = tau2 < tau1 ? code_c : code_a, code_hi = tau2 < tau1 ? code_a : code_c;
    // trial block sequence: the span between the two new operators is walked (check_sst_sequence, Trace.F90:1243-1302: from the
    // block in front of the earlier one through the old operators between them to the block behind the later one); the blocks
    // outside the span stay
    const int q_lo = mat_lower(x, lo), q_hi = mat_lower(x, hi);      // insertion indices in the old list
    {
        const int nf2 = 2 * C.nflav;
        wave_sync();
        int sb = uni(x.osst[q_lo]);
        sb = uni(x.t_conn[sb * nf2 + code_lo]);
        if (x.lane == 0) x.tsst[q_lo + 1] = sb;
        for (int i = q_lo; i < q_hi && sb >= 0; ++i) {
            sb = uni(x.t_conn[sb * nf2 + uni(x.om[i])]);
            if (x.lane == 0) x.tsst[i + 2] = sb;
        }
        if (sb >= 0) sb = uni(x.t_conn[sb * nf2 + code_hi]);
        if (sb < 0 || sb != uni(x.osst[q_hi])) { x.count(C_QNADD); return; }
        for (int j = x.lane; j <= x.nops + 2; j += 64) {
            if (j <= q_lo) x.tsst[j] = x.osst[j];
            else if (j >= q_hi + 2) x.tsst[j] = x.osst[j - 2];
        }
    }
    const int p_lo = q_lo, p_hi = q_hi + 1;
    mat_insert(x, p_lo, lo, code_lo); mat_insert(x, p_hi, hi, code_hi);

    // S = F(tau_a - tau_c) - R^T M Q; the flavour indices of the three look-ups as the reference writes them
    double *Mf = x.mat(0);
    for (int e = x.lane; e < kt; e += 64) {
        x.Q[e] = hybr_full(C, fc, fas[e], tap[e] - tau2);
        x.R[e] = hybr_full(C, fa, fcs[e], tau1 - tcp[e]);
    }
    const double S0 = hybr_full(C, fa, fc, tau1 - tau2);
    for (int e = x.lane; e < kt; e += 64) { x.MQ[e] = 0.0; x.RM[e] = 0.0; }
    wave_sync();
    double acc = 0.0;
    for (int e = x.lane; e < kt * kt; e += 64) {
        const int i = e / kt, j = e - i * kt;
        const double m = Mf[(size_t)i * kmax + j];
        acc += x.R[i] * m * x.Q[j];
        atomicAdd(&x.MQ[i], m * x.Q[j]); atomicAdd(&x.RM[j], x.R[i] * m);
    }
    const double S = S0 - wave_sum(acc);
    double tl_new, ts_new;
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new);
    const double rnd = x.draw();
    int du0, du1;
    const int np_new = od_pairs(x, 0, du0, du1);
    double prob = fabs(tdf * (double)((2 * C.nd) * (2 * C.nd)) / (double)np_new * fabs(S) * exp(tl_new - x.tr_log));
    if (ts_new == 0.0) prob = 0.0;
    if (!(rnd < prob)) { mat_remove(x, p_hi); mat_remove(x, p_lo); return; }

    const int pc = od_rank(x, tcp, fcs, kt, fc, tau2), pa = od_rank(x, tap, fas, kt, fa, tau1);
    const double sig = ((pc - pa) & 1) ? -1.0 : 1.0;
    const double invS = 1.0 / S;
    lds_sync();
    for (int e = x.lane; e < kt * kt; e += 64) {
        const int i = e / kt, j = e - i * kt;
        Mf[(size_t)i * kmax + j] += S * (-x.MQ[i] * invS) * (-x.RM[j] * invS);
    }
    for (int e = x.lane; e < kt; e += 64) { Mf[(size_t)e * kmax + kt] = -x.MQ[e] * invS; Mf[(size_t)kt * kmax + e] = -x.RM[e] * invS; }
    if (x.lane == 0) { Mf[(size_t)kt * kmax + kt] = invS; tcp[kt] = tau2; tap[kt] = tau1; fcs[kt] = fc; fas[kt] = fa; }
    x.set_order(0, kt + 1);
    if (kt + 1 >= kmax) x.err |= ERR_KMAX;
    wave_sync();
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = np_new; x.wvalid = 0;
    x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); x.pvalid = 0; x.kval = 0;
    x.count(C_ACCADD);
    accpair_add<false>(x, tau1 < tau2 ? fa : fc, hi - lo);          // flavour of the earlier operator (CTQMC.F90:3579)
}

// StepRem with offdiag (CTQMC.F90:3892-4150)
template <int MAT>
__device__ W2DYN_INLINE void step_rem_od(Ctx &x)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    x.count(C_TRYREM);
    int i1 = 0, i2 = 0;
    const int np = od_pairs(x, 0, i1, i2);
    x.npairs = np;
    if (np <= 0) { x.count(C_QNREM); return; }
    od_pairs(x, x.randint(1, np), i1, i2);
    const double t1 = uni(x.ot[i1]), t2 = uni(x.ot[i2]); const int c1 = uni(x.om[i1]), c2 = uni(x.om[i2]);
    const bool first_is_c = (c1 & 1) == 0;
    const double tau_c = first_is_c ? t1 : t2, tau_a = first_is_c ? t2 : t1;
    const int fc = (first_is_c ? c1 : c2) >> 1, fa = (first_is_c ? c2 : c1) >> 1;
    const double hybf = (double)np / ((x.wmax - x.wmin) * (fmin(tau_a + C.taudiff_max, x.wmax) - fmax(tau_a - C.taudiff_max, x.wmin)));
    const int kt = x.order(0);
    double *tcp = x.tc(0), *tap = x.ta(0); int *fcs = od_fc(x), *fas = od_fa(x);
    int ic = -1, ia = -1;
    for (int base = 0; base < kt; base += 64) {
        const int e = base + x.lane; const bool v = e < kt;
        const unsigned long long mc = __ballot(v && tcp[e] == tau_c), ma = __ballot(v && tap[e] == tau_a);
        if (mc) ic = base + __ffsll((long long)mc) - 1;
        if (ma) ia = base + __ffsll((long long)ma) - 1;
    }
    if (ic < 0 || ia < 0) { x.err |= ERR_BLOCKSEQ; return; }
    {   // the span between the two operators is walked from the block in front of the earlier one through the operators between
        // them; it must arrive at the block behind the later one (check_sst_sequence); the blocks outside the span stay
        const int nf2 = 2 * C.nflav;
        wave_sync();
        int sb = uni(x.osst[i1]);
        for (int i = i1 + 1; i < i2 && sb >= 0; ++i) {
            if (x.lane == 0) x.tsst[i - 1] = sb;
            sb = uni(x.t_conn[sb * nf2 + uni(x.om[i])]);
        }
        if (sb < 0 || sb != uni(x.osst[i2 + 1])) { x.count(C_QNREM); return; }
        for (int j = x.lane; j <= x.nops - 2; j += 64) {
            if (j < i1) x.tsst[j] = x.osst[j];
            else if (j >= i2 - 1) x.tsst[j] = x.osst[j + 2];
        }
    }
    mat_remove(x, i2); mat_remove(x, i1);
    wave_sync();
    double *Mf = x.mat(0);
    const double S = uni(Mf[(size_t)ic * kmax + ia]);
    double tl_new, ts_new;
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new);
    double prob = fabs(hybf / (double)((2 * C.nd) * (2 * C.nd)) * fabs(S) * exp(tl_new - x.tr_log));
    if (ts_new == 0.0) prob = 0.0;
    const double rnd = x.draw();
    if (!(rnd < prob)) { mat_insert(x, i1, t1, c1); mat_insert(x, i2, t2, c2); return; }
    if (fabs(S) < 1e-12) { x.err |= ERR_DETRAT; mat_insert(x, i1, t1, c1); mat_insert(x, i2, t2, c2); return; }
    // ranks of the two operators among the others of their kind
    const int pc = od_rank(x, tcp, fcs, kt, fc, tau_c), pa = od_rank(x, tap, fas, kt, fa, tau_a);
    const double sig = ((pc - pa) & 1) ? -1.0 : 1.0;
    const int last = kt - 1;
    const double invS = 1.0 / S;
    for (int e = x.lane; e < kt; e += 64) { x.RM[e] = Mf[(size_t)ic * kmax + e]; x.MQ[e] = Mf[(size_t)e * kmax + ia] * invS; }
    wave_sync();
    // M'_ij = M_ij - M_i,ia M_ic,j / S for i != ic, j != ia, the last row / column moving into the freed slots: first all new values
    // in place (rows ic / columns ia excluded), then the moves
    for (int e = x.lane; e < kt * kt; e += 64) {
        const int i = e / kt, j = e - i * kt;
        if (i != ic && j != ia) Mf[(size_t)i * kmax + j] -= x.MQ[i] * x.RM[j];
    }
    wave_sync();
    if (ic != last) for (int j = x.lane; j < kt; j += 64) Mf[(size_t)ic * kmax + j] = Mf[(size_t)last * kmax + j];
    wave_sync();
    if (ia != last) for (int i = x.lane; i < kt; i += 64) Mf[(size_t)i * kmax + ia] = Mf[(size_t)i * kmax + last];
    wave_sync();
    if (x.lane == 0) { tcp[ic] = tcp[last]; fcs[ic] = fcs[last]; tap[ia] = tap[last]; fas[ia] = fas[last]; }
    x.set_order(0, last);
    wave_sync();
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = -1; x.wvalid = 0;
    x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); x.pvalid = 0; x.kval = 0;
    x.count(C_ACCREM);
    accpair_add<false>(x, fc, t2 - t1);                             // flavour of the creator (CTQMC.F90:4114-4122)
}

// tau shift with offdiag (CTQMC.F90:4641-4835; update_InvFullHybrShiftTau_offdiag, Trace.F90:2462-2601)
template <int MAT>
__device__ W2DYN_INLINE void step_shift_od(Ctx &x)
{
    const DevConst &C = x.C;
    const int kt = x.order(0), kmax = C.kmax;
    if (kt == 0) {
        step_change_outer(x);
        x.pvalid = 0;
        mat_walk_sst(x); mat_trace_current<MAT>(x);
        return;
    }
    const double dt = x.draw() * C.beta;
    if (dt == 0.0 || dt == C.beta) return;
    x.count(C_TRYSHIFT);
    int outer = x.outer;
    { const int first = mat_lower(x, dt); if (first < x.nops) outer = uni(x.osst[first]); }
    double *tcp = x.tc(0), *tap = x.ta(0), *Mf = x.mat(0); const int *fcs = od_fc(x), *fas = od_fa(x);
    // rows of creators and columns of annihilators that wrap change sign; per flavour the block's rows (columns) rotate, which
    // costs (-1)^(size * wrapped)
    double detrat = 1.0;
    for (int f = 0; f < C.nflav; ++f) {
        int sr = 0, wr = 0, sc = 0, wc = 0;
        for (int base = 0; base < kt; base += 64) {
            const int e = base + x.lane; const bool v = e < kt;
            sr += count_pred(v && fcs[e] == f); wr += count_pred(v && fcs[e] == f && tcp[e] < dt);
            sc += count_pred(v && fas[e] == f); wc += count_pred(v && fas[e] == f && tap[e] < dt);
        }
        if ((sr * wr) & 1) detrat = -detrat;
        if ((sc * wc) & 1) detrat = -detrat;
    }
    for (int e = x.lane; e < kt * kt; e += 64) {
        const int i = e / kt, j = e - i * kt;
        if ((tcp[i] < dt) != (tap[j] < dt)) Mf[(size_t)i * kmax + j] = -Mf[(size_t)i * kmax + j];
    }
    wave_sync();
    for (int e = x.lane; e < kt; e += 64) {
        double a = tcp[e] - dt; if (a < 0.0) a += C.beta; tcp[e] = a;
        double b = tap[e] - dt; if (b < 0.0) b += C.beta; tap[e] = b;
    }
    wave_sync();
    (void)x.draw();
    x.outer = outer;
    x.det_sign *= detrat;
    {
        const int nw = C.num_windows;
        int a = (int)floor((dt / C.beta) * (double)(nw + 1)); a = ((a % nw) + nw) % nw;
        x.wpos = (((x.wpos - a) % nw) + nw) % nw;
        x.window_bounds();
    }
    x.npairs = -1; x.wvalid = 0;
    od_build_list(x); mat_walk_sst(x); mat_trace_current<MAT>(x); x.pvalid = 0;
    x.count(C_ACCSHIFT);
}

// global move with offdiag (CTQMC.F90:4433-4554): the flavours of all operators are relabelled (and creators / annihilators
// exchanged), the full matrix is built anew, DetRat = DetNew / Det
template <int MAT>
__device__ W2DYN_INLINE void step_glob_od(Ctx &x)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd, kmax = C.kmax;
    x.count(C_TRYGLOB);
    if (x.nops == 0) return;
    x.wvalid = 0;
    int gu[MAXFLAV + 1];
    const double r = x.draw();
    bool spinflip = false;
    if (r < 0.3) { spinflip = true; gu[0] = 0; for (int i = 1; i <= nd; ++i) gu[i] = i + nd; for (int i = nd + 1; i <= 2 * nd; ++i) gu[i] = i - nd; }
    else if (r < 0.6) { gu[0] = 1; for (int i = 1; i <= nf; ++i) gu[i] = i; }
    else if (r < 0.8 && C.nsym > 0) {
        const int isym = x.randint(1, 2 * C.nsym) - 1;
        gu[0] = 0; for (int i = 1; i <= nf; ++i) gu[i] = C.symmoves[isym * nf + i - 1];
    } else {
        for (int i = 1; i <= nf; ++i) gu[i] = i;
        for (int j = 1; j <= nf; ++j) { const int i = x.randint(j, nf); const int t = gu[j]; gu[j] = gu[i]; gu[i] = t; }
        gu[0] = x.randint(0, 1);
    }
    const bool flip = gu[0] != 0;
    const int kt = x.order(0);
    int *lut = x.wm;                                   // code -> relabelled code; gmap behind it
    if (x.lane < nf) {
        const int f = x.lane, num = (f >> 1) + nd * (f & 1) + 1, t = gu[num];
        const int g = 2 * ((t - 1) % nd) + (t - 1) / nd;
        lut[2 * f] = 2 * g + (flip ? 1 : 0); lut[2 * f + 1] = 2 * g + (flip ? 0 : 1);
        lut[2 * MAXFLAV + f] = g;
    }
    wave_sync();
    // new slot arrays and tags into the chain's scratch: times, then the tags as integers behind them
    double *TS = C.TS + (size_t)x.w * nf * 2 * kmax;
    double *tcn = TS, *tan = TS + kmax; int *fcn = (int *)(TS + 2 * kmax), *fan = fcn + kmax;
    {
        const double *tcp = x.tc(0), *tap = x.ta(0); const int *fcs = od_fc(x), *fas = od_fa(x);
        for (int e = x.lane; e < kt; e += 64) {
            tcn[e] = flip ? tap[e] : tcp[e]; fcn[e] = lut[2 * MAXFLAV + (flip ? fas[e] : fcs[e])];
            tan[e] = flip ? tcp[e] : tap[e]; fan[e] = lut[2 * MAXFLAV + (flip ? fcs[e] : fas[e])];
        }
    }
    int outer_new = x.outer;
    double sst_factor = 1.0;
    const bool single = spinflip && C.has_szt;
    int pold = 0;
    if (!single) pold = mat_possible(x, 0);
    for (int e = x.lane; e < x.nops; e += 64) { const int c = x.om[e]; x.wc[e] = c; x.om[e] = lut[c]; }   // backup + relabel
    wave_sync();
    if (single) {
        const int st = C.sst_first[x.outer];
        const int down = st & ((1 << nd) - 1), up = (st >> nd) & ((1 << nd) - 1);
        outer_new = C.sst_of_state[up | (down << nd)];
        const int nf2 = 2 * nf;
        int s = outer_new;
        for (int i = 0; i < x.nops && s >= 0; ++i) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
        if (s != outer_new) sst_factor = 0.0;
    } else {
        const int pnew = mat_possible(x, 0);
        if (pnew <= 0) sst_factor = 0.0;
        else {
            sst_factor = (double)pnew / (double)pold;
            outer_new = mat_possible(x, x.randint(1, pnew));
        }
    }
    if (sst_factor == 0.0) {
        for (int e = x.lane; e < x.nops; e += 64) x.om[e] = x.wc[e];
        wave_sync();
        x.count(C_QNGLOB); return;
    }
    double *Mb = C.MB + (size_t)x.w * nf * kmax * kmax;
    double ld, sg;
    // a relabelling that leaves creators creators and maps every two-flavour table onto an identical one leaves the hybridisation matrix - slot
    // by slot - as it is: the inverse and log|det| stay, the sign changes by the parities of the (flavour, time) orders before and after
    // (the reference builds and inverts anew and gets the same numbers up to rounding; see step_glob)
    bool carried = !flip;
    if (carried) {
        int bad = 0;
        for (int e = x.lane; e < nf * nf; e += 64) { const int a = e / nf, c = e - a * nf; bad |= C.fclass[lut[2 * MAXFLAV + a] * nf + lut[2 * MAXFLAV + c]] != C.fclass[e] ? 1 : 0; }
        carried = __ballot(bad != 0) == 0ull;
    }
    bool singular = false;
    if (carried) {
        ld = x.det_log;
        sg = x.det_sign * od_sort_parity(x, x.tc(0), od_fc(x), kt) * od_sort_parity(x, x.ta(0), od_fa(x), kt);
    } else {
        const int err0 = x.err;
        od_build_and_invert(x, kt, tcn, tan, fcn, fan, Mb, ld, sg);
        // relabelled flavours that the bath does not couple leave a matrix with a vanishing determinant: the reference takes
        // det = 0 and rejects the move (MatrixUpdate.F90:74-78); nothing to report
        singular = (x.err & ERR_SINGULAR) != 0;
        x.err = err0;
    }
    sg *= od_sort_parity(x, tcn, fcn, kt) * od_sort_parity(x, tan, fan, kt);
    double tl, ts;
    mat_walk_into(x, outer_new, x.tsst);
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl, ts);
    const double rnd = x.draw();
    const double ratio = (ts == 0.0 || singular) ? 0.0 : fabs(exp((tl - x.tr_log) + (ld - x.det_log)) * sst_factor);
    if (!(rnd < ratio)) {
        for (int e = x.lane; e < x.nops; e += 64) x.om[e] = x.wc[e];
        wave_sync();
        return;
    }
    {
        double *tcp = x.tc(0), *tap = x.ta(0), *Mg = x.mat(0); int *fcs = od_fc(x), *fas = od_fa(x);
        for (int e = x.lane; e < kt; e += 64) { tcp[e] = tcn[e]; tap[e] = tan[e]; fcs[e] = fcn[e]; fas[e] = fan[e]; }
        if (!carried) for (int e = x.lane; e < kt * kt; e += 64) { const int i = e / kt, j = e - i * kt; Mg[(size_t)i * kmax + j] = Mb[(size_t)i * kmax + j]; }
    }
    wave_sync();
    x.outer = outer_new;
    x.det_log = ld; x.det_sign = sg;
    x.tr_log = tl; x.tr_sign = ts;
    mat_commit_sst(x);
    x.npairs = -1; x.pvalid = 0; x.kval = 0;
    x.count(C_ACCGLOB);
}

// StepFlavourchange_general (CTQMC.F90:4838-4989; flavourchange_moves = 1): two operators inside the window, chosen as
// propose_flavourexchange_general does (Trace.F90:2191-2274), get new random flavours; their kinds and times stay.  Quantum
// numbers over the span between them, trace in full, the matrix built anew, DetRat = DetNew / Det.
template <int MAT>
__device__ W2DYN_INLINE void step_flavc_od(Ctx &x)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, kmax = C.kmax;
    x.count(C_TRYFLAVC);
    const int jlo = mat_lower(x, x.wmin), N = mat_lower_eq(x, x.wmax) - jlo;
    if (N <= 1) return;
    const int pos1 = x.randint(1, N - 1), pos2 = x.randint(1, N - pos1);
    const int i1 = jlo + pos1 - 1, i2 = i1 + pos2;
    const int f1n = 2 * (x.randint(1, C.nd) - 1) + (x.randint(1, 2) - 1);
    const int f2n = 2 * (x.randint(1, C.nd) - 1) + (x.randint(1, 2) - 1);
    const int c1 = uni(x.om[i1]), c2 = uni(x.om[i2]);
    const int c1n = 2 * f1n + (c1 & 1), c2n = 2 * f2n + (c2 & 1);
    const double t1 = uni(x.ot[i1]), t2 = uni(x.ot[i2]);
    const int kt = x.order(0);
    double *tcp = x.tc(0), *tap = x.ta(0); int *fcs = od_fc(x), *fas = od_fa(x);
    {   // block sequence: walked anew between the two operators, must arrive at the block behind the later one
        const int nf2 = 2 * nf;
        wave_sync();
        int sb = uni(x.osst[i1]);
        sb = uni(x.t_conn[sb * nf2 + c1n]);
        for (int i = i1 + 1; i < i2 && sb >= 0; ++i) {
            if (x.lane == 0) x.tsst[i] = sb;
            sb = uni(x.t_conn[sb * nf2 + uni(x.om[i])]);
        }
        if (sb >= 0) { if (x.lane == 0) x.tsst[i2] = sb; sb = uni(x.t_conn[sb * nf2 + c2n]); }
        if (sb < 0 || sb != uni(x.osst[i2 + 1])) return;
        for (int j = x.lane; j <= x.nops; j += 64) if (j <= i1 || j > i2) x.tsst[j] = x.osst[j];
    }
    if (x.lane == 0) { x.om[i1] = c1n; x.om[i2] = c2n; }
    wave_sync();
    double tl, ts;
    mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl, ts);
    // slot arrays with the two flavours changed, into the chain's scratch
    double *TS = C.TS + (size_t)x.w * nf * 2 * kmax;
    double *tcn = TS, *tan = TS + kmax; int *fcn = (int *)(TS + 2 * kmax), *fan = fcn + kmax;
    for (int e = x.lane; e < kt; e += 64) {
        tcn[e] = tcp[e]; tan[e] = tap[e];
        int fc = fcs[e], fa = fas[e];
        if ((c1 & 1) == 0 && tcp[e] == t1) fc = f1n;
        if ((c2 & 1) == 0 && tcp[e] == t2) fc = f2n;
        if ((c1 & 1) == 1 && tap[e] == t1) fa = f1n;
        if ((c2 & 1) == 1 && tap[e] == t2) fa = f2n;
        fcn[e] = fc; fan[e] = fa;
    }
    wave_sync();
    x.wvalid = 0;
    double *Mb = C.MB + (size_t)x.w * nf * kmax * kmax;
    double ld, sg;
    const int err0 = x.err;
    od_build_and_invert(x, kt, tcn, tan, fcn, fan, Mb, ld, sg);
    const bool singular = (x.err & ERR_SINGULAR) != 0;
    x.err = err0;
    sg *= od_sort_parity(x, tcn, fcn, kt) * od_sort_parity(x, tan, fan, kt);
    const double prob = (ts == 0.0 || singular) ? 0.0 : fabs(exp((tl - x.tr_log) + (ld - x.det_log)));
    const double rnd = x.draw();
    if (!(rnd < prob)) {
        if (x.lane == 0) { x.om[i1] = c1; x.om[i2] = c2; }
        wave_sync();
        return;
    }
    {
        double *Mg = x.mat(0);
        for (int e = x.lane; e < kt; e += 64) { fcs[e] = fcn[e]; fas[e] = fan[e]; }
        for (int e = x.lane; e < kt * kt; e += 64) { const int i = e / kt, j = e - i * kt; Mg[(size_t)i * kmax + j] = Mb[(size_t)i * kmax + j]; }
    }
    wave_sync();
    x.det_log = ld; x.det_sign = sg;
    x.tr_log = tl; x.tr_sign = ts;
    mat_commit_sst(x);
    x.pvalid = 0; x.kval = 0;
    x.count(C_ACCFLAVC);
}

// sweep kernel with offdiag: the Z-space loop of k_sweep (CTQMC.F90:8428-8533) over the moves above
template <int MAT>
__global__ __launch_bounds__(256, W2DYN_WPE_MAT) void k_sweep_od(DevConst C, long long nsteps_launch, int count_sampling, int resume)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    for (;;) {
        ctx_setup(x, smem, next_chain(C));
        if (x.w >= C.nwalkers) break;
        const long long nsteps = launch_steps(C, x.w, x.lane, nsteps_launch, resume);
        if (nsteps == 0) continue;
        ctx_load(x);
        od_load(x);
        flag_full(x);
        mat_trace_current<MAT>(x);
        const double pg = C.p_glob, ps = C.p_shift;
        unsigned int phase = (unsigned int)((x.cnt[C_TRYADD] + x.cnt[C_TRYREM]) % (unsigned long long)C.nslide);
        phase = (unsigned int)uni((int)phase);
        long long it = 0;
        for (; it < nsteps && !(x.err & ERR_KMAX); ++it) {
            const double r = x.draw();
            // CTQMC.F90:8441-8449, 8456-8464: with flavourchange_moves the flavour change takes the place of the two-pair moves
            // (probability Percentage4OperatorMove; the host lets p_4op > 0 through only then)
            if (r <= 1.0 - ps - pg) {
                if (x.draw() >= C.p_4op) {
                    x.skip(1);
                    if (r < (1.0 - ps - pg) / 2.0) step_add_od<MAT>(x); else step_rem_od<MAT>(x);
                    phase += 1;
                } else step_flavc_od<MAT>(x);
            }
            else if (r <= 1.0 - ps) step_glob_od<MAT>(x);
            else step_shift_od<MAT>(x);
            if (phase == (unsigned int)C.nslide) phase = 0;
            if (phase == 0) {
                if (C.num_windows > 1) {
                    x.wpos += 1; if (x.wpos == C.num_windows) x.wpos = 0;
                    x.window_bounds(); x.npairs = -1; x.wvalid = 0; x.pvalid = 0;
                } else x.wpos = 0;
            }
        }
        if (count_sampling && x.lane == 0) x.cnt[C_SAMPLING] += (unsigned long long)it;
        od_store(x);
        ctx_store(x, nsteps - it);
    }
}

#ifdef W2DYN_TU_OFFDIAG
// MeasGtauRem_full (CTQMC.F90:2475-2584): every element of the full matrix into gtau_full / gleg_full[flavour of the
// annihilator][flavour of the creator]; one wave per chain, lane = element, the Legendre recursion of the reference
__global__ __launch_bounds__(256) void k_meas_full(DevConst C, double *__restrict__ acc_gleg_full, double *__restrict__ acc_gtau_full, int lds_leg)
{
    extern __shared__ double sleg[];           // lds_leg: [nflav^2][nleg] sums of the workgroup's four chains (they share a bin: walkers_per_bin is a multiple of 4)
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nf = C.nflav, nleg = C.nleg, ntau = C.ntau, kmax = C.kmax;
    if (lds_leg) { for (int i = threadIdx.x; i < nf * nf * nleg; i += blockDim.x) sleg[i] = 0.0; __syncthreads(); }
    if (w < C.nwalkers) {
        const int kt = C.K[(size_t)w * nf];
        const double *tc = C.TC + (size_t)w * nf * kmax, *ta = C.TA + (size_t)w * nf * kmax, *M = C.M + (size_t)w * nf * kmax * kmax;
        const int *fc = C.FC + (size_t)w * kmax, *fa = C.FA + (size_t)w * kmax;
        const double cfgsign = C.cfgsign[w], beta = C.beta;
        const int bin = w / C.walkers_per_bin;
        double *gl = acc_gleg_full + (size_t)bin * nf * nf * nleg, *gt = acc_gtau_full + (size_t)bin * nf * nf * ntau;
        for (int e = lane; e < kt * kt; e += 64) {
            const int i = e / kt, j = e - i * kt;
            double m = M[(size_t)i * kmax + j], xx; int idt;
            const double d = ta[j] - tc[i];
            if (d > 0.0) { idt = (int)(d / beta * (double)ntau) + 1; xx = 2.0 / beta * d - 1.0; }
            else { idt = (int)((beta + ta[j] - tc[i]) / beta * (double)ntau) + 1; xx = 2.0 / beta * (beta + ta[j] - tc[i]) - 1.0; m = -m; }
            m = m * cfgsign;
            if (idt > ntau) idt = ntau;
            const size_t ff = (size_t)fa[j] * nf + fc[i];
            atomicAdd(&gt[ff * ntau + idt - 1], m);
            double *dst = lds_leg ? sleg + ff * nleg : gl + ff * nleg;
            double plx = 0.0, plm1 = 0.0, plp1 = 0.0;
            for (int il = 0; il < nleg; ++il) {
                if (il == 0) plp1 = 1.0;
                else if (il == 1) plp1 = xx;
                else { const int l = il - 1; plp1 = (2.0 * (double)l + 1.0) / (double)(l + 1) * xx * plx - (double)l / (double)(l + 1) * plm1; }
                atomicAdd(&dst[il], m * plp1);
                plm1 = plx; plx = plp1;
            }
        }
    }
    if (lds_leg) {
        __syncthreads();
        const int w0 = blockIdx.x * (blockDim.x >> 6);
        if (w0 < C.nwalkers) {
            double *gl = acc_gleg_full + (size_t)(w0 / C.walkers_per_bin) * nf * nf * nleg;
            for (int i = threadIdx.x; i < nf * nf * nleg; i += blockDim.x) { const double v = sleg[i]; if (v != 0.0) atomicAdd(&gl[i], v); }
        }
    }
}
#endif  // W2DYN_TU_OFFDIAG
