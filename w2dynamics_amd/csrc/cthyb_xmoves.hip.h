// Moves outside the default mix, compiled into the k_sweep<..., XM = true> kernels only (the host picks them when
// Percentage4OperatorMove or PercentageOuterMove is set): two pairs at once, StepAdd4 / StepRem4 (CTQMC.F90:3615-3889,
// 4153-4430).  Like the reference these moves build the inverse matrices of the flavours they touch anew instead of
// updating them (get_InvFullHybr, Trace.F90:2279-2333); the old determinant comes from the same routine.  The pair moves
// that change the outer block (PercentageOuterMove) are branches of step_add / step_rem.
#pragma once
// (included from cthyb_kernels.hip.h, inside namespace w2dyn)

// removable pairs a creator at tc and an annihilator at ta of flavour f add to the present configuration (update_NPairs,
// Trace.F90:1886-1935)
__device__ W2DYN_INLINE int xm_dnp(Ctx &x, int f, double tc, double ta)
{
    const DevConst &C = x.C;
    const int k = x.order(f);
    const double *tcp = x.tc(f), *tap = x.ta(f);
    int d = 0;
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        const double tce = v ? tcp[e] : 0.0, tae = v ? tap[e] : 0.0;
        d += count_pred(v && pair_ok(tc, tae, x.wmin, x.wmax, C.taudiff_max));
        d += count_pred(v && pair_ok(tce, ta, x.wmin, x.wmax, C.taudiff_max));
    }
    return d + (pair_ok(tc, ta, x.wmin, x.wmax, C.taudiff_max) ? 1 : 0);
}

// One-state blocks: may a pair (creator tc, annihilator ta) of flavour f be put into the present configuration on its own
// (no operator of the flavour between the two, the occupation below them fitting the earlier one)
__device__ W2DYN_INLINE bool xm_pair_fits(Ctx &x, int f, double tc, double ta)
{
    const DevConst &C = x.C;
    const int k = x.order(f);
    const double *tcp = x.tc(f), *tap = x.ta(f);
    const double lo = fmin(tc, ta), hi = fmax(tc, ta);
    int c_lo = 0, a_lo = 0; bool blocked = (tc == ta);
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        const double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        c_lo += count_pred(tce < lo); a_lo += count_pred(tae < lo);
        blocked |= __ballot((tce >= lo && tce <= hi) || (tae >= lo && tae <= hi)) != 0ull;
    }
    const int occ_lo = ((x.outer >> flav_bit(f, C.nd)) & 1) + c_lo - a_lo;
    return !blocked && occ_lo == (tc < ta ? 0 : 1);
}

// Four operators of ONE flavour (times t[0..3], ca[i] = 1 creator / 2 annihilator), added to (add = true) or taken out of the
// present configuration of one-state blocks.  s[] receives the times in ascending order, first[] the kind of the operators at
// s[0] and s[2]: the occupation of the flavour changes on (s0,s1) and (s2,s3).  The new sequence alternates iff no other
// operator of the flavour lies inside these two intervals and, for an insertion, each new operator meets the occupation it needs.
__device__ W2DYN_INLINE bool xm_four_fit(Ctx &x, int f, const double (&t)[4], const int (&ca)[4], bool add, double (&s)[4], int (&first)[2])
{
    const DevConst &C = x.C;
    int idx[4] = {0, 1, 2, 3};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3 - a; ++b) if (t[idx[b]] > t[idx[b + 1]]) { const int q = idx[b]; idx[b] = idx[b + 1]; idx[b + 1] = q; }
    int cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] = t[idx[i]]; cs[i] = ca[idx[i]]; }
    first[0] = cs[0]; first[1] = cs[2];
    if (s[0] == s[1] || s[1] == s[2] || s[2] == s[3]) return false;
    const int k = x.order(f);
    const double *tcp = x.tc(f), *tap = x.ta(f);
    bool bad = false; int below[4] = {0, 0, 0, 0};
    for (int base = 0; base < k; base += 64) {
        const int e = base + x.lane; const bool v = e < k;
        const double tce = v ? tcp[e] : 3.0 * C.beta, tae = v ? tap[e] : 3.0 * C.beta;
        // the four operators themselves (removal) sit on the ends of the intervals, never inside
        bad |= __ballot(v && ((tce > s[0] && tce < s[1]) || (tce > s[2] && tce < s[3]) || (tae > s[0] && tae < s[1]) || (tae > s[2] && tae < s[3]))) != 0ull;
        if (add) {
            bad |= __ballot(v && (tce == s[0] || tce == s[1] || tce == s[2] || tce == s[3] || tae == s[0] || tae == s[1] || tae == s[2] || tae == s[3])) != 0ull;
#pragma unroll
            for (int i = 0; i < 4; ++i) below[i] += count_pred(tce < s[i]) + count_pred(tae < s[i]);
        }
    }
    if (bad) return false;
    if (add) {
        const int n0 = (x.outer >> flav_bit(f, C.nd)) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) if ((cs[i] == 1) != (((n0 + below[i] + i) & 1) == 0)) return false;      // a creator needs the empty level
    }
    return true;
}

// Removable pairs of the window list in LDS (x.wt, x.wm as build_window has just left them) without the entries ex[0..3]
// (-1: none); with pick > 0 the pick-th pair in the reference's enumeration (gen_OperRemove, Trace.F90:4879-4917: by the time
// of the earlier operator, then of the later one) comes back in (e1, e2).  Uses x.wc; the window list is stale afterwards.
__device__ W2DYN_INLINE int xm_pairs(Ctx &x, const int (&ex)[4], int pick, int &e1, int &e2)
{
    const int nw = x.nwin; const double tm = x.C.taudiff_max;
    int total = 0;
    for (int base = 0; base < nw; base += 64) {
        const int e = base + x.lane;
        int c = 0;
        if (e < nw) {
            if (e != ex[0] && e != ex[1] && e != ex[2] && e != ex[3]) {
                const double te = x.wt[e], thi = te + tm; const int want = (x.wm[e] & 0xffff) ^ 0x300;
                for (int q = 0; q < nw; ++q) {
                    const double tq = x.wt[q];
                    c += ((x.wm[q] & 0xffff) == want && tq > te && !(tq > thi) && q != ex[0] && q != ex[1] && q != ex[2] && q != ex[3]) ? 1 : 0;
                }
            }
            x.wc[e] = c;
        }
        total += c;
    }
    total = wave_sum(total);
    wave_sync();
    x.wvalid = 0;
    if (pick <= 0) return total;
    int estar = -1, rnk = 0;
    for (int base = 0; base < nw; base += 64) {
        const int e = base + x.lane;
        int P = 0; bool hit = false;
        if (e < nw) {
            const double te = x.wt[e];
            for (int q = 0; q < nw; ++q) P += (x.wt[q] < te) ? x.wc[q] : 0;
            hit = (P < pick) && (pick <= P + x.wc[e]);
        }
        const unsigned long long m = __ballot(hit);
        if (m) { const int src = __ffsll((long long)m) - 1; estar = base + src; rnk = pick - __shfl(P, src, 64); }
    }
    estar = uni(estar); rnk = uni(rnk);
    const double t1 = x.wt[estar], thi = t1 + tm; const int want = (x.wm[estar] & 0xffff) ^ 0x300;
    int qstar = -1;
    for (int base = 0; base < nw; base += 64) {
        const int q = base + x.lane;
        bool hit = false;
        if (q < nw && q != ex[0] && q != ex[1] && q != ex[2] && q != ex[3]) {
            const double tq = x.wt[q];
            if ((x.wm[q] & 0xffff) == want && tq > t1 && !(tq > thi)) {
                int less = 0;
                for (int p = 0; p < nw; ++p) {
                    const double tp = x.wt[p];
                    less += ((x.wm[p] & 0xffff) == want && tp > t1 && !(tp > thi) && tp < tq && p != ex[0] && p != ex[1] && p != ex[2] && p != ex[3]) ? 1 : 0;
                }
                hit = less == rnk - 1;
            }
        }
        const unsigned long long m = __ballot(hit);
        if (m) qstar = base + __ffsll((long long)m) - 1;
    }
    e1 = estar; e2 = uni(qstar);
    return total;
}

// sum over the ways a removal of two pairs takes the same four operators out (CTQMC.F90:3684-3772, 4207-4289);
// crossed: both mixed pairs (c1,a2) and (c2,a1) are removable pairs as well; n14: count with the mixed pair of the earlier
// operator of the first pair only, n23: with the other mixed pair only
__device__ __forceinline__ double xm_rempair_factor(int n_all, int n12, int n34, bool crossed, int n14, int n23, double &tdf)
{
    if (crossed) {
        tdf = tdf / 4.0;
        return (1.0 / (double)n_all) * (1.0 / (double)n12 + 1.0 / (double)n34 + 1.0 / (double)n14 + 1.0 / (double)n23);
    }
    tdf = tdf / 2.0;
    return (1.0 / (double)n_all) * (1.0 / (double)n12 + 1.0 / (double)n34);
}

// change of the log of the scalar local trace when the occupations of flavour f1 on A = (a0,a1) and of f2 on B = (b0,b1)
// change by s1, s2 = +-1 (f1 != f2), every integral taken over the PRESENT configuration:
//   - [ s1 I_f1(A) + s2 I_f2(B) + U_f1f2 s1 s2 |A and B| ]
__device__ W2DYN_INLINE double xm_dlog_two(Ctx &x, int f1, double a0, double a1, double s1, int f2, double b0, double b1, double s2)
{
    const DevConst &C = x.C;
    const double i1 = level_integral<0>(x, f1, a0, a1), i2 = level_integral<0>(x, f2, b0, b1);
    const double ov = fmax(fmin(a1, b1) - fmax(a0, b0), 0.0);
    return -(s1 * i1 + s2 * i2 + C.udd[f1 * C.nflav + f2] * s1 * s2 * ov);
}

// new inverse matrix of flavour g for the operators in TS (kn pairs) into the chain's second buffer; the old block's
// determinant by the same route.  Adds log|det F_new| - log|det F_old| to dlog, multiplies sgn by the two signs (time order).
__device__ W2DYN_INLINE void xm_rebuild(Ctx &x, int g, int kn, double &dlog, double &sgn)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax, ko = x.order(g);
    double *Mb = C.MB + ((size_t)x.w * C.nflav + g) * kmax * kmax;
    const double *tcn = C.TS + ((size_t)x.w * C.nflav * 2 + 2 * g) * kmax, *tan = tcn + kmax;
    double ld, sg;
    build_and_invert(x, g, ko, x.tc(g), x.ta(g), Mb, ld, sg);
    if (ko > 0) sg *= sort_parity(x, x.tc(g), ko) * sort_parity(x, x.ta(g), ko);
    dlog -= ld; sgn *= sg;
    wave_sync();
    build_and_invert(x, g, kn, tcn, tan, Mb, ld, sg);
    if (kn > 0) sg *= sort_parity(x, tcn, kn) * sort_parity(x, tan, kn);
    dlog += ld; sgn *= sg;
}

// accepted: the operators in TS and the matrix in the second buffer become flavour g's
__device__ W2DYN_INLINE void xm_install(Ctx &x, int g, int kn)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    const double *Mn = C.MB + ((size_t)x.w * C.nflav + g) * kmax * kmax;
    const double *tcn = C.TS + ((size_t)x.w * C.nflav * 2 + 2 * g) * kmax, *tan = tcn + kmax;
    double *tcp = x.tc(g), *tap = x.ta(g), *Mg = x.mat(g);
    for (int e = x.lane; e < kn; e += 64) { tcp[e] = tcn[e]; tap[e] = tan[e]; }
    for (int j0 = 0; j0 < kn; j0 += 64) { const int j = j0 + x.lane; if (j < kn) for (int i = 0; i < kn; ++i) Mg[(size_t)i * kmax + j] = Mn[(size_t)i * kmax + j]; }
    x.set_order(g, kn);
    wave_sync();
}

// CTQMC.F90:3615-3889 StepAdd4
template <int MAT>
__device__ W2DYN_INLINE void step_add4(Ctx &x)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    x.count(C_TRYADD4);
    // pair_OperAdd with four operators (Trace.F90:3657-3698): annihilator, creator, annihilator, creator
    double ta_[2], tc_[2]; int fl[2]; double tdf = 1.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const double width1 = x.wmax - x.wmin;
        ta_[p] = x.wmin + width1 * x.draw();
        const int orb = x.randint(1, C.nd), spin = x.randint(1, 2);
        fl[p] = 2 * (orb - 1) + (spin - 1);
        const double tmin = fmax(ta_[p] - C.taudiff_max, x.wmin);
        const double width2 = fmin(ta_[p] + C.taudiff_max, x.wmax) - tmin;
        tc_[p] = tmin + width2 * x.draw();
        tdf *= width1; tdf *= width2;
    }
    const bool same = fl[0] == fl[1];
    if (x.npairs < 0) x.npairs = build_window(x);
    const int n_old = x.npairs;
    const double t4[4] = {tc_[0], ta_[0], tc_[1], ta_[1]};
    const int ca4[4] = {1, 2, 1, 2};
    bool equal = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) equal |= t4[i] == t4[j];
    if constexpr (MAT) {
        for (int base = 0; base < x.nops; base += 64) {
            const int e = base + x.lane; const double t = e < x.nops ? x.ot[e] : -1.0;
            equal |= __ballot(t == t4[0] || t == t4[1] || t == t4[2] || t == t4[3]) != 0ull;
        }
    }
    if (equal) return;
    // removable pairs with the first pair only, the second only, both
    const int d1 = xm_dnp(x, fl[0], tc_[0], ta_[0]), d2 = xm_dnp(x, fl[1], tc_[1], ta_[1]);
    const bool x12 = same && pair_ok(tc_[0], ta_[1], x.wmin, x.wmax, C.taudiff_max), x21 = same && pair_ok(tc_[1], ta_[0], x.wmin, x.wmax, C.taudiff_max);
    const int n12 = n_old + d1, n34 = n_old + d2, n_all = n_old + d1 + d2 + (x12 ? 1 : 0) + (x21 ? 1 : 0);
    int n14 = 0, n23 = 0;
    if (x12 && x21) {
        const int nx12 = n_old + xm_dnp(x, fl[0], tc_[0], ta_[1]), nx21 = n_old + xm_dnp(x, fl[0], tc_[1], ta_[0]);
        const bool c1_first = tc_[0] < ta_[0];             // the earlier operator of the first pair is its creator: its mixed pair is (c1, a2)
        n14 = c1_first ? nx12 : nx21; n23 = c1_first ? nx21 : nx12;
    }
    const double rpf = xm_rempair_factor(n_all, n12, n34, x12 && x21, n14, n23, tdf);

    // quantum numbers and local trace
    double dlogtr = 0.0, tl_new = 0.0, ts_new = 1.0;
    if constexpr (MAT) {
        if (x.nops + 4 > C.wcap) { x.err |= ERR_WCAP; return; }
        int idx[4] = {0, 1, 2, 3};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3 - a; ++b) if (t4[idx[b]] > t4[idx[b + 1]]) { const int q = idx[b]; idx[b] = idx[b + 1]; idx[b + 1] = q; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double t = t4[idx[i]];
            mat_insert(x, mat_lower(x, t), t, mat_code(fl[idx[i] >> 1], ca4[idx[i]]));
        }
        bool ok = mat_walk_into(x, x.outer, x.tsst);
        if (ok) { mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new); dlogtr = tl_new - x.tr_log; }
        if (!ok) {
#pragma unroll
            for (int i = 3; i >= 0; --i) mat_remove(x, mat_lower(x, t4[idx[i]]));
            return;
        }
    } else {
        if (!same) {
            if (!xm_pair_fits(x, fl[0], tc_[0], ta_[0]) || !xm_pair_fits(x, fl[1], tc_[1], ta_[1])) return;
            dlogtr = xm_dlog_two(x, fl[0], fmin(tc_[0], ta_[0]), fmax(tc_[0], ta_[0]), tc_[0] < ta_[0] ? 1.0 : -1.0,
                                 fl[1], fmin(tc_[1], ta_[1]), fmax(tc_[1], ta_[1]), tc_[1] < ta_[1] ? 1.0 : -1.0);
        } else {
            double s[4]; int first[2];
            if (!xm_four_fit(x, fl[0], t4, ca4, true, s, first)) return;
            dlogtr = -((first[0] == 1 ? 1.0 : -1.0) * level_integral<0>(x, fl[0], s[0], s[1]) + (first[1] == 1 ? 1.0 : -1.0) * level_integral<0>(x, fl[0], s[2], s[3]));
        }
    }
    // new operator arrays of the flavours touched (scratch), new inverse matrices, determinant ratio
    x.wvalid = 0;
    const int nfl = same ? 1 : 2;
    int kn[2];
    for (int p = 0; p < nfl; ++p) {
        const int g = fl[p], ko = x.order(g);
        double *tcn = C.TS + ((size_t)x.w * C.nflav * 2 + 2 * g) * kmax, *tan = tcn + kmax;
        for (int e = x.lane; e < ko; e += 64) { tcn[e] = x.tc(g)[e]; tan[e] = x.ta(g)[e]; }
        if (x.lane == 0) {
            tcn[ko] = tc_[p]; tan[ko] = ta_[p];
            if (same) { tcn[ko + 1] = tc_[1]; tan[ko + 1] = ta_[1]; }
        }
        kn[p] = ko + (same ? 2 : 1);
    }
    wave_sync();
    double dlogdet = 0.0, dsgn = 1.0;
    for (int p = 0; p < nfl; ++p) xm_rebuild(x, fl[p], kn[p], dlogdet, dsgn);
    const double rnd = x.draw();
    double prob = fabs(tdf * (double)((2 * C.nd) * (2 * C.nd)) * rpf * exp(dlogtr + dlogdet));
    if constexpr (MAT) { if (ts_new == 0.0) prob = 0.0; }
    if ((x.err & ERR_SINGULAR) || !(rnd < prob)) {
        if constexpr (MAT) {
            int idx[4] = {0, 1, 2, 3};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3 - a; ++b) if (t4[idx[b]] > t4[idx[b + 1]]) { const int q = idx[b]; idx[b] = idx[b + 1]; idx[b + 1] = q; }
#pragma unroll
            for (int i = 3; i >= 0; --i) mat_remove(x, mat_lower(x, t4[idx[i]]));
        }
        return;
    }
    for (int p = 0; p < nfl; ++p) { xm_install(x, fl[p], kn[p]); if (kn[p] >= kmax - 1) x.err |= ERR_KMAX; }
    x.det_log += dlogdet; x.det_sign *= dsgn;
    x.npairs = n_all;
    if constexpr (MAT) { x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); x.pvalid = 0; x.kval = 0; }
    x.count(C_ACCADD4);
}

// CTQMC.F90:4153-4430 StepRem4
template <int MAT>
__device__ W2DYN_INLINE void step_rem4(Ctx &x)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    x.count(C_TRYREM4);
    const int n_old = build_window(x);                 // the list in LDS must be the present one (accepted pair moves only keep its register copy current)
    x.npairs = n_old;
    if (n_old <= 0) return;
    const int none[4] = {-1, -1, -1, -1};
    int e[4];                                          // list entries of the two pairs, each in time order
    xm_pairs(x, none, x.randint(1, n_old), e[0], e[1]);
    const int ex1[4] = {e[0], e[1], -1, -1};
    const int n34 = xm_pairs(x, ex1, 0, e[2], e[3]);
    if (n34 <= 0) return;
    xm_pairs(x, ex1, x.randint(1, n34), e[2], e[3]);
    double t[4]; int m[4], ca[4], fl4[4], slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { t[i] = x.wt[e[i]]; m[i] = x.wm[e[i]]; fl4[i] = m[i] & 0xff; ca[i] = (m[i] >> 8) & 0xff; slot[i] = m[i] >> 16; }
    const int f1 = fl4[0], f2 = fl4[2]; const bool same = f1 == f2;
    const double W = x.wmax - x.wmin;
    const double ta1 = ca[0] == 2 ? t[0] : t[1], ta2 = ca[2] == 2 ? t[2] : t[3];
    const double hf1 = (double)n_old / (W * (fmin(ta1 + C.taudiff_max, x.wmax) - fmax(ta1 - C.taudiff_max, x.wmin)));
    const double hf2 = (double)n34 / (W * (fmin(ta2 + C.taudiff_max, x.wmax) - fmax(ta2 - C.taudiff_max, x.wmin)));
    double tdf = (double)n_old * (double)n34 / hf1 / hf2;                          // CTQMC.F90:4195-4196
    const int ex2[4] = {e[2], e[3], -1, -1};
    int du0, du1;
    const int n12 = xm_pairs(x, ex2, 0, du0, du1);
    // the mixed pairs: the operator of the second pair whose kind differs from that of the first pair's earlier operator
    const int px = ca[2] != ca[0] ? 2 : 3, py = 5 - px;             // partner of e[0], partner of e[1]
    const bool crossed = same && pair_ok(t[0], t[px], x.wmin, x.wmax, C.taudiff_max) && pair_ok(t[1], t[py], x.wmin, x.wmax, C.taudiff_max);
    int n14 = 0, n23 = 0;
    if (crossed) {
        const int exa[4] = {e[0], e[px], -1, -1}, exb[4] = {e[1], e[py], -1, -1};
        n23 = xm_pairs(x, exa, 0, du0, du1); n14 = xm_pairs(x, exb, 0, du0, du1);
    }
    const double rpf = xm_rempair_factor(n_old, n12, n34, crossed, n14, n23, tdf);

    double dlogtr = 0.0, tl_new = 0.0, ts_new = 1.0;
    int li[4] = {0, 0, 0, 0}, lc[4] = {0, 0, 0, 0};    // MAT: list indices (ascending) and codes of the four operators
    double lt[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (MAT) {
        int idx[4] = {0, 1, 2, 3};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3 - a; ++b) if (t[idx[b]] > t[idx[b + 1]]) { const int q = idx[b]; idx[b] = idx[b + 1]; idx[b + 1] = q; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { lt[i] = t[idx[i]]; lc[i] = mat_code(fl4[idx[i]], ca[idx[i]]); li[i] = mat_lower(x, lt[i]); }
#pragma unroll
        for (int i = 3; i >= 0; --i) mat_remove(x, li[i]);
        wave_sync();
        const bool ok = mat_walk_into(x, x.outer, x.tsst);
        if (ok) { mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new); dlogtr = tl_new - x.tr_log; }
        if (!ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mat_insert(x, li[i], lt[i], lc[i]);
            return;
        }
    } else {
        if (!same) {
            // each pair on its own: nothing of its flavour strictly between its operators
            bool blocked = false;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int f = fl4[2 * p], k = x.order(f); const double a = t[2 * p], b = t[2 * p + 1];
                for (int base = 0; base < k; base += 64) {
                    const int q = base + x.lane; const bool v = q < k;
                    const double tce = v ? x.tc(f)[q] : -1.0, tae = v ? x.ta(f)[q] : -1.0;
                    blocked |= __ballot((tce > a && tce < b) || (tae > a && tae < b)) != 0ull;
                }
            }
            if (blocked) return;
            dlogtr = xm_dlog_two(x, f1, t[0], t[1], ca[0] == 1 ? -1.0 : 1.0, f2, t[2], t[3], ca[2] == 1 ? -1.0 : 1.0);
        } else {
            double s[4]; int first[2];
            if (!xm_four_fit(x, f1, t, ca, false, s, first)) return;
            dlogtr = -((first[0] == 1 ? -1.0 : 1.0) * level_integral<0>(x, f1, s[0], s[1]) + (first[1] == 1 ? -1.0 : 1.0) * level_integral<0>(x, f1, s[2], s[3]));
        }
    }
    // operator arrays without the four, new inverse matrices
    const int nfl = same ? 1 : 2;
    int kn[2], fls[2] = {f1, f2};
    for (int p = 0; p < nfl; ++p) {
        const int g = fls[p], ko = x.order(g);
        // slots of the creators / annihilators of flavour g that go
        int sc0 = -1, sc1 = -1, sa0 = -1, sa1 = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (fl4[i] == g) { if (ca[i] == 1) { if (sc0 < 0) sc0 = slot[i]; else sc1 = slot[i]; } else { if (sa0 < 0) sa0 = slot[i]; else sa1 = slot[i]; } }
        double *tcn = C.TS + ((size_t)x.w * C.nflav * 2 + 2 * g) * kmax, *tan = tcn + kmax;
        for (int base = 0; base < ko; base += 64) {
            const int q = base + x.lane; const bool v = q < ko;
            const bool kc = v && q != sc0 && q != sc1, ka = v && q != sa0 && q != sa1;
            // compact in slot order: position = number of kept slots below
            const int pc = (q > sc0 && sc0 >= 0 ? 1 : 0) + (q > sc1 && sc1 >= 0 ? 1 : 0), pa = (q > sa0 && sa0 >= 0 ? 1 : 0) + (q > sa1 && sa1 >= 0 ? 1 : 0);
            if (kc) tcn[q - pc] = x.tc(g)[q];
            if (ka) tan[q - pa] = x.ta(g)[q];
        }
        kn[p] = ko - (same ? 2 : 1);
    }
    wave_sync();
    double dlogdet = 0.0, dsgn = 1.0;
    for (int p = 0; p < nfl; ++p) xm_rebuild(x, fls[p], kn[p], dlogdet, dsgn);
    double prob = fabs(1.0 / (double)((2 * C.nd) * (2 * C.nd)) / tdf / rpf * exp(dlogtr + dlogdet));
    if constexpr (MAT) { if (ts_new == 0.0) prob = 0.0; }
    const double rnd = x.draw();
    if ((x.err & ERR_SINGULAR) || !(rnd < prob)) {
        if constexpr (MAT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mat_insert(x, li[i], lt[i], lc[i]);
        }
        return;
    }
    for (int p = 0; p < nfl; ++p) xm_install(x, fls[p], kn[p]);
    x.det_log += dlogdet; x.det_sign *= dsgn;
    x.npairs = -1;
    if constexpr (MAT) { x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); x.pvalid = 0; x.kval = 0; }
    x.count(C_ACCREM4);
}

