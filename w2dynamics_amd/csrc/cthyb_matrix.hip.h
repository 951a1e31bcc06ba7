// Matrix local-trace path: blocks of H_loc with more than one state (Kanamori / Coulomb interactions, or
// density-density interactions with coarser quantum numbers).  Reference: get_Trace_EB_fullmatrix
// (Trace.F90:5734-6033), check_sst_sequence (1243-1302), check_qn_all_ssts (1402-1433), update_b2_states_eb
// (6213-6470).  Included by cthyb_kernels.hip.h inside namespace w2dyn, after Ctx.
//
// Data kept per chain in the wave's LDS slice (in addition to the slot-ordered tau arrays that the M updates use):
//   ot[0..nops)   operator times, ascending          om[0..nops)  operator code = 2*flavour + (ca-1)
//   osst[0..nops] block ("superstate") in front of operator i; osst[nops] = block after the last one (= outer)
// Tables of the local problem (block sizes, offsets, eigenvalues, Connect, operators in the eigenbasis padded to
// NSM x NSM, row = target state) are staged once per workgroup in LDS and shared by its four chains.
//
// Trace evaluation: the state vectors of ALL outer states are propagated together, lane = (outer state s, component i)
// with NSM*NSM <= 64 lanes in use; one operator application is NSM ds_bpermute + FMA per lane.  The vectors share one
// scale factor (kept as a log), renormalised every few operators.
#pragma once

__device__ void build_and_invert(Ctx &x, int f, int k, const double *tcp, const double *tap, double *A, double &logdet, double &sgn);
__device__ double sort_parity(Ctx &x, const double *t, int k);

// code of an operator: 2*flavour + (ca - 1), ca 1 = creator, 2 = annihilator
__device__ __forceinline__ int mat_code(int f, int ca) { return 2 * f + (ca - 1); }

// number of operators with tau < t (the list is sorted: this is the insertion index of t)
__device__ __forceinline__ int mat_lower(Ctx &x, double t)
{
    int c = 0;
    for (int base = 0; base < x.nops; base += 64) { int e = base + x.lane; c += count_pred(e < x.nops && x.ot[e] < t); }
    return c;
}
__device__ __forceinline__ int mat_lower_eq(Ctx &x, double t)
{
    int c = 0;
    for (int base = 0; base < x.nops; base += 64) { int e = base + x.lane; c += count_pred(e < x.nops && x.ot[e] <= t); }
    return c;
}

__device__ void mat_insert(Ctx &x, int pos, double tau, int code)
{
    for (int base = (x.nops - 1) & ~63; base >= 0 && base + 63 >= pos; base -= 64) {
        int e = base + x.lane;
        bool mv = e >= pos && e < x.nops;
        double t = mv ? x.ot[e] : 0.0; int m = mv ? x.om[e] : 0;
        wave_sync();
        if (mv) { x.ot[e + 1] = t; x.om[e + 1] = m; }
        wave_sync();
    }
    if (x.lane == 0) { x.ot[pos] = tau; x.om[pos] = code; }
    x.nops += 1;
    wave_sync();
}

__device__ void mat_remove(Ctx &x, int pos)
{
    for (int base = pos & ~63; base < x.nops; base += 64) {
        int e = base + x.lane;
        bool mv = e >= pos && e + 1 < x.nops;
        double t = mv ? x.ot[e + 1] : 0.0; int m = mv ? x.om[e + 1] : 0;
        wave_sync();
        if (mv) { x.ot[e] = t; x.om[e] = m; }
        wave_sync();
    }
    x.nops -= 1;
}

// time-ordered list from the slot-ordered tau arrays (rank of every operator by counting)
__device__ void mat_build_list(Ctx &x)
{
    const DevConst &C = x.C;
    int n = 0;
    for (int f = 0; f < C.nflav; ++f) n += 2 * x.order(f);
    x.nops = n;
    if (n > C.wcap) { x.err |= ERR_WCAP; x.nops = 0; return; }
    wave_sync();
    for (int f = 0; f < C.nflav; ++f) {
        const int k = x.order(f);
        for (int e = x.lane; e < 2 * k; e += 64) {
            const int ca = e < k ? 1 : 2;
            const double tau = e < k ? x.tc(f)[e] : x.ta(f)[e - k];
            int rank = 0;
            for (int g = 0; g < C.nflav; ++g) {
                const int kg = x.order(g);
                const double *cg = x.tc(g), *ag = x.ta(g);
                for (int j = 0; j < kg; ++j) rank += (cg[j] < tau ? 1 : 0) + (ag[j] < tau ? 1 : 0);
            }
            x.ot[rank] = tau; x.om[rank] = mat_code(f, ca);
        }
    }
    wave_sync();
}

// block in front of every operator for outer block x.outer; false if the sequence violates the quantum numbers
__device__ bool mat_walk_sst(Ctx &x)
{
    const int nf2 = 2 * x.C.nflav;
    int s = x.outer;
    bool ok = true;
    for (int i = 0; i < x.nops; ++i) {
        if (x.lane == 0) x.osst[i] = s;
        if (s >= 0) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
    }
    if (x.lane == 0) x.osst[x.nops] = s;
    ok = s == x.outer;
    wave_sync();
    return ok;
}

// One propagation step of all outer-state vectors: exp(-dt H) inside block sst, then operator `code`
// (Trace.F90:5802-5829).  Returns the target block (-1: forbidden).  v is the lane's component, lacc the common log scale.
__device__ __forceinline__ int mat_apply(Ctx &x, int sst, int code, double dt, double &v, double &lacc)
{
    const DevConst &C = x.C;
    const int nsm = C.nsm, nf2 = 2 * C.nflav;
    const int nsst = uni(x.t_conn[sst * nf2 + code]);
    if (nsst < 0) return -1;
    const int n = uni(x.t_n[sst]), off = uni(x.t_off[sst]);
    const double e0 = uni(x.t_eval[off]);
    if (n > 1) {
        const double ei = (x.mi_ < n) ? x.t_eval[off + x.mi_] : e0;
        v *= exp(-(ei - e0) * dt);
    }
    lacc -= (e0 - C.egs) * dt;
    const size_t rowoff = ((size_t)(sst * nf2 + code) * nsm + x.mi_) * nsm;
    const bool act = x.lane < nsm * nsm;
    double nv = 0.0;
    for (int j = 0; j < n; ++j) {
        const double vj = __shfl(v, x.ms_ * nsm + j, 64);
        const double o = act ? (x.t_op ? x.t_op[rowoff + j] : C.opmat[rowoff + j]) : 0.0;
        nv = fma(o, vj, nv);
    }
    v = nv;
    return nsst;
}

__device__ __forceinline__ bool mat_rescale(double &v, double &lacc)
{
    const double m = wave_max(fabs(v));
    if (!(m > 0.0)) return false;
    v /= m; lacc += log(m);
    return true;
}

// free evolution of the vectors inside block sst over dt (no operator)
__device__ __forceinline__ void mat_evolve(Ctx &x, int sst, double dt, double &v, double &lacc)
{
    const int n = uni(x.t_n[sst]), off = uni(x.t_off[sst]);
    const double e0 = uni(x.t_eval[off]);
    if (n > 1) {
        const double ei = (x.mi_ < n) ? x.t_eval[off + x.mi_] : e0;
        v *= exp(-(ei - e0) * dt);
    }
    lacc -= (e0 - x.C.egs) * dt;
}

// Local trace of the operator list (entries skip1, skip2 left out) summed over the states of block `outer`.
// tsign = 0 (and tlog = -inf) when the sequence is forbidden or the trace vanishes.
__device__ void mat_trace(Ctx &x, int outer, int skip1, int skip2, double &tlog, double &tsign)
{
    const DevConst &C = x.C;
    const int nsm = C.nsm;
    const int nout = uni(x.t_n[outer]);
    const bool diag = x.lane < nsm * nsm && x.ms_ == x.mi_ && x.ms_ < nout;
    double v = diag ? 1.0 : 0.0, lacc = 0.0, tprev = 0.0;
    int sst = outer, cnt = 0;
    tlog = -HUGE_VAL; tsign = 0.0;
    for (int i = 0; i < x.nops; ++i) {
        if (i == skip1 || i == skip2) continue;
        const int code = uni(x.om[i]); const double tau = uni(x.ot[i]);
        sst = mat_apply(x, sst, code, tau - tprev, v, lacc);
        if (sst < 0) return;
        tprev = tau;
        if ((++cnt & 3) == 0) { if (!mat_rescale(v, lacc)) return; }
    }
    if (sst != outer) return;
    mat_evolve(x, sst, C.beta - tprev, v, lacc);
    const double tot = wave_sum(diag ? v : 0.0);
    if (tot == 0.0 || !(fabs(tot) > 0.0)) return;
    tlog = lacc + log(fabs(tot)); tsign = tot < 0.0 ? -1.0 : 1.0;
}

// check_qn_all_ssts (Trace.F90:1402-1433): number of outer blocks for which the operator sequence is allowed;
// with select > 0 returns instead the index of the select-th allowed block.
__device__ int mat_possible(Ctx &x, int select)
{
    const DevConst &C = x.C;
    const int nf2 = 2 * C.nflav;
    int total = 0;
    for (int base = 0; base < C.nsst; base += 64) {
        const int s0 = base + x.lane;
        int s = s0 < C.nsst ? s0 : -1;
        for (int i = 0; i < x.nops; ++i) {
            const int code = uni(x.om[i]);
            if (s >= 0) s = x.t_conn[s * nf2 + code];
        }
        unsigned long long m = __ballot(s >= 0 && s == s0);
        const int c = __popcll(m);
        if (select > 0 && select <= total + c) {
            int k = select - total;
            while (--k > 0) m &= m - 1ull;
            return base + __ffsll((long long)m) - 1;
        }
        total += c;
    }
    return total;
}

// global move for the matrix path: CTQMC.F90:4433-4554 with perform_spinflip_check_qn / perform_global_check_qn
// (Trace.F90:1306-1394).  The times stay where they are, so the sorted list keeps its order and only the codes change.
__device__ void step_glob_mat(Ctx &x)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd, kmax = C.kmax;
    x.count(C_TRYGLOB);
    if (x.nops == 0) return;
    x.wvalid = 0;
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
    double *TS = C.TS + (size_t)x.w * nf * 2 * kmax;
    int Kn[MAXFLAV];
    int *lut = x.wm;                                   // code -> relabelled code
    for (int f = 0; f < nf; ++f) {
        int num = (f >> 1) + nd * (f & 1) + 1;
        int t = gu[num];
        int g = 2 * ((t - 1) % nd) + (t - 1) / nd;
        int k = x.order(f);
        Kn[g] = k;
        if (x.lane == 0) { lut[2 * f] = 2 * g + (gu[0] ? 1 : 0); lut[2 * f + 1] = 2 * g + (gu[0] ? 0 : 1); }
        const double *srcc = gu[0] ? x.ta(f) : x.tc(f), *srca = gu[0] ? x.tc(f) : x.ta(f);
        for (int e = x.lane; e < k; e += 64) { TS[(size_t)(2 * g) * kmax + e] = srcc[e]; TS[(size_t)(2 * g + 1) * kmax + e] = srca[e]; }
    }
    wave_sync();
    int outer_new = x.outer;
    double sst_factor = 1.0;
    const bool single = spinflip && C.has_szt;
    int pold = 0;
    if (!single) pold = mat_possible(x, 0);
    for (int e = x.lane; e < x.nops; e += 64) { int c = x.om[e]; x.wc[e] = c; x.om[e] = lut[c]; }   // backup + relabel
    wave_sync();
    if (single) {
        int st = C.sst_first[x.outer];
        int down = st & ((1 << nd) - 1), up = (st >> nd) & ((1 << nd) - 1);
        outer_new = C.sst_of_state[up | (down << nd)];
        const int nf2 = 2 * nf;
        int s = outer_new;
        for (int i = 0; i < x.nops && s >= 0; ++i) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
        if (s != outer_new) sst_factor = 0.0;
    } else {
        int pnew = mat_possible(x, 0);
        if (pnew <= 0) sst_factor = 0.0;
        else {
            sst_factor = (double)pnew / (double)pold;
            int kk = x.randint(1, pnew);
            outer_new = mat_possible(x, kk);
        }
    }
    if (sst_factor == 0.0) {
        for (int e = x.lane; e < x.nops; e += 64) x.om[e] = x.wc[e];
        wave_sync();
        x.count(C_QNGLOB); return;
    }
    // the inversion below uses wm as pivot storage: the code backup lives in wc, the lut is no longer needed
    double *MBw = C.MB + (size_t)x.w * nf * kmax * kmax;
    double logdet_new = 0.0, sign_new = 1.0;
    for (int g = 0; g < nf; ++g) {
        double ld, sg;
        const double *tcg = TS + (size_t)(2 * g) * kmax, *tag = TS + (size_t)(2 * g + 1) * kmax;
        build_and_invert(x, g, Kn[g], tcg, tag, MBw + (size_t)g * kmax * kmax, ld, sg);
        if (Kn[g] > 0) sg *= sort_parity(x, tcg, Kn[g]) * sort_parity(x, tag, Kn[g]);
        logdet_new += ld; sign_new *= sg;
    }
    double tl, ts;
    mat_trace(x, outer_new, -1, -1, tl, ts);
    double rnd = x.draw();
    double ratio = (ts == 0.0) ? 0.0 : fabs(exp((tl - x.tr_log) + (logdet_new - x.det_log)) * sst_factor);
    if ((x.err & ERR_SINGULAR) || !(rnd < ratio)) {
        for (int e = x.lane; e < x.nops; e += 64) x.om[e] = x.wc[e];
        wave_sync();
        return;
    }
    for (int g = 0; g < nf; ++g) {
        int k = Kn[g];
        double *tcp = x.tc(g), *tap = x.ta(g), *Mg = x.mat(g);
        const double *Mn = MBw + (size_t)g * kmax * kmax;
        for (int e = x.lane; e < k; e += 64) { tcp[e] = TS[(size_t)(2 * g) * kmax + e]; tap[e] = TS[(size_t)(2 * g + 1) * kmax + e]; }
        for (int j0 = 0; j0 < k; j0 += 64) { int j = j0 + x.lane; if (j < k) for (int i = 0; i < k; ++i) Mg[(size_t)i * kmax + j] = Mn[(size_t)i * kmax + j]; }
        x.set_order(g, k);
    }
    wave_sync();
    x.outer = outer_new;
    x.det_log = logdet_new; x.det_sign = sign_new;
    x.tr_log = tl; x.tr_sign = ts;
    mat_walk_sst(x);
    x.npairs = -1;
    x.count(C_ACCGLOB);
}
