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
// scale factor (kept as a log), renormalised every few operators.  Blocks of 9..16 states (5-orbital Kanamori: 10) take
// the outer states in batches of 64 / NSM.
#pragma once

template <bool OD = false>
__device__ W2DYN_INLINE void build_and_invert(Ctx &x, int f, int k, const double *tcp, const double *tap, double *A, double &logdet, double &sgn,
                                              const int *fcp = nullptr, const int *fap = nullptr);
__device__ W2DYN_INLINE double sort_parity(Ctx &x, const double *t, int k);

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

__device__ W2DYN_INLINE void mat_insert(Ctx &x, int pos, double tau, int code)
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

__device__ W2DYN_INLINE void mat_remove(Ctx &x, int pos)
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
__device__ W2DYN_INLINE void mat_build_list(Ctx &x)
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

// block in front of every operator for outer block `outer`, written to dst[0..nops]; false if the sequence violates
// the quantum numbers
__device__ W2DYN_INLINE bool mat_walk_into(Ctx &x, int outer, int *dst)
{
    const int nf2 = 2 * x.C.nflav;
    int s = outer;
    for (int i = 0; i < x.nops; ++i) {
        if (x.lane == 0) dst[i] = s;
        if (s >= 0) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
    }
    if (x.lane == 0) dst[x.nops] = s;
    wave_sync();
    return s == outer;
}
__device__ __forceinline__ bool mat_walk_sst(Ctx &x) { return mat_walk_into(x, x.outer, x.osst); }
__device__ __forceinline__ void mat_commit_sst(Ctx &x)      // osst <- tsst after an accepted move
{
    for (int e = x.lane; e <= x.nops; e += 64) x.osst[e] = x.tsst[e];
    x.kval = min(x.kval, x.k_i1);            // checkpoints of the window part: valid up to the first operator that changed
    wave_sync();
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
    const bool act = x.lane < x.spb_ * nsm;
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
__device__ W2DYN_INLINE void mat_trace(Ctx &x, int outer, int skip1, int skip2, double &tlog, double &tsign)
{
    const DevConst &C = x.C;
    const int nsm = C.nsm;
    const int nout = uni(x.t_n[outer]);
    tlog = -HUGE_VAL; tsign = 0.0;
    double sum = 0.0, lref = 0.0;                   // trace = sum * exp(lref), summed over the batches of outer states
    bool any = false;
    for (int s0 = 0; s0 < nout; s0 += x.spb_) {
        const bool diag = x.lane < x.spb_ * nsm && s0 + x.ms_ == x.mi_ && s0 + x.ms_ < nout;
        double v = diag ? 1.0 : 0.0, lacc = 0.0, tprev = 0.0;
        int sst = outer, cnt = 0;
        bool dead = false;
        for (int i = 0; i < x.nops; ++i) {
            if (i == skip1 || i == skip2) continue;
            const int code = uni(x.om[i]); const double tau = uni(x.ot[i]);
            sst = mat_apply(x, sst, code, tau - tprev, v, lacc);
            if (sst < 0) return;                     // the quantum-number violation is the same for every batch
            tprev = tau;
            if ((++cnt & 3) == 0) { if (!mat_rescale(v, lacc)) { dead = true; break; } }
        }
        if (dead) { if (x.spb_ >= nout) return; continue; }
        if (sst != outer) return;
        mat_evolve(x, sst, C.beta - tprev, v, lacc);
        const double tot = wave_sum(diag ? v : 0.0);
        if (tot == 0.0 || !(fabs(tot) > 0.0)) continue;
        if (!any) { sum = tot; lref = lacc; any = true; }
        else if (lacc > lref) { sum = sum * exp(lref - lacc) + tot; lref = lacc; }
        else sum += tot * exp(lacc - lref);
    }
    if (!any || sum == 0.0 || !(fabs(sum) > 0.0)) return;
    tlog = lref + log(fabs(sum)); tsign = sum < 0.0 ? -1.0 : 1.0;
}

// ---------------------------------------------------------------------------------------------------------------
// Tree evaluation of the trace for NSM <= 4: lane = operator.  With the block in front of every operator known
// (tsst, maintained incrementally), every operator's transfer matrix  T = Op * diag(exp(-(E - E0) dt))  is built
// independently, lanes multiply their own short run of matrices and six shuffle steps combine the 64 partial
// products:  trace = sum_i [ D_end T_{n-1} ... T_1 T_0 ]_ii.  Magnitudes are carried as (matrix scaled by a power
// of two, integer exponent) per lane plus the sum of the scalar ground-level factors.
// Entry j of the trial sequence is list entry j with skip1 < skip2 left out (removal proposals); entry n is the final
// free evolution to beta.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mat_old_index(int j, int skip1, int skip2)
{
    int o = j;
    if (skip1 >= 0 && o >= skip1) ++o;
    if (skip2 >= 0 && o >= skip2) ++o;
    return o;
}

template <int NSM>
__device__ __forceinline__ void mat_matmul(const double (&A)[NSM * NSM], const double (&B)[NSM * NSM], double (&Cm)[NSM * NSM])
{
#pragma unroll
    for (int i = 0; i < NSM; ++i)
#pragma unroll
        for (int j = 0; j < NSM; ++j) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < NSM; ++k) a = fma(A[i * NSM + k], B[k * NSM + j], a);
            Cm[i * NSM + j] = a;
        }
}

template <int NSM>
__device__ __forceinline__ void mat_normalise(double (&P)[NSM * NSM], int &ex)
{
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < NSM * NSM; ++i) m = fmax(m, fabs(P[i]));
    if (m > 0.0) {
        int e = ilogb(m);
#pragma unroll
        for (int i = 0; i < NSM * NSM; ++i) P[i] = ldexp(P[i], -e);
        ex += e;
    }
}

// transfer matrix of trial entry j (j == n: final evolution to tend with the identity as operator).  tbeg is the time
// the evolution in front of entry `first` starts from (0 for the full trace).
template <int NSM>
__device__ __forceinline__ void mat_transfer(Ctx &x, const int *tsst, int j, int n, int skip1, int skip2, int first, double tbeg,
                                             double tend, double (&T)[NSM * NSM], double &lsum, int &bad)
{
    const DevConst &C = x.C;
    const int nf2 = 2 * C.nflav;
    int sst = tsst[j];
    int o = mat_old_index(j, skip1, skip2);
    // a block index or list position outside its table would address memory outside the wave's LDS slice: flag it
    // (reported by the host as an internal error) and continue with safe values instead of faulting
    if ((unsigned)sst >= (unsigned)C.nsst || (unsigned)o > (unsigned)C.wcap || (j != n && (unsigned)x.om[o < C.wcap ? o : 0] >= (unsigned)nf2)) {
        bad = 1; sst = 0; o = 0;
    }
    const double tprev = (j == first) ? tbeg : x.ot[mat_old_index(j - 1, skip1, skip2)];
    const double tau = (j == n) ? tend : x.ot[o];
    const double dt = tau - tprev;
    const int nst = x.t_n[sst], off = x.t_off[sst];
    const double e0 = x.t_eval[off];
    double d[NSM];
#pragma unroll
    for (int k = 0; k < NSM; ++k) d[k] = (k < nst) ? ((k == 0) ? 1.0 : exp(-(x.t_eval[off + k] - e0) * dt)) : 0.0;
    lsum -= (e0 - C.egs) * dt;
    if (j == n) {
#pragma unroll
        for (int i = 0; i < NSM; ++i)
#pragma unroll
            for (int k = 0; k < NSM; ++k) T[i * NSM + k] = (i == k) ? d[k] : 0.0;
    } else {
        const int nsm = C.nsm;                         // the tables are padded to nsm x nsm (nsm <= NSM)
        const size_t base = (size_t)(sst * nf2 + x.om[o]) * nsm * nsm;
#pragma unroll
        for (int i = 0; i < NSM; ++i)
#pragma unroll
            for (int k = 0; k < NSM; ++k) {
                double op = 0.0;
                if (i < nsm && k < nsm) op = x.t_op ? x.t_op[base + i * nsm + k] : C.opmat[base + i * nsm + k];
                T[i * NSM + k] = op * d[k];
            }
    }
}

// product of the transfer matrices of trial entries [first, last] (last may be n = final evolution), left in lane 0:
// P (scaled), binary exponent ex, and the wave-uniform scalar log factor lsum
template <int NSM>
__device__ W2DYN_INLINE void mat_tree_product(Ctx &x, const int *tsst, int n, int skip1, int skip2, int first, int last, double tbeg, double tend,
                                 double (&P)[NSM * NSM], int &ex, double &lsum)
{
    const int cnt = last - first + 1;
    const int per = (cnt + 63) >> 6;
    ex = 0; lsum = 0.0;
#pragma unroll
    for (int i = 0; i < NSM * NSM; ++i) P[i] = ((i / NSM) == (i % NSM)) ? 1.0 : 0.0;
    const int j0 = first + x.lane * per, j1 = min(j0 + per, last + 1);
    int bad = 0;
    for (int j = j0; j < j1; ++j) {
        double T[NSM * NSM], R[NSM * NSM];
        mat_transfer<NSM>(x, tsst, j, n, skip1, skip2, first, tbeg, tend, T, lsum, bad);
        mat_matmul<NSM>(T, P, R);
#pragma unroll
        for (int i = 0; i < NSM * NSM; ++i) P[i] = R[i];
        if (((j - j0) & 3) == 3) mat_normalise<NSM>(P, ex);
    }
    mat_normalise<NSM>(P, ex);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        double Q[NSM * NSM], R[NSM * NSM];
#pragma unroll
        for (int i = 0; i < NSM * NSM; ++i) Q[i] = __shfl_down(P[i], s, 64);
        const int eq = __shfl_down(ex, s, 64);
        if ((x.lane & (2 * s - 1)) == 0 && x.lane + s < 64) {
            mat_matmul<NSM>(Q, P, R);                  // later operators act from the left
#pragma unroll
            for (int i = 0; i < NSM * NSM; ++i) P[i] = R[i];
            ex += eq;
            mat_normalise<NSM>(P, ex);
        }
    }
    lsum = wave_sum(lsum);
    if (__ballot(bad != 0) != 0ull) x.err |= ERR_BLOCKSEQ;
}

template <int NSM>
__device__ W2DYN_INLINE void mat_trace_tree(Ctx &x, const int *tsst, int n, int skip1, int skip2, double &tlog, double &tsign)
{
    double P[NSM * NSM]; int ex; double lsum;
    mat_tree_product<NSM>(x, tsst, n, skip1, skip2, 0, n, 0.0, x.C.beta, P, ex, lsum);
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < NSM; ++i) tr += P[i * NSM + i];
    tr = uni(tr); ex = uni(ex);
    tlog = -HUGE_VAL; tsign = 0.0;
    if (tr == 0.0 || !(fabs(tr) > 0.0)) return;
    tlog = lsum + (double)ex * 0.6931471805599453 + log(fabs(tr)); tsign = tr < 0.0 ? -1.0 : 1.0;
}

// Sequential evaluation of the trace of a trial sequence whose block sequence tsst is known (blocks of 5..16 states):
// lane = (outer state of the batch, component), all batches of outer states advance together on one common scale.
// A wave issues at most one vector instruction every four clocks whatever the occupancy, so the cost of one operator is
// its instruction count, and the loop over the operators holds nothing but the product itself:
//   * everything that is the same for all lanes (time step, block, scalar decay, position of the matrix) is worked out for
//     64 operators at once, lane jj for operator base + jj;
//   * the decay factors exp(-(E_i - E_0) dt) of the 64 operators are computed NT per lane and step into an LDS table (one
//     exp per operator and state instead of one per operator and lane);
//   * every lane reads ITS row of the operator matrix straight from global memory (L2; the table is padded to NT x NT by
//     the host, so a row is NT / 2 aligned 16-byte loads from a scalar base), four operators ahead into registers: no
//     staging through LDS, no index arithmetic per element;
//   * the vectors are exchanged through a zero-padded LDS tile: rows of vectors that do not exist read as zero and rows of
//     the matrix beyond the target block are zero, so no lane needs a mask inside the product.
constexpr int MAT_NB = 4, MAT_AHEAD = 4, MAT_TILE_ROWS = 16;
// checkpoints of the window part (mat_trace_seq_window): entry size in doubles, entries per chain (window operators / 2 at
// most 128, plus two scratch entries that take the stores nobody wants: 2 * 65 >= 128 doubles, one slot per lane and kind)
__host__ __device__ inline int mat_ck_doubles(int nt) { return nt * nt + 1; }
constexpr int MAT_CK_ENTRIES = 131;

// doubles of LDS the evaluator needs per wave: vector tile [MAT_TILE_ROWS][NT], then the decay table [64][NT]
__host__ __device__ inline int mat_seq_lds_doubles(int nt) { return (MAT_TILE_ROWS + 64) * nt; }

// Everything the evaluator needs, by value: LDS locations as byte offsets into the dynamic LDS (so that the out-of-line
// function below addresses LDS as LDS, not through generic pointers) and plain numbers.  The function is a REAL call on
// purpose: inlined into the sweep kernel its loop shares the register file with the whole step logic and the
// allocator spills loop-carried values to scratch (measured: ~5000 cycles per operator, 0.5 M cycles per trace); as a
// call the caller's state is saved once per trace and the loop runs out of registers.
struct MatSeqArgs {
    unsigned tsst, om, ot, conn, tn, toff, teval, opbuf;           // byte offsets
    unsigned ptout;                                                // 0xffffffff or byte offset of nsm + 1 doubles: partial traces of the outer states, then their common log scale
    unsigned pin, pout;                                            // 0xffffffff or byte offset of NT * NT + 1 doubles: vectors to start from / to leave the result in ([vector][component], stride NT, then the log scale)
    const double *opmat;                                           // [nsst][nf2][NT][NT] (DevConst::opmat_pad)
    // checkpoints in global memory (mat_trace_seq_window): gin = vectors to start from (same format as pin) or null;
    // gout = where the vectors after the 2nd, 4th, ... operator of the stretch go (entries of MAT_CK_DOUBLES(NT) doubles,
    // the first nstore / 2 of them are kept, the rest lands in the scratch entry gdump); STORE instances only
    const double *gin; double *gout, *gdump; int nstore;
    int n, skip1, skip2, nsm, spb, nf2, nsst;
    // the stretch of the trial sequence to apply: cnt operators from trial index jbeg on (indices beyond n - 1 wrap around
    // to 0, their times count from beta), between the times t0 and t1 (same unwrapped scale), from block sst_start to
    // block sst_end, on ncols vectors.  The whole trace: jbeg 0, cnt n, t0 0, t1 beta, both blocks the outer one,
    // ncols its states, unit start vectors, no result vectors (the sum of the diagonal is returned).
    int jbeg, cnt, sst_start, sst_end, ncols;
    double t0, t1;
    double beta, egs;
};

// padded block size the evaluator is instantiated for (operator table, vector tile and vector exchange use this stride)
__host__ __device__ inline int mat_seq_pad(int nsm) { return nsm <= 8 ? 8 : (nsm <= 10 ? 10 : 16); }

// What actually travels to the out-of-line evaluator: the same, packed into 29 dwords (LDS offsets and the small integers
// as 16-bit halves).  How it travels (read off the ISA of tu_seq, round 3; `tools/vgpr.sh tu_seq [-DW2DYN_SEQ_UNPACKED]`):
// clang passes an aggregate of more than 16 dwords INDIRECTLY whatever its size - the caller writes a copy to offset 0 of
// its own private segment (scratch_store_dwordx4 off, v[..], off offset:0..104) and hands over the private address in v0;
// the callee reads it back with scratch_load_dwordx4 v[..], v0, off and makes everything wave-uniform with
// v_readfirstlane.  That holds for this structure and for the 42-dword MatSeqArgs alike; in both builds the kernel sets
// s32 to its own frame size (0x500 / 0x660 for k_sweep<9, false>) and its descriptor asks for that plus the deepest callee's
// 12 bytes (private_segment_fixed_size 1292 / 1644, uses_dynamic_stack 0).  So the earlier explanation of the wrong
// trajectories seen once with the unpacked structure - "arguments beyond 32 dwords go on the stack" - was wrong: nothing of
// either structure is passed in argument registers, and the ISA of the two builds differs in frame offsets only.  That
// intermediate state also carried the first, unfinished version of the window checkpoints (same commit), which is where the
// fault must have been; it cannot be rebuilt from the history.  The packing stays because it is cheaper (120 instead of
// 168 bytes written and read per trace); the size check below documents the layout, it protects nothing.
struct MatSeqPacked {
    unsigned h[13];                // halves: 11 LDS offsets in dwords (0xffff: none), then 13 small integers (skip + 1), 2 spare
    const double *opmat, *gin; double *gout, *gdump;
    double t0, t1, beta, egs;
};
static_assert(sizeof(MatSeqPacked) == 120, "layout of the evaluator's packed arguments (29 dwords and one of padding)");
__device__ __forceinline__ MatSeqPacked mat_seq_pack(const MatSeqArgs &a)
{
    MatSeqPacked p;
    auto off = [](unsigned b) { return b == 0xffffffffu ? 0xffffu : (b >> 2); };
    const unsigned v[26] = {off(a.tsst), off(a.om), off(a.ot), off(a.conn), off(a.tn), off(a.toff), off(a.teval), off(a.opbuf), off(a.ptout), off(a.pin), off(a.pout),
                            (unsigned)a.n, (unsigned)(a.skip1 + 1), (unsigned)(a.skip2 + 1), (unsigned)a.nsm, (unsigned)a.spb, (unsigned)a.nf2, (unsigned)a.nsst,
                            (unsigned)a.jbeg, (unsigned)a.cnt, (unsigned)a.sst_start, (unsigned)a.sst_end, (unsigned)a.ncols, (unsigned)a.nstore, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 13; ++i) p.h[i] = (v[2 * i] & 0xffffu) | (v[2 * i + 1] << 16);
    p.opmat = a.opmat; p.gin = a.gin; p.gout = a.gout; p.gdump = a.gdump;
    p.t0 = a.t0; p.t1 = a.t1; p.beta = a.beta; p.egs = a.egs;
    return p;
}
__device__ __forceinline__ MatSeqArgs mat_seq_unpack(const MatSeqPacked &p)
{
    MatSeqArgs a;
    auto half = [&](int i) { const unsigned w = (unsigned)uni((int)p.h[i >> 1]); return (i & 1) ? (w >> 16) : (w & 0xffffu); };
    auto off = [&](int i) { const unsigned v = half(i); return v == 0xffffu ? 0xffffffffu : (v << 2); };
    a.tsst = off(0); a.om = off(1); a.ot = off(2); a.conn = off(3); a.tn = off(4); a.toff = off(5); a.teval = off(6); a.opbuf = off(7);
    a.ptout = off(8); a.pin = off(9); a.pout = off(10);
    a.n = (int)half(11); a.skip1 = (int)half(12) - 1; a.skip2 = (int)half(13) - 1; a.nsm = (int)half(14); a.spb = (int)half(15); a.nf2 = (int)half(16);
    a.nsst = (int)half(17); a.jbeg = (int)half(18); a.cnt = (int)half(19);
    a.sst_start = (int)(short)half(20); a.sst_end = (int)(short)half(21); a.ncols = (int)half(22); a.nstore = (int)half(23);
    a.opmat = p.opmat; a.gin = p.gin; a.gout = p.gout; a.gdump = p.gdump;
    a.t0 = p.t0; a.t1 = p.t1; a.beta = p.beta; a.egs = p.egs;
    return a;
}

typedef double mat_d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) mat_d2 *MatRowPtr;

// NB: batches of vectors in use, ceil(ncols / spb) (compile-time, so that the loop has no branches at all: a taken branch costs
// a single wave more than the dozen instructions it skips)
#ifdef W2DYN_SEQ_UNPACKED      // ISA study only (tools/vgpr.sh tu_seq -DW2DYN_SEQ_UNPACKED): the 42-dword structure by value
template <int NT, int NB, bool STORE = false>
__device__ __noinline__ double2 mat_trace_seq_core(MatSeqArgs a)
{
#else
template <int NT, int NB, bool STORE = false>
__device__ __noinline__ double2 mat_trace_seq_core(MatSeqPacked packed)
{
    const MatSeqArgs a = mat_seq_unpack(packed);
#endif
    extern __shared__ unsigned char lds_base[];
    const int lane = (int)__lane_id();
    const int *tsst = (const int *)(lds_base + uni((int)a.tsst)), *om = (const int *)(lds_base + uni((int)a.om));
    const double *ot = (const double *)(lds_base + uni((int)a.ot));
    const short *t_conn = (const short *)(lds_base + uni((int)a.conn));
    const int *t_n = (const int *)(lds_base + uni((int)a.tn)), *t_off = (const int *)(lds_base + uni((int)a.toff));
    const double *t_eval = (const double *)(lds_base + uni((int)a.teval));
    double *vbuf = (double *)(lds_base + uni((int)a.opbuf)), *fdt = vbuf + MAT_TILE_ROWS * NT;
    const unsigned long long ops_bits = ((unsigned long long)(unsigned)uni((int)((unsigned long long)a.opmat >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)(unsigned long long)a.opmat);
    const int n = uni(a.n), skip1 = uni(a.skip1), skip2 = uni(a.skip2), nsm = uni(a.nsm), spb = uni(a.spb), nf2 = uni(a.nf2), nsst = uni(a.nsst);
    const int jbeg = uni(a.jbeg), cnt = uni(a.cnt), sst_start = uni(a.sst_start), sst_end = uni(a.sst_end), ncols = uni(a.ncols);
    const double beta = uni(a.beta), egs = uni(a.egs), t0 = uni(a.t0), t1 = uni(a.t1);
    const bool has_pin = uni((int)a.pin) != -1, has_pout = uni((int)a.pout) != -1;
    typedef const __attribute__((address_space(1))) double *GInPtr;
    typedef __attribute__((address_space(1))) double *GOutPtr;
    auto uni_ptr = [](const void *p) { const unsigned long long b = (unsigned long long)p; return ((unsigned long long)(unsigned)uni((int)(b >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)b); };
    const GInPtr gin = (GInPtr)uni_ptr(a.gin);
    const unsigned long long gout_b = uni_ptr(a.gout), gdump_b = uni_ptr(a.gdump);
    const int nstore = uni(a.nstore);
    double2 res = make_double2(-HUGE_VAL, 0.0);
    if ((unsigned)sst_start >= (unsigned)nsst || (unsigned)sst_end >= (unsigned)nsst) return res;
    if (cnt > 0 && uni(tsst[jbeg >= n ? jbeg - n : jbeg]) != sst_start) return res;
    if (cnt == 0 && sst_start != sst_end) return res;
    // lanes beyond the last full vector repeat lane 0's arithmetic and never write
    const bool act = lane < spb * nsm;
    const int ms = act ? lane / nsm : 0, mi = act ? lane - ms * nsm : 0;
    bool diag[NB], wr[NB]; double v[NB];
    double lacc = 0.0;
    {
        const double *pin = (const double *)(lds_base + (has_pin ? uni((int)a.pin) : 0));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int st = b * spb + ms;
            const bool inr = st < ncols;
            wr[b] = act && inr;
            diag[b] = wr[b] && st == mi;
            if (STORE && gin) v[b] = inr ? gin[st * NT + mi] : 0.0;
            else v[b] = has_pin ? (inr ? pin[st * NT + mi] : 0.0) : ((inr && st == mi) ? 1.0 : 0.0);
        }
        if (STORE && gin) lacc = uni(gin[NT * NT]);
        else if (has_pin) lacc = uni(pin[NT * NT]);
    }
    lds_sync();
    for (int e = lane; e < MAT_TILE_ROWS * NT; e += 64) vbuf[e] = 0.0;      // what no lane writes below stays zero
    const MatRowPtr myrow = (MatRowPtr)(ops_bits + (unsigned long long)(mi * NT) * sizeof(double));
    mat_d2 R[MAT_AHEAD][NT / 2];                    // the lane's rows of the next MAT_AHEAD operators
    auto trial_index = [&](int q) { const int j = jbeg + q; return j >= n ? j - n : j; };
    double tlast = t0;
    int iexp = 0;                                   // common binary exponent taken out of the vectors
    for (int base = 0; base < cnt; base += 64) {
        const int qm = base + lane;
        double my_dt = 0.0, my_sc = 0.0; int my_on = 0, my_op = nsst * nf2; bool my_bad = false;      // beyond the stretch: the unit matrix the host appends to the table
        if (qm < cnt) {
            const int jm = trial_index(qm), sst = tsst[jm], o = mat_old_index(jm, skip1, skip2);
            const double tau = ot[o] + (jbeg + qm >= n ? beta : 0.0);
            double tp = t0;
            if (qm > 0) { const int jp = trial_index(qm - 1); tp = ot[mat_old_index(jp, skip1, skip2)] + (jbeg + qm - 1 >= n ? beta : 0.0); }
            my_dt = tau - tp;
            if ((unsigned)sst >= (unsigned)nsst) my_bad = true;
            else {
                const int code = om[o];
                my_bad = (unsigned)code >= (unsigned)nf2 || t_conn[sst * nf2 + code] < 0;
                const int off = t_off[sst];
                my_on = off | (t_n[sst] << 16);
                if (!my_bad) my_op = sst * nf2 + code;
                my_sc = (t_eval[off] - egs) * my_dt;
            }
            if (qm == cnt - 1) tlast = tau;
        }
        if (__ballot(my_bad) != 0ull) return res;
        const double lacc_g0 = lacc;
        double sc_pre = my_sc;                      // STORE: lane jj = scalar decay of the group's operators up to and including jj
        if constexpr (STORE) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(sc_pre, o, 64); if (lane >= o) sc_pre += t; }
        }
        lacc -= wave_sum(my_sc);
        // The group is filled up to a multiple of MAT_AHEAD with unit operators (decay factor one, unit matrix: the vectors
        // pass unchanged, exactly), so that the loop below has no branch around its loads: the compiler's count of loads in
        // flight stays exact and a product waits for ITS rows only, not for the ones requested after them.
        const int cntg = min(64, cnt - base), cntp = (cntg + MAT_AHEAD - 1) / MAT_AHEAD * MAT_AHEAD;
#pragma unroll
        for (int u = 0; u < MAT_AHEAD; ++u) {        // rows of the first operators of the group: in flight while the decay table is made
            const MatRowPtr p = myrow + (size_t)__builtin_amdgcn_readlane(my_op, u) * (NT * NT / 2);
#pragma unroll
            for (int k = 0; k < NT / 2; ++k) R[u][k] = p[k];
        }
        lds_sync();                                 // the table of the group before is not read any more
        for (int e0 = 0; e0 < cntp * NT; e0 += 64) {      // all lanes stay in the loop: the values are fetched from other lanes
            const int e = e0 + lane, jj = e / NT, i = e - jj * NT;
            const double dtj = __shfl(my_dt, jj & 63, 64);
            const int onj = __shfl(my_on, jj & 63, 64), nj = onj >> 16, off = onj & 0xffff;
            const double f = (jj < cntg && i < nj) ? exp(-(t_eval[off + i] - t_eval[off]) * dtj) : 0.0;
            if (e < cntp * NT) fdt[e] = jj >= cntg ? 1.0 : f;
        }
        lds_sync();
        double fd = fdt[mi];
        for (int jj0 = 0; jj0 < cntp; jj0 += MAT_AHEAD) {
#pragma unroll
            for (int u = 0; u < MAT_AHEAD; ++u) {
                const int jj = jj0 + u;
                // vectors that do not exist are zero and stay zero; lanes beyond the last full vector write what lane 0 writes
#pragma unroll
                for (int b = 0; b < NB; ++b) vbuf[(b * spb + ms) * NT + mi] = v[b] * fd;
                lds_sync();
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const double *vr = vbuf + (b * spb + ms) * NT;
                    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                    for (int k = 0; k < NT / 2; ++k) { acc0 = fma(R[u][k].x, vr[2 * k], acc0); acc1 = fma(R[u][k].y, vr[2 * k + 1], acc1); }
                    v[b] = acc0 + acc1;
                }
                fd = fdt[(jj + 1) * NT + mi];        // the next operator's factor travels with the vector reads (beyond the group: never used)
                lds_sync();
                if constexpr (STORE) {
                    // checkpoint after every second operator: all lanes store, always (a store under a branch or a lane mask
                    // would blur the compiler's count of accesses in flight); what nobody wants goes to the scratch entry
                    if ((u & 1) == 1) {        // constant after unrolling
                        const int qa = base + jj;                                  // this was operator qa of the stretch
                        const bool keep = qa < nstore && jj < cntg;
                        const unsigned long long dst = keep ? gout_b + (unsigned long long)(qa >> 1) * (NT * NT + 1) * sizeof(double) : gdump_b;
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const unsigned long long at = (keep && wr[b]) ? dst + (unsigned long long)((b * spb + ms) * NT + mi) * sizeof(double)
                                                                           : gdump_b + (unsigned long long)lane * sizeof(double);
                            *(GOutPtr)at = v[b];
                        }
                        const double lsc = lacc_g0 - readlane_d(sc_pre, jj & 63) + (double)iexp * 0.6931471805599453;
                        const unsigned long long at = (keep && lane == 0) ? dst + (unsigned long long)(NT * NT) * sizeof(double)
                                                                          : gdump_b + (unsigned long long)(65 + lane) * sizeof(double);
                        *(GOutPtr)at = lsc;
                    }
                }
                {   // the operator MAT_AHEAD further on (beyond the group: a row that is never used)
                    const MatRowPtr p = myrow + (size_t)__builtin_amdgcn_readlane(my_op, (jj + MAT_AHEAD) & 63) * (NT * NT / 2);
#pragma unroll
                    for (int k = 0; k < NT / 2; ++k) R[u][k] = p[k];
                }
            }
            {   // keep the vectors near one: take out a power of two (exact)
                double m = 0.0;
#pragma unroll
                for (int b = 0; b < NB; ++b) m = fmax(m, fabs(v[b]));
                m = wave_max(m);
                if (!(m > 0.0)) return res;
                const int e2 = ilogb(m);
#pragma unroll
                for (int b = 0; b < NB; ++b) v[b] = ldexp(v[b], -e2);
                iexp += e2;
            }
        }
    }
    const double tprev = cnt > 0 ? uni(wave_max(tlast)) : t0;
    lacc += (double)iexp * 0.6931471805599453;
    {   // free evolution to t1 inside the last block
        const int off = uni(t_off[sst_end]), nend = uni(t_n[sst_end]);
        const double e0 = uni(t_eval[off]), dt = t1 - tprev;
        lacc -= (e0 - egs) * dt;
        const double fdec = (nend > 1 && mi < nend) ? exp(-(t_eval[off + mi] - e0) * dt) : 1.0;
        if (has_pout) {                              // the vectors themselves are the result
            double *pout = (double *)(lds_base + uni((int)a.pout));
            double m = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) { v[b] = (wr[b] && mi < nend) ? v[b] * fdec : 0.0; m = fmax(m, fabs(v[b])); }
            m = wave_max(m);
            if (!(m > 0.0)) return res;
            const int e2 = ilogb(m);
            for (int e = lane; e < NT * NT; e += 64) pout[e] = 0.0;
            lds_sync();
#pragma unroll
            for (int b = 0; b < NB; ++b) if (wr[b]) pout[(b * spb + ms) * NT + mi] = ldexp(v[b], -e2);
            if (lane == 0) pout[NT * NT] = lacc + (double)e2 * 0.6931471805599453;
            lds_sync();
            res.x = 0.0; res.y = 1.0;
            return res;
        }
        double part = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) part += diag[b] ? v[b] * fdec : 0.0;
        if (uni((int)a.ptout) != -1) {               // estimator pass only: hand the partial traces to the caller
            double *pto = (double *)(lds_base + uni((int)a.ptout));
            lds_sync();
#pragma unroll
            for (int b = 0; b < NB; ++b) if (diag[b]) pto[b * spb + ms] = v[b] * fdec;
            if (lane == 0) pto[nsm] = lacc;
            lds_sync();
        }
        const double tot = wave_sum(part);
        if (tot == 0.0 || !(fabs(tot) > 0.0)) return res;
        res.x = lacc + log(fabs(tot)); res.y = tot < 0.0 ? -1.0 : 1.0;
    }
    return res;
}

__device__ __forceinline__ MatSeqArgs mat_seq_args(Ctx &x, const int *tsst, int n, int skip1, int skip2)
{
    extern __shared__ unsigned char lds_base[];
    const DevConst &C = x.C;
    auto off = [&](const void *p) { return (unsigned)((const unsigned char *)p - lds_base); };
    MatSeqArgs a;
    a.tsst = off(tsst); a.om = off(x.om); a.ot = off(x.ot); a.conn = off(x.t_conn); a.tn = off(x.t_n); a.toff = off(x.t_off);
    a.teval = off(x.t_eval); a.opbuf = off(x.opbuf);
    a.ptout = a.pin = a.pout = 0xffffffffu;
    a.gin = nullptr; a.gout = a.gdump = nullptr; a.nstore = 0;
    a.opmat = C.opmat_pad;
    a.n = n; a.skip1 = skip1; a.skip2 = skip2; a.nsm = C.nsm; a.spb = x.spb_; a.nf2 = 2 * C.nflav; a.nsst = C.nsst;
    a.beta = C.beta; a.egs = C.egs;
    return a;
}
// The choice among the instances is made inline at the call site.  A second out-of-line level (kernel -> dispatcher ->
// evaluator, the dispatcher taking the arguments by reference, i.e. as a generic pointer into the kernel's private
// segment) hung the GPU in one diagnostic build of round 2 and is not used: one call level, arguments by value.  The cause
// of that hang is NOT established (the build no longer exists; it is not to be re-created on a shared machine).
template <bool STORE = false>
__device__ __forceinline__ double2 mat_seq_dispatch(const MatSeqArgs &args, int nt)
{
    const int nb = uni((args.ncols + args.spb - 1) / args.spb);
#ifdef W2DYN_SEQ_UNPACKED
    const MatSeqArgs &a = args;
#else
    const MatSeqPacked a = mat_seq_pack(args);
#endif
    if (nt == 8) return mat_trace_seq_core<8, 1, STORE>(a);                  // up to 8 states: all vectors in one batch
    if (nt == 10) return nb <= 1 ? mat_trace_seq_core<10, 1, STORE>(a) : mat_trace_seq_core<10, 2, STORE>(a);
    return nb <= 1 ? mat_trace_seq_core<16, 1, STORE>(a) : (nb == 2 ? mat_trace_seq_core<16, 2, STORE>(a) : (nb == 3 ? mat_trace_seq_core<16, 3, STORE>(a) : mat_trace_seq_core<16, 4, STORE>(a)));
}
template <bool STORE = false>
__device__ __forceinline__ double2 mat_seq_call(const DevConst &C, const MatSeqArgs &a) { return mat_seq_dispatch<STORE>(a, mat_seq_pad(C.nsm)); }

__device__ __forceinline__ void mat_trace_seq(Ctx &x, const int *tsst, int n, int skip1, int skip2, double &tlog, double &tsign, double *ptout = nullptr)
{
    extern __shared__ unsigned char lds_base[];
    const DevConst &C = x.C;
    MatSeqArgs a = mat_seq_args(x, tsst, n, skip1, skip2);
    a.ptout = ptout ? (unsigned)((const unsigned char *)ptout - lds_base) : 0xffffffffu;
    wave_sync();
    const int outer = uni(tsst[0]);
    tlog = -HUGE_VAL; tsign = 0.0;
    if ((unsigned)outer >= (unsigned)C.nsst || uni(tsst[n]) != outer) return;
    a.jbeg = 0; a.cnt = n; a.sst_start = outer; a.sst_end = outer; a.ncols = uni(x.t_n[outer]); a.t0 = 0.0; a.t1 = C.beta;
    const double2 r = mat_seq_call(C, a);
    tlog = uni(r.x); tsign = uni(r.y);
}

// The same trace with everything outside the sliding window taken from a cache.  A pair move changes operators inside
// [wmin, wmax] only, so the product of all operators from wmax over beta = 0 to wmin,
//     P = (ops before wmin) (ops after wmax) : block at wmax -> block at wmin,
// stays what it is until the window moves or a global move, a tau shift or a change of the outer block is accepted
// (x.pvalid).  The trial trace is then  Tr[ (ops inside the window) P ] : the stored vectors P are propagated through the
// operators inside the window only (the reference keeps bra / ket vectors outside the window for the same reason,
// Trace.F90:5384-5522, 6109-6209).  Z-space sweep only: worm operators may sit anywhere.
// CAUTION: a -DW2DYN_STAMPS -DW2DYN_STAMPS_SEQ build of the state with window checkpoints faulted the GPU (the production
// build of the same source passes every parity test); the numbers quoted in DESIGN.md come from the state before them.  Do not
// run such a build on a shared machine without finding out why first.
#ifdef W2DYN_STAMPS_SEQ     // diagnostic build: counts and cycles of the cached-window evaluation in the DBG slots
#ifndef W2DYN_STAMPS_SEQ_ISA_ONLY
#error "W2DYN_STAMPS_SEQ: a build of this kind faulted the GPU in round 2 and the cause is not known (see CAUTION above); compile it to ISA only (-DW2DYN_STAMPS_SEQ_ISA_ONLY with hipcc -S), do not run it"
#endif
#define SEQ_COUNT(x, c, v) do { if ((x).lane == 0) (x).cnt[c] += (unsigned long long)(v); } while (0)
#define SEQ_T0() const unsigned long long tq0_ = __builtin_amdgcn_s_memtime()
#define SEQ_T1(x, c) do { if ((x).lane == 0) (x).cnt[c] += __builtin_amdgcn_s_memtime() - tq0_; } while (0)
#else
#define SEQ_COUNT(x, c, v)
#define SEQ_T0()
#define SEQ_T1(x, c)
#endif
__device__ W2DYN_INLINE void mat_trace_seq_window(Ctx &x, const int *tsst, int n, int skip1, int skip2, double &tlog, double &tsign)
{
    extern __shared__ unsigned char lds_base[];
    const DevConst &C = x.C;
    tlog = -HUGE_VAL; tsign = 0.0;
    wave_sync();
    const int outer = uni(tsst[0]);
    if ((unsigned)outer >= (unsigned)C.nsst || uni(tsst[n]) != outer) return;
    // trial indices of the window: [jlo, jhi) - the operators left out (skip1, skip2) lie inside the window
    int clo = 0, chi = 0;
    const int nstored = n + (skip1 >= 0 ? 1 : 0) + (skip2 >= 0 ? 1 : 0);
    for (int base = 0; base < nstored; base += 64) {
        const int e = base + x.lane; const bool in = e < nstored;
        const double t = in ? x.ot[e] : 0.0;
        clo += count_pred(in && t < x.wmin); chi += count_pred(in && t <= x.wmax);
    }
    const int jlo = clo, jhi = chi - (nstored - n);
    SEQ_COUNT(x, C_DBG0, 1); SEQ_COUNT(x, C_DBG7, n);
    if (jhi < jlo || (skip1 >= 0 && (skip1 < clo || skip1 >= chi)) || (skip2 >= 0 && (skip2 < clo || skip2 >= chi))) {
        SEQ_COUNT(x, C_DBG6, 1);
        mat_trace_seq(x, tsst, n, skip1, skip2, tlog, tsign);       // not a window-local change: the plain evaluation
        return;
    }
    const int sst_lo = uni(tsst[jlo]), sst_hi = uni(tsst[jhi]);     // tsst[n] = outer covers jhi = n
    if ((unsigned)sst_lo >= (unsigned)C.nsst || (unsigned)sst_hi >= (unsigned)C.nsst) return;
    MatSeqArgs a = mat_seq_args(x, tsst, n, skip1, skip2);
    const unsigned poff = (unsigned)((const unsigned char *)x.pmat - lds_base);
    if (!x.pvalid || x.p_lo != sst_lo || x.p_hi != sst_hi) {
        // operators jhi .. n-1, then 0 .. jlo-1, from wmax to beta + wmin, unit vectors on the block at wmax
        a.jbeg = jhi; a.cnt = (n - jhi) + jlo; a.sst_start = sst_hi; a.sst_end = sst_lo; a.ncols = uni(x.t_n[sst_hi]);
        a.t0 = x.wmax; a.t1 = C.beta + x.wmin; a.pout = poff;
        SEQ_COUNT(x, C_DBG1, 1); SEQ_COUNT(x, C_DBG2, a.cnt); SEQ_T0();
        const double2 r = mat_seq_call(C, a);
        SEQ_T1(x, C_DBG4);
        x.kval = 0;
        if (uni(r.y) == 0.0) { x.pvalid = 0; return; }
        x.pvalid = 1; x.p_lo = sst_lo; x.p_hi = sst_hi;
        a.pout = 0xffffffffu;
    }
    // Checkpoints inside the window (global memory, per chain): the vectors after every second window operator of the
    // CURRENT configuration.  The operators in front of the first one a proposal changes (x.k_first) are the current ones,
    // so the walk starts at the last checkpoint not beyond it and leaves new checkpoints up to it on the way; an accepted
    // move cuts the valid checkpoints back to its first operator (mat_commit_sst), a new outside product drops them all.
    const int i1 = max(min(x.k_first, jhi) - jlo, 0) & ~1;       // window operators (even) known to be unchanged
    const int s = x.kc ? min(i1, x.kval) : 0;
    x.k_i1 = i1;
    const int stride = mat_ck_doubles(mat_seq_pad(C.nsm));
    a.jbeg = jlo + s; a.cnt = (jhi - jlo) - s; a.sst_start = s ? uni(tsst[jlo + s]) : sst_lo; a.sst_end = sst_hi; a.ncols = uni(x.t_n[sst_hi]);
    a.t0 = s ? uni(x.ot[jlo + s - 1]) : x.wmin; a.t1 = x.wmax; a.pin = poff;
    SEQ_COUNT(x, C_DBG3, a.cnt); SEQ_T0();
    double2 r;
    if (x.kc) {
        a.gin = s ? x.kc + (size_t)(s / 2 - 1) * stride : nullptr;
        a.gout = x.kc + (size_t)(s / 2) * stride; a.gdump = x.kc + (size_t)(MAT_CK_ENTRIES - 3) * stride; a.nstore = i1 - s;
        r = mat_seq_call<true>(C, a);
        x.kval = uni(r.y) != 0.0 ? max(x.kval, i1) : s;
    } else r = mat_seq_call(C, a);
    SEQ_T1(x, C_DBG5);
    tlog = uni(r.x); tsign = uni(r.y);
}

// trace of the trial sequence whose block sequence is in tsst: tree for small blocks (MATN = 2, 3, 4 = padded block
// size the kernel was instantiated for), sequential walk otherwise (MATN = 9)
template <int MATN, bool WINDOW = false>
__device__ W2DYN_INLINE void mat_trace_trial(Ctx &x, const int *tsst, int n, int skip1, int skip2, double &tlog, double &tsign)
{
    if constexpr (MATN >= 2 && MATN <= 4) mat_trace_tree<MATN>(x, tsst, n, skip1, skip2, tlog, tsign);
    else if constexpr (WINDOW) mat_trace_seq_window(x, tsst, n, skip1, skip2, tlog, tsign);
    else mat_trace_seq(x, tsst, n, skip1, skip2, tlog, tsign);
}

// Density matrix at beta/2 (MeasDensityMatrix_beta_half, CTQMC.F90:3088-3151) from two tree products:
// K = evolution to beta/2 times the operators with tau <= beta/2, A = the rest up to beta.  Summed over the outer
// states s, bra_s(i1) ket_s(i2) = A(s,i1) K(i2,s), i.e. rho(i1,i2) += fac * (K A)(i2,i1).
template <int NSM>
__device__ W2DYN_INLINE void mat_dm_tree(Ctx &x, double cfgsign, int bin)
{
    const DevConst &C = x.C;
    const double b2 = 0.5 * C.beta;
    const int nk = mat_lower_eq(x, b2);
    double K[NSM * NSM], A[NSM * NSM], R[NSM * NSM]; int exk, exa; double lk, la;
    mat_tree_product<NSM>(x, x.osst, nk, -1, -1, 0, nk, 0.0, b2, K, exk, lk);
    mat_tree_product<NSM>(x, x.osst, x.nops, -1, -1, nk, x.nops, b2, C.beta, A, exa, la);
    if (x.lane == 0) {
        mat_matmul<NSM>(K, A, R);
        const int sb = x.osst[nk], n = x.t_n[sb];
        const double fac = -cfgsign * x.tr_sign * exp(lk + la + (double)(exk + exa) * 0.6931471805599453 - x.tr_log);
        double *dmb = C.acc_dm + (size_t)bin * C.dmsize + C.dmoff[sb];
#pragma unroll
        for (int i1 = 0; i1 < NSM; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < NSM; ++i2)
                if (i1 < n && i2 < n) atomicAdd(&dmb[i1 * n + i2], fac * R[i2 * NSM + i1]);
    }
}

// Blocks of up to 4 states: the same two halves as tree products (lane 0 holds K and A), from which the estimator kernel
// takes the trace (sum of the diagonal of A K), the partial traces of the outer states (that diagonal) and the density
// matrix at beta/2 - two tree products per measurement instead of four.  Sets x.tr_log / x.tr_sign; lane s < n(outer)
// receives the partial trace pt on the scale exp(lsc); nk = operators up to beta/2.
template <int NSM>
__device__ W2DYN_INLINE void mat_halves_tree(Ctx &x, double (&K)[NSM * NSM], double (&A)[NSM * NSM], double &pt, double &lsc, int &nk)
{
    const DevConst &C = x.C;
    const double b2 = 0.5 * C.beta;
    pt = 0.0; lsc = 0.0; nk = 0;
    x.tr_log = -HUGE_VAL; x.tr_sign = 0.0;
#pragma unroll
    for (int i = 0; i < NSM * NSM; ++i) { K[i] = 0.0; A[i] = 0.0; }
    if (!(uni(x.osst[x.nops]) == x.outer)) return;
    nk = mat_lower_eq(x, b2);
    int exk, exa; double lk, la;
    mat_tree_product<NSM>(x, x.osst, nk, -1, -1, 0, nk, 0.0, b2, K, exk, lk);
    mat_tree_product<NSM>(x, x.osst, x.nops, -1, -1, nk, x.nops, b2, C.beta, A, exa, la);
    lsc = lk + la + (double)(uni(exk) + uni(exa)) * 0.6931471805599453;
    double tot = 0.0;
#pragma unroll
    for (int sidx = 0; sidx < NSM; ++sidx) {
        double d = 0.0;
#pragma unroll
        for (int i = 0; i < NSM; ++i) d = fma(A[sidx * NSM + i], K[i * NSM + sidx], d);
        d = uni(d);                                           // lane 0 holds the products
        tot += d;
        if (x.lane == sidx) pt = d;
    }
    if (tot == 0.0 || !(fabs(tot) > 0.0)) { pt = 0.0; return; }
    x.tr_log = lsc + log(fabs(tot)); x.tr_sign = tot < 0.0 ? -1.0 : 1.0;
}

// density matrix at beta/2 from those halves: rho(i1,i2) += fac (K A)(i2,i1), fac = -cfgsign sign(trace) exp(lsc - log|trace|)
template <int NSM>
__device__ W2DYN_INLINE void mat_dm_tree_from(Ctx &x, const double (&K)[NSM * NSM], const double (&A)[NSM * NSM], int nk, double fac, int bin)
{
    const DevConst &C = x.C;
    if (x.lane == 0) {
        double R[NSM * NSM];
        mat_matmul<NSM>(K, A, R);
        const int sb = x.osst[nk], n = x.t_n[sb];
        double *dmb = C.acc_dm + (size_t)bin * C.dmsize + C.dmoff[sb];
#pragma unroll
        for (int i1 = 0; i1 < NSM; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < NSM; ++i2)
                if (i1 < n && i2 < n) atomicAdd(&dmb[i1 * n + i2], fac * R[i2 * NSM + i1]);
    }
}

// trace of the stored configuration (osst must be current)
template <int MATN>
__device__ W2DYN_INLINE void mat_trace_current(Ctx &x)
{
    if (!(uni(x.osst[x.nops]) == x.outer)) { x.tr_log = -HUGE_VAL; x.tr_sign = 0.0; return; }
    mat_trace_trial<MATN>(x, x.osst, x.nops, -1, -1, x.tr_log, x.tr_sign);
}

// Partial traces of the states of the outer block for the stored configuration (parttrace, Trace.F90:6029-6031): lane i < n
// receives <i| ... |i> on the common scale exp(lsc); zero everywhere when the trace vanishes.  Tree path: the diagonal of the
// full product; sequential path: the evaluator leaves them in LDS.
template <int MATN>
__device__ W2DYN_INLINE void mat_part_traces(Ctx &x, double &pt, double &lsc)
{
    pt = 0.0; lsc = 0.0;
    if (!(uni(x.osst[x.nops]) == x.outer)) return;
    if constexpr (MATN >= 2 && MATN <= 4) {
        double P[MATN * MATN]; int ex; double lsum;
        mat_tree_product<MATN>(x, x.osst, x.nops, -1, -1, 0, x.nops, 0.0, x.C.beta, P, ex, lsum);
#pragma unroll
        for (int i = 0; i < MATN; ++i) { const double d = uni(P[i * MATN + i]); if (x.lane == i) pt = d; }
        lsc = lsum + (double)uni(ex) * 0.6931471805599453;
    } else {
        // the evaluator's own operator stage (>= 4 * 64 doubles) takes the nsm + 1 results once the last operator is applied
        double tl, ts;
        double *buf = x.opbuf;
        mat_trace_seq(x, x.osst, x.nops, -1, -1, tl, ts, buf);
        wave_sync();
        if (ts != 0.0) { pt = x.lane < uni(x.t_n[x.outer]) ? buf[x.lane] : 0.0; lsc = uni(buf[x.C.nsm]); }
    }
}

// Blocks of more than 4 states: everything the Z-space estimators need from the propagated states comes out of ONE walk
// over the operators, cut at beta/2 (two stretches of the sequential evaluator):
//   K(i, s) = <i| (operators with tau <= beta/2, evolution to beta/2) |s>     s: states of the outer block, i: block at beta/2
//   A(s, i) = <s| (evolution to beta, operators with tau > beta/2) |i>
// The partial trace of outer state s is sum_i A(s,i) K(i,s) (parttrace, Trace.F90:6029-6031), the trace their sum, and the
// density matrix at beta/2 is rho(i1,i2) = sum_s A(s,i1) K(i2,s) / trace (MeasDensityMatrix_beta_half, CTQMC.F90:3088-3151).
// Leaves A in x.pmat ([i][s], stride NT) and K in x.opbuf ([s][i]), their log scales behind them; sets x.tr_log / x.tr_sign;
// lane s < n(outer) receives the partial trace pt on the scale exp(lsc).  sk: block at beta/2.
__device__ W2DYN_INLINE void mat_halves(Ctx &x, double &pt, double &lsc, int &sk)
{
    extern __shared__ unsigned char lds_base[];
    const DevConst &C = x.C;
    const int nt = mat_seq_pad(C.nsm), n = x.nops;
    pt = 0.0; lsc = 0.0; sk = x.outer;
    x.tr_log = -HUGE_VAL; x.tr_sign = 0.0;
    wave_sync();
    if (!(uni(x.osst[n]) == x.outer)) return;
    const double b2 = 0.5 * C.beta;
    const int nk = mat_lower_eq(x, b2);
    sk = uni(x.osst[nk]);                                  // osst[n] = outer covers nk = n
    if ((unsigned)sk >= (unsigned)C.nsst) return;
    const int nout = uni(x.t_n[x.outer]), nmid = uni(x.t_n[sk]);
    MatSeqArgs a = mat_seq_args(x, x.osst, n, -1, -1);
    a.pout = (unsigned)((const unsigned char *)x.pmat - lds_base);
    a.jbeg = 0; a.cnt = nk; a.sst_start = x.outer; a.sst_end = sk; a.ncols = nout; a.t0 = 0.0; a.t1 = b2;
    double2 r = mat_seq_call(C, a);
    if (uni(r.y) == 0.0) return;
    wave_sync();
    double keep[5];                                        // NT * NT + 1 <= 257 doubles
#pragma unroll
    for (int q = 0; q < 5; ++q) { const int e = q * 64 + x.lane; keep[q] = e <= nt * nt ? x.pmat[e] : 0.0; }
    wave_sync();
    a.jbeg = nk; a.cnt = n - nk; a.sst_start = sk; a.sst_end = x.outer; a.ncols = nmid; a.t0 = b2; a.t1 = C.beta;
    r = mat_seq_call(C, a);
    if (uni(r.y) == 0.0) return;
    wave_sync();
#pragma unroll
    for (int q = 0; q < 5; ++q) { const int e = q * 64 + x.lane; if (e <= nt * nt) x.opbuf[e] = keep[q]; }     // the evaluator's own space is free again
    wave_sync();
    const double *A = x.pmat, *K = x.opbuf;
    if (x.lane < nout) for (int i = 0; i < nmid; ++i) pt = fma(A[i * nt + x.lane], K[x.lane * nt + i], pt);
    lsc = uni(A[nt * nt]) + uni(K[nt * nt]);
    const double tot = wave_sum(pt);
    if (tot == 0.0 || !(fabs(tot) > 0.0)) { pt = 0.0; return; }
    x.tr_log = lsc + log(fabs(tot)); x.tr_sign = tot < 0.0 ? -1.0 : 1.0;
}

// density matrix at beta/2 from the two halves left by mat_halves: fac = -cfgsign * sign(trace) / |sum of the partial traces|
__device__ W2DYN_INLINE void mat_dm_from_halves(Ctx &x, int sk, double fac, int bin)
{
    const DevConst &C = x.C;
    const int nt = mat_seq_pad(C.nsm), nout = uni(x.t_n[x.outer]), n = uni(x.t_n[sk]);
    const double *A = x.pmat, *K = x.opbuf;
    double *dmb = C.acc_dm + (size_t)bin * C.dmsize + C.dmoff[sk];
    for (int e = x.lane; e < n * n; e += 64) {             // rho(i1 = bra index, i2 = ket index)
        const int i1 = e / n, i2 = e - i1 * n;
        double acc = 0.0;
        for (int s = 0; s < nout; ++s) acc = fma(A[i1 * nt + s], K[s * nt + i2], acc);
        atomicAdd(&dmb[i1 * n + i2], fac * acc);
    }
}

// check_qn_all_ssts (Trace.F90:1402-1433): number of outer blocks for which the operator sequence is allowed;
// with select > 0 returns instead the index of the select-th allowed block.
__device__ W2DYN_INLINE int mat_possible(Ctx &x, int select)
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
template <int MATN>
__device__ W2DYN_INLINE void step_glob_mat(Ctx &x)
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
    else if (r < 0.8 && C.nsym > 0) {                // gen_SymUpdate, Trace.F90:1170-1182
        const int isym = x.randint(1, 2 * C.nsym) - 1;
        gu[0] = 0; for (int i = 1; i <= nf; ++i) gu[i] = C.symmoves[isym * nf + i - 1];
    } else {
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
    mat_walk_into(x, outer_new, x.tsst);
    mat_trace_trial<MATN>(x, x.tsst, x.nops, -1, -1, tl, ts);
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
    mat_commit_sst(x);
    x.npairs = -1; x.pvalid = 0;
    x.count(C_ACCGLOB);
}
