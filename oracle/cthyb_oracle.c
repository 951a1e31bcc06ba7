/* TEST INFRASTRUCTURE ONLY (oracle) - see cthyb_oracle.h.
 *
 * CPU restatement of the reference CT-HYB hot path for density-density local problems with diagonal
 * hybridisation.  References are to files under /root/reference/src/ctqmc_fortran unless stated otherwise.
 * The reference keeps operators in doubly linked lists and reallocates matrices; here the configuration is one
 * time-sorted array and M blocks are dense row-major arrays (row = creator rank, column = annihilator rank in
 * time order inside the flavour, as `MInv%Mat(creator, annihilator)`, CTQMC.F90:2303-2304).
 *
 * Index conventions:
 *   flavour f = 2*(orb-1) + (spin-1)             (array order (Nd,2) of every result array)
 *   Fock bit of flavour: (orb-1) + Nd*(spin-1)   (States.F90:198-199, 328-334)
 *   global-move flavour number i = orb + Nd*(spin-1)   (Trace.F90:2148-2151)
 *   ca: 1 = creator, 2 = annihilator              (Trace.F90:35-38)
 *   superstate (block) index: order of first appearance of its quantum numbers over the Fock states 0..4^Nd-1
 *   (States.F90:149-184); with "Nt Szt Azt" every block holds one state and the index is the Fock integer
 */
#include "cthyb_oracle.h"
#include "locprob.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXFLAV 16

typedef struct {
    double tau;
    int flav; /* 0-based flavour */
    int ca;   /* 1 creator, 2 annihilator */
    int wid;  /* 0: hybridising operator (has_hyb); 1..4: worm operator wormContainer(wid) (Trace.F90:123-125) */
} op_t;

typedef struct { double log, sign; } logval;

struct ora_walker {
    ora_params p;
    ora_rng *rng;
    int nflav, nstates;
    ora_locprob *lp; int nsst;
    double *ftab;   /* [nflav][nftau] diagonal F(tau) */
    double *E;      /* [nstates] eigenvalues of H_loc in block order (= lp->eval) */
    double Egs;
    logval *ptr;    /* [nsmax] partial traces of the outer states (parttrace, Trace.F90:6118) */
    double *dmblk; int *dmoff; int dmsize;   /* density matrix in the eigenbasis, block b at dmoff[b], [bra][ket] row major */
    /* configuration */
    op_t *ops; int nops, cap;
    double *M[MAXFLAV]; int k[MAXFLAV];
    /* offdiag = 1: ONE matrix over all flavours, rows = creators, columns = annihilators, each ordered by flavour, then time
     * (MInv_full, Trace.F90:2690-2770); F(tau) with both flavour indices [annihilator flavour][creator flavour][nftau] */
    double *Mf; int kt; double *ffull; double *gleg_full, *gtau_full;
    int outer_sst;
    logval det, tr;
    double cfgsign;
    /* window (Trace.F90:1608-1751) */
    double taudiff_max, wmin, wmax; int num_windows, wpos;
    int nslide;
    /* counters */
    /* worm sampling */
    int sector, isector, icomponent;          /* current sector (1 = Z, 4 = G4), sampled worm sector and component */
    double eta;
    double wtry_add, wacc_add, wtry_rem, wacc_rem;                       /* TryAdd/AccAdd/TryRem/AccRem(worm sector) */
    double try_wadd, acc_wadd, accqn_wadd, try_wrem, acc_wrem, accqn_wrem, try_wrep, acc_wrep;
    double cnt_sampling_w, cnt_meas_w;
    double *g4w, *g4wpp;                      /* g4iw_worm, g4iwpp_worm accumulators */
    /* AccumGtau = 2: autocorracc (Accumulators.F90:206-362): per-measurement buffer, (spill, mean, sqdiff) per component and level */
    double *acbuf, *acdat; int aclev; int64_t acn;
    double *g4tau, *g4leg;                    /* FourPnt & 1: G4tau (Nd,2,Nd,2,N4tau^3); FourPnt & 2: G4leg (Nd,2,Nd,2,N4tau,N4tau,2N4iwb+1) complex */
    double *giww, *gtauw;                     /* giw_worm (2Niw complex), gtau_worm (Ntau) accumulators (sector 2) */
    double *gsw;                              /* GSigmaiw_worm (2Niw complex) accumulator (sector 3) */
    double *umat;                             /* U matrix (2Nd)^4, flavour index 2 * orbital + spin, as handed to init_solver */
    double *wmore[11]; size_t wmore_n[11];    /* h4iw, qq, qqtau, qqqq, nqqdag, qqdd, ucaca, ucacatau, uccaa, uccaatau, quddag worm accumulators (sectors 5, 10 - 16) */
    double *weq[6]; size_t weq_n[6];          /* p2iw, p2iwpp, p2tau, p2taupp, p3iw, p3iwpp worm accumulators (sectors 6 - 9) */
    double try_add, acc_add, accqn_add, try_rem, acc_rem, accqn_rem, try_glob, acc_glob, accqn_glob;
    double try_shift, acc_shift, cnt_sampling, cnt_meas;
    double try_flavc, acc_flavc;                      /* TryFlavc, AccFlavc (CTQMC.F90:4857, 4951) */
    double try_add4, acc_add4, try_rem4, acc_rem4;   /* TryAdd4, AccAdd4, TryRem4, AccRem4 (CTQMC.F90:3645, 3861, 4175, 4394) */
    double try_add_outer, acc_add_outer, try_rem_outer, acc_rem_outer;   /* TryAddOuter ... (CTQMC.F90:3385, 3589, 3917, 4135) */
    int64_t *accpair;
    int naccpair; double accpair_max;          /* NAccPair, AccPairMax (CTQMC.F90:68-69): Ntau bins over beta outside the calibration */
    double *accpairtau; int napt, apt_index;   /* AccPairTau list of the window calibration, phase 1 (CTQMC.F90:67, 70) */
    /* accumulators */
    double *gleg, *gtau, *giw, *histo, *dm, *outer_histo;
    double *ntau_n0;                      /* [nflav][nflav][ntau+1], MeasSusz */
    double *gms, *gmid, *sstep;           /* [nflav][nmeas] x2, [nmeas]: Gtau_mean_step, Gtau_mid_step, sign_step */
    /* MeasOuterHistograms (CTQMC.F90:3285-3339): per block [nsst] and per eigenstate [nstates], block order */
    double *sign_sst, *tcontrib_sst, *outer_state_histo, *sign_state, *tcontrib_state;
    double *g2iw, *g4iw, *g4iwpp;    /* complex interleaved, NULL when not measured */
    double meas_sign; int64_t cnt_sign;
    double mean_sign;
    int aborted;
};

/* ------------------------------------------------------------------------------------------------------- */
/* pure pieces                                                                                               */
/* ------------------------------------------------------------------------------------------------------- */

/* Trace.F90:1014-1033 get_HybrF: linear interpolation, antiperiodic continuation for negative arguments */
double ora_hybr(const double *ftab, int nftau, double beta, double dtau)
{
    double sig = 1.0;
    if (dtau < 0.0) { dtau += beta; sig = -1.0; }
    double n = dtau / beta * (double)(nftau - 1);
    int i = (int)n;
    return sig * (ftab[i] + (n - (double)i) * (ftab[i + 1] - ftab[i]));
}

/* MatrixUpdate.F90:392-407: S - sum_ij R_i N_ij Q_j, same accumulation order */
double ora_detrat_add(const double *M, int k, const double *Q, const double *R, double S)
{
    double r = S;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            r = r - R[i] * M[i * k + j] * Q[j];
    return r;
}

/* MatrixUpdate.F90:410-442: new (k+1)x(k+1) inverse after inserting row pos_c / column pos_a (0-based here) */
void ora_mat_add(const double *M, int k, const double *Q, const double *R, double S, int pc, int pa, double *out)
{
    int n = k + 1;
    out[pc * n + pa] = 1.0 / S;
    for (int i = 0; i < k; ++i) {              /* new column: -N.Q/S */
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += M[i * k + j] * Q[j];
        int in = i < pc ? i : i + 1;
        out[in * n + pa] = -s / S;
    }
    for (int j = 0; j < k; ++j) {              /* new row: -R.N/S */
        double s = 0.0;
        for (int i = 0; i < k; ++i) s += R[i] * M[i * k + j];
        int jn = j < pa ? j : j + 1;
        out[pc * n + jn] = -s / S;
    }
    for (int i = 0; i < k; ++i) {
        int in = i < pc ? i : i + 1;
        for (int j = 0; j < k; ++j) {
            int jn = j < pa ? j : j + 1;
            out[in * n + jn] = M[i * k + j] + S * out[in * n + pa] * out[pc * n + jn];
        }
    }
}

/* MatrixUpdate.F90:455-482 */
void ora_mat_rem(const double *M, int k, int pc, int pa, double *out)
{
    int n = k - 1;
    for (int i = 0; i < n; ++i) {
        int io = i < pc ? i : i + 1;
        for (int j = 0; j < n; ++j) {
            int jo = j < pa ? j : j + 1;
            out[i * n + j] = M[io * k + jo] - M[io * k + pa] * M[pc * k + jo] / M[pc * k + pa];
        }
    }
}

/* What get_MatLogDetFull (MatrixUpdate.F90:41-99) gets from dgetrf/dgetri: inverse, log|det| and sign(det).
 * Own Gauss-Jordan with partial pivoting (LAPACK itself is third-party; any conforming result agrees to rounding). */
int ora_inverse_logdet(double *A, int k, double *logdet, double *sign)
{
    *logdet = 0.0; *sign = 1.0;
    if (k == 0) return 0;
    double *inv = (double *)calloc((size_t)k * k, sizeof(double));
    for (int i = 0; i < k; ++i) inv[i * k + i] = 1.0;
    for (int c = 0; c < k; ++c) {
        int piv = c; double best = fabs(A[c * k + c]);
        for (int r = c + 1; r < k; ++r)
            if (fabs(A[r * k + c]) > best) { best = fabs(A[r * k + c]); piv = r; }
        if (best == 0.0) { free(inv); *logdet = -HUGE_VAL; *sign = 0.0; return 1; }
        if (piv != c) {
            for (int j = 0; j < k; ++j) {
                double t = A[c * k + j]; A[c * k + j] = A[piv * k + j]; A[piv * k + j] = t;
                t = inv[c * k + j]; inv[c * k + j] = inv[piv * k + j]; inv[piv * k + j] = t;
            }
            *sign = -*sign;
        }
        double d = A[c * k + c];
        *logdet += log(fabs(d));
        if (d < 0) *sign = -*sign;
        double id = 1.0 / d;
        for (int j = 0; j < k; ++j) { A[c * k + j] *= id; inv[c * k + j] *= id; }
        for (int r = 0; r < k; ++r) {
            if (r == c) continue;
            double f = A[r * k + c];
            if (f == 0.0) continue;
            for (int j = 0; j < k; ++j) { A[r * k + j] -= f * A[c * k + j]; inv[r * k + j] -= f * inv[c * k + j]; }
        }
    }
    memcpy(A, inv, (size_t)k * k * sizeof(double));
    free(inv);
    return 0;
}

/* CTQMC.F90:2285-2424 (MeasGtauRem, MInv branch) for one flavour, plus Nfft_z.F90:376-414 (naive get_giw).
 * M(ic, ia); loops: annihilators outer, creators inner as in the reference. */
void ora_meas_flavour(const double *M, int k, const double *tau_c, const double *tau_a, double beta, double cfgsign,
                      int nleg, double *gleg, int ntau, double *gtau, int niw2, double *giw)
{
    for (int ia = 0; ia < k; ++ia) {
        for (int ic = 0; ic < k; ++ic) {
            double m = M[ic * k + ia];
            double d = tau_a[ia] - tau_c[ic];
            int idt; double x;
            if (d > 0.0) {
                idt = (int)(d / beta * (double)ntau) + 1;
                x = 2.0 / beta * d - 1.0;
            } else {
                idt = (int)((beta + tau_a[ia] - tau_c[ic]) / beta * (double)ntau) + 1;
                x = 2.0 / beta * (beta + tau_a[ia] - tau_c[ic]) - 1.0;
                m = -m;
            }
            m = m * cfgsign;
            if (gleg) {
                double plx = 0.0, plm1 = 0.0, plp1 = 0.0;
                for (int il = 0; il < nleg; ++il) {
                    if (x == -1.0 || x == 1.0) plp1 = pow(x, il);
                    if (il == 0) plp1 = 1.0;
                    else if (il == 1) plp1 = x;
                    else {
                        int l = il - 1;
                        plp1 = (2.0 * (double)l + 1.0) / (double)(l + 1) * x * plx - (double)l / (double)(l + 1) * plm1;
                    }
                    gleg[il] += m * plp1;
                    plm1 = plx; plx = plp1;
                }
            }
            if (gtau) {
                if (idt > ntau) idt = ntau;
                gtau[idt - 1] += m;
            }
        }
    }
    if (giw) {
        /* no configuration sign here (CTQMC.F90:1039-1043) */
        for (int ifr = 1; ifr <= niw2; ++ifr) {
            double freq = M_PI / beta * (double)(2 * ifr - niw2 - 1);
            double re = 0.0, im = 0.0;
            for (int ic = 0; ic < k; ++ic) {
                double cr = cos(-freq * tau_c[ic]), ci = sin(-freq * tau_c[ic]);
                for (int ia = 0; ia < k; ++ia) {
                    double ar = cos(freq * tau_a[ia]), ai = sin(freq * tau_a[ia]);
                    double m = M[ic * k + ia];
                    double pr = m * cr, pi = m * ci;            /* (m*exp_c)*exp_a, left to right */
                    re += pr * ar - pi * ai;
                    im += pr * ai + pi * ar;
                }
            }
            giw[2 * (ifr - 1)] += re;
            giw[2 * (ifr - 1) + 1] += im;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------- */
/* local problem                                                                                             */
/* ------------------------------------------------------------------------------------------------------- */

static inline int flav_bit(const ora_walker *w, int f) { return (f >> 1) + w->p.nd * (f & 1); }

/* Connect(orb,spin,ca) of block `sst` (Operator.F90:145-197): target block or -1 */
static inline int connect(const ora_walker *w, int sst, int f, int ca)
{
    return ora_lp_conn(w->lp, sst, f, ca);
}

/* ------------------------------------------------------------------------------------------------------- */
/* configuration helpers                                                                                     */
/* ------------------------------------------------------------------------------------------------------- */

static void ensure_cap(ora_walker *w, int n)
{
    if (n <= w->cap) return;
    w->cap = n * 2 + 16;
    w->ops = (op_t *)realloc(w->ops, (size_t)w->cap * sizeof(op_t));
}

/* insert keeping time order; returns index */
static int list_insert(ora_walker *w, op_t o)
{
    ensure_cap(w, w->nops + 1);
    int i = w->nops;
    while (i > 0 && w->ops[i - 1].tau > o.tau) { w->ops[i] = w->ops[i - 1]; --i; }
    w->ops[i] = o; w->nops++;
    return i;
}

static void list_remove(ora_walker *w, int idx)
{
    for (int i = idx; i + 1 < w->nops; ++i) w->ops[i] = w->ops[i + 1];
    w->nops--;
}

/* rank (0-based) of op idx among operators of the same flavour and type */
static int frank(const ora_walker *w, int idx)
{
    int r = 0;
    for (int i = 0; i < idx; ++i)
        if (w->ops[i].wid == 0 && w->ops[i].flav == w->ops[idx].flav && w->ops[i].ca == w->ops[idx].ca) ++r;
    return r;
}

/* quantum-number check of the whole sequence for outer superstate `outer` (equivalent, for a configuration that
 * was valid before the local change, to check_sst_sequence on the changed span, Trace.F90:1243-1302) */
static int sequence_valid(const ora_walker *w, int outer)
{
    int s = outer;
    for (int i = 0; i < w->nops; ++i) {
        s = connect(w, s, w->ops[i].flav, w->ops[i].ca);
        if (s < 0) return 0;
    }
    return s == outer;
}

static logval lv_add(logval a, logval b)       /* addLogTr, Trace.F90:298-309 */
{
    logval r;
    if (a.sign == 0.0) return b;
    if (b.sign == 0.0) return a;
    if (a.log >= b.log) { r.log = a.log + log(1.0 + a.sign * b.sign * exp(b.log - a.log)); r.sign = a.sign; }
    else { r.log = b.log + log(1.0 + a.sign * b.sign * exp(a.log - b.log)); r.sign = b.sign; }
    return r;
}

static void split_log(double *v, int n, double *lf)   /* Trace.F90:5341-5348 */
{
    double mv = 0.0;
    for (int i = 0; i < n; ++i) if (fabs(v[i]) > mv) mv = fabs(v[i]);
    if (mv > 0.0) for (int i = 0; i < n; ++i) v[i] /= mv;
    *lf += log(mv);
}

/* one step of the ket propagation: exp(-dt H) in block sst, then the operator (Trace.F90:5802-5829);
 * v has n(sst) entries on entry and n(target) on exit; returns the target block */
static int propagate(const ora_walker *w, int sst, double dt, int f, int ca, double *v, double *lf)
{
    const ora_locprob *lp = w->lp;
    int n = lp->sst_n[sst], nsst = connect(w, sst, f, ca);
    if (nsst < 0) return -1;
    int nn = lp->sst_n[nsst];
    const double *E = lp->eval + lp->sst_off[sst], *O = ora_lp_op(lp, sst, f, ca);
    double tmp[64], out[64];
    if (n > 1) {
        for (int i = 0; i < n; ++i) tmp[i] = exp(-(E[i] - E[0]) * dt) * v[i];
        *lf += -(E[0] - w->Egs) * dt;
        split_log(tmp, n, lf);
        for (int i = 0; i < nn; ++i) { double a = 0.0; for (int j = 0; j < n; ++j) a += O[i + nn * j] * tmp[j]; out[i] = a; }
        split_log(out, nn, lf);
    } else {
        *lf += -(E[0] - w->Egs) * dt;
        for (int i = 0; i < nn; ++i) out[i] = v[0] * O[i];
    }
    for (int i = 0; i < nn; ++i) v[i] = out[i];
    return nsst;
}

/* local trace from scratch (FullR/FullL): scalar path Trace.F90:5556-5727 when every block holds one state,
 * matrix path Trace.F90:5734-6033 otherwise (sum over the states of the outer block).  If part != NULL it receives
 * the partial traces (tparttrace). */
static logval full_trace_part(const ora_walker *w, int outer, logval *part)
{
    const ora_locprob *lp = w->lp;
    logval zero = { -HUGE_VAL, 0.0 };
    if (lp->nsmax == 1) {
        logval t = { 0.0, 1.0 };
        int s = outer; double tprev = 0.0;
        for (int i = 0; i < w->nops; ++i) {
            int ns = connect(w, s, w->ops[i].flav, w->ops[i].ca);
            if (ns < 0) return zero;
            double o11 = ora_lp_op(lp, s, w->ops[i].flav, w->ops[i].ca)[0];
            t.log += -(lp->eval[lp->sst_off[s]] - w->Egs) * (w->ops[i].tau - tprev);
            t.log += log(fabs(o11));
            t.sign *= (o11 < 0.0) ? -1.0 : 1.0;
            s = ns; tprev = w->ops[i].tau;
        }
        if (s != outer) return zero;
        t.log += -(lp->eval[lp->sst_off[s]] - w->Egs) * (w->p.beta - tprev);
        if (part) part[0] = t;
        return t;
    }
    logval tot = zero;
    int nout = lp->sst_n[outer];
    for (int st = 0; st < nout; ++st) {
        double v[64]; double lf = 0.0;
        for (int i = 0; i < nout; ++i) v[i] = (i == st) ? 1.0 : 0.0;
        int s = outer; double tprev = 0.0;
        for (int i = 0; i < w->nops; ++i) {
            s = propagate(w, s, w->ops[i].tau - tprev, w->ops[i].flav, w->ops[i].ca, v, &lf);
            if (s < 0) return zero;
            tprev = w->ops[i].tau;
        }
        if (s != outer) return zero;
        const double *E = lp->eval + lp->sst_off[s];
        double bk = exp(-(E[st] - E[0]) * (w->p.beta - tprev)) * v[st];     /* <st| exp(-(beta-t)H) |ket> */
        lf += -(E[0] - w->Egs) * (w->p.beta - tprev);
        logval pt;
        if (bk == 0.0) { pt = zero; } else { pt.log = lf + log(fabs(bk)); pt.sign = bk < 0.0 ? -1.0 : 1.0; }
        if (part) part[st] = pt;
        tot = lv_add(tot, pt);
    }
    return tot;
}

static logval full_trace(const ora_walker *w, int outer) { return full_trace_part(w, outer, NULL); }

/* Trace.F90:3594-3618 */
static double block_sign(const ora_walker *w)
{
    int nd = w->p.nd; long perm = 0;
    int tn[MAXFLAV];
    for (int f = 0; f < w->nflav; ++f) tn[f] = 2 * w->k[f];
    for (int i = 0; i < w->nops; ++i) {
        if (w->ops[i].wid) continue;                         /* only operators with hybridisation lines */
        int f = w->ops[i].flav, orb = f >> 1, spin = f & 1;  /* orb 0-based, spin 0 -> Fortran 1 */
        tn[f] -= 1;
        if (spin == 1) { for (int o = 0; o <= orb; ++o) perm += tn[2 * o + 0]; }
        else { for (int o = 0; o < orb; ++o) perm += tn[2 * o + 1]; }
        for (int o = 0; o < orb; ++o) perm += tn[2 * o + spin];
    }
    (void)nd;
    return (perm & 1) ? -1.0 : 1.0;
}

/* Trace.F90:3554-3588 */
static double permaltca(const ora_walker *w)
{
    long perm = 0; int tn[MAXFLAV];
    for (int f = 0; f < w->nflav; ++f) tn[f] = 2 * w->k[f];
    int nhyb = 0;
    for (int i = 0; i < w->nops; ++i) {
        if (w->ops[i].wid) continue;
        int f = w->ops[i].flav;
        tn[f] -= 1; ++nhyb;
        perm += tn[f] % (3 - w->ops[i].ca);
    }
    if (((nhyb / 2) % 2) == 1) perm += 1;
    return (perm & 1) ? -1.0 : 1.0;
}

/* Trace.F90:3479-3498 (diagonal hybridisation, no worm) */
/* Trace.F90:3623-3651 get_WormSign */
/* legendrep, LegendrePoly.F90:21-46: P_0..P_{n-1} at x by Bonnet's forward recursion */
static void legendre_p(double x, int n, double *pl)
{
    if (x == 1.0) { for (int i = 0; i < n; ++i) pl[i] = 1.0; return; }
    if (x == -1.0) { for (int i = 0; i < n; ++i) pl[i] = (i & 1) ? -1.0 : 1.0; return; }
    pl[0] = 1.0; if (n > 1) pl[1] = x;
    for (int i = 2; i < n; ++i) pl[i] = ((2 * i - 1) * x * pl[i - 1] - (i - 1) * pl[i - 2]) / i;
}

static double worm_sign(const ora_walker *w)
{
    long perm = 0; double wt[13]; int nw = 0;
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) { wt[w->ops[i].wid] = w->ops[i].tau; ++nw; }
    if (nw == 0) return 1.0;
    for (int i = 0; i < w->nops; ++i) {
        if (w->ops[i].wid) continue;
        for (int j = 1; j <= nw; ++j) if (w->ops[i].tau > wt[j]) ++perm;
    }
    for (int i = 1; i < nw; ++i) for (int j = i + 1; j <= nw; ++j) if (wt[i] < wt[j]) ++perm;
    return (perm & 1) ? -1.0 : 1.0;
}

static double recalc_sign(const ora_walker *w);
static double get_sign(const ora_walker *w)
{
    if (w->p.offdiag) return w->tr.sign * w->det.sign * recalc_sign(w) * worm_sign(w);      /* Trace.F90:3483-3486 */
    return w->tr.sign * w->det.sign * block_sign(w) * permaltca(w) * worm_sign(w);
}

/* ------------------------------------------------------------------------------------------------------- */
/* window and NPairs                                                                                         */
/* ------------------------------------------------------------------------------------------------------- */

static void set_window(ora_walker *w) /* Trace.F90:1691-1751 (only the bounds matter here) */
{
    if (w->num_windows > 1) {
        w->wmin = ((double)w->wpos / (double)(w->num_windows + 1)) * w->p.beta;
        w->wmax = ((double)(w->wpos + 2) / (double)(w->num_windows + 1)) * w->p.beta;
    }
}

static void init_window(ora_walker *w, double tdm) /* Trace.F90:1608-1648 */
{
    w->wpos = 0;
    if (tdm > 0.0) {
        w->taudiff_max = tdm;
        int nw = (int)floor(w->p.beta / tdm) - 1;
        w->num_windows = nw > 1 ? nw : 1;
        w->wmin = ((double)w->wpos / (double)(w->num_windows + 1)) * w->p.beta;
        w->wmax = ((double)(w->wpos + 2) / (double)(w->num_windows + 1)) * w->p.beta;
    } else {
        w->taudiff_max = w->p.beta; w->num_windows = 1; w->wmin = 0.0; w->wmax = w->p.beta;
    }
    set_window(w);
}

static void shift_window(ora_walker *w) /* Trace.F90:1652-1678 */
{
    if (w->num_windows > 1) {
        w->wpos += 1;
        if (w->wpos == w->num_windows) w->wpos = 0;
        set_window(w);
    } else w->wpos = 0;
}

/* Trace.F90:1770-1815: pairs (El earlier, mEl later) of one creator and one annihilator of equal flavour, both in
 * [wmin, wmax], distance <= taudiff_max.  If pick > 0 returns through (*i1,*i2) the pick-th pair in the reference's
 * enumeration order (gen_OperRemove, Trace.F90:4879-4917). */
static int count_pairs(const ora_walker *w, int pick, int *i1, int *i2)
{
    int n = 0;
    for (int i = 0; i < w->nops; ++i) {
        const op_t *a = &w->ops[i];
        if (a->tau < w->wmin) continue;
        if (pick == 0 && a->tau > w->wmax) break;
        if (a->wid) continue;
        for (int j = i + 1; j < w->nops; ++j) {
            const op_t *b = &w->ops[j];
            if (b->flav != a->flav && !w->p.offdiag) continue;          /* offdiag: any flavours (Trace.F90:1820-1865) */
            if (b->tau > w->wmax) break;
            if (b->tau > a->tau + w->taudiff_max) break;
            if (b->wid == 0 && b->ca != a->ca) {
                ++n;
                if (pick > 0 && n == pick) { *i1 = i; *i2 = j; return n; }
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------------- */
/* moves                                                                                                     */
/* ------------------------------------------------------------------------------------------------------- */

static inline double U01(ora_walker *w) { return w->rng->uniform(w->rng->state); }
static inline int RINT(ora_walker *w, int lo, int hi) { return w->rng->randint(w->rng->state, lo, hi); }

static int check_equal_time(const ora_walker *w, int sector);

static void accpair_add(ora_walker *w, int f, double dt)
{
    int n = w->naccpair;
    int bin = (int)floor(dt / w->accpair_max * (double)n) + 1;
    if (bin > n) bin = n;
    w->accpair[(size_t)f * n + (bin - 1)] += 1;
    if (w->accpairtau && w->apt_index < w->napt) w->accpairtau[w->apt_index++] = dt;     /* CTQMC.F90:3583-3588, 4123-4128 */
}

/* Block in front of the first operator at or after tau (at_or_after = 1) or after tau (0) in the present sequence */
static int sst_at(const ora_walker *w, double tau, int at_or_after)
{
    int s = w->outer_sst;
    for (int i = 0; i < w->nops; ++i) {
        if (at_or_after ? w->ops[i].tau >= tau : w->ops[i].tau > tau) break;
        s = connect(w, s, w->ops[i].flav, w->ops[i].ca);
    }
    return s;
}

/* Pair moves that may change the outer block (change_outer, CTQMC.F90:3422-3449, 3935-3962; check_sst_sequence walking over
 * beta, Trace.F90:1278-1295): the blocks between the two operators stay as they were, s_hi in front of the later one and s_lo
 * behind the earlier one; the sequence from the later operator over beta to the earlier one is derived anew and names the
 * new outer block.  `closed` : the two operators are part of the list (insertion) or not (removal).  Returns the new outer block, -1 if
 * the sequence is not allowed. */
static int outer_sequence(const ora_walker *w, double tlo, double thi, int closed, int s_lo, int s_hi)
{
    int s = s_hi, outer;
    for (int i = 0; i < w->nops; ++i) {
        if (closed ? w->ops[i].tau < thi : w->ops[i].tau <= thi) continue;
        s = connect(w, s, w->ops[i].flav, w->ops[i].ca);
        if (s < 0) return -1;
    }
    outer = s;
    for (int i = 0; i < w->nops; ++i) {
        if (closed ? w->ops[i].tau > tlo : w->ops[i].tau >= tlo) break;
        s = connect(w, s, w->ops[i].flav, w->ops[i].ca);
        if (s < 0) return -1;
    }
    return s == s_lo ? outer : -1;
}

/* CTQMC.F90:3353-3612 StepAdd (diagonal hybridisation) */
static void step_add(ora_walker *w, int change_outer)
{
    const double *ft;
    if (w->sector == 1) w->try_add += 1; else w->wtry_add += 1;
    if (change_outer) w->try_add_outer += 1;
    const int outer_old = w->outer_sst;
    /* pair_OperAdd, Trace.F90:3657-3698 */
    double width1 = w->wmax - w->wmin;
    double tau1 = w->wmin + width1 * U01(w);            /* i=1: CA = mod(1,2)+1 = 2 (annihilator) */
    int orb = RINT(w, 1, w->p.nd);
    int spin = RINT(w, 1, 2);
    double tmin = fmax(tau1 - w->taudiff_max, w->wmin);
    double width2 = fmin(tau1 + w->taudiff_max, w->wmax) - tmin;
    double tau2 = tmin + width2 * U01(w);               /* i=2: CA = 1 (creator) */
    double taudiff_factor = width1 * width2;
    int f = 2 * (orb - 1) + (spin - 1);

    /* process_OperAdd, Trace.F90:4275-4462: equal times are rejected and counted as QN rejections */
    int equal = (tau1 == tau2);
    for (int i = 0; i < w->nops && !equal; ++i)
        if (w->ops[i].tau == tau1 || w->ops[i].tau == tau2) equal = 1;
    if (equal) { w->accqn_add += 1; return; }

    op_t oa = { tau1, f, 2 }, oc = { tau2, f, 1 };
    int ia, ic;
    const double tlo = fmin(tau1, tau2), thi = fmax(tau1, tau2);
    const int s_lo = change_outer ? sst_at(w, tlo, 1) : 0, s_hi = change_outer ? sst_at(w, thi, 1) : 0;
    if (tau1 < tau2) { ia = list_insert(w, oa); ic = list_insert(w, oc); }
    else { ic = list_insert(w, oc); ia = list_insert(w, oa); if (ic >= ia) {} }
    /* recompute indices after both insertions */
    ia = ic = -1;
    for (int i = 0; i < w->nops; ++i) {
        if (w->ops[i].tau == tau1 && w->ops[i].flav == f && w->ops[i].ca == 2) ia = i;
        if (w->ops[i].tau == tau2 && w->ops[i].flav == f && w->ops[i].ca == 1) ic = i;
    }

    int seq_ok;
    if (change_outer) { const int o = outer_sequence(w, tlo, thi, 1, s_lo, s_hi); seq_ok = o >= 0; if (seq_ok) w->outer_sst = o; }
    else seq_ok = sequence_valid(w, w->outer_sst);
    if ((w->sector != 1 && !check_equal_time(w, w->sector)) || !seq_ok) {      /* CTQMC.F90:3455 */
        w->outer_sst = outer_old;
        list_remove(w, ia > ic ? ia : ic); list_remove(w, ia > ic ? ic : ia);
        w->accqn_add += 1;
        return;
    }

    /* get_LogDetRatPairAdd, Trace.F90:2775-2832 */
    int k = w->k[f];
    ft = w->ftab + (size_t)f * w->p.nftau;
    double *Q = (double *)malloc(sizeof(double) * (size_t)(k + 1)), *R = (double *)malloc(sizeof(double) * (size_t)(k + 1));
    double S0 = ora_hybr(ft, w->p.nftau, w->p.beta, tau1 - tau2);
    int iq = 0, ir = 0;
    for (int i = 0; i < w->nops; ++i) {
        const op_t *e = &w->ops[i];
        if (e->flav != f || i == ia || i == ic || e->wid) continue;
        if (e->ca == 2) Q[iq++] = ora_hybr(ft, w->p.nftau, w->p.beta, e->tau - tau2);   /* annihilators vs new creator */
        else R[ir++] = ora_hybr(ft, w->p.nftau, w->p.beta, tau1 - e->tau);              /* new annihilator vs creators */
    }
    double S = ora_detrat_add(w->M[f], k, Q, R, S0);
    int pc = frank(w, ic), pa = frank(w, ia);
    double sig = ((abs(pc - pa) % 2) == 0) ? 1.0 : -1.0;
    logval detrat = { log(fabs(S)), (S < 0 ? -1.0 : 1.0) * sig };

    logval trnew = full_trace(w, w->outer_sst);
    double rnd = U01(w);
    int npairs = count_pairs(w, 0, NULL, NULL);
    double prob = fabs(taudiff_factor * (double)(2 * w->p.nd) / (double)npairs
                       * exp((trnew.log - w->tr.log) + detrat.log));
    if (rnd < prob) {
        double *Mn = (double *)malloc(sizeof(double) * (size_t)(k + 1) * (k + 1));
        ora_mat_add(w->M[f], k, Q, R, S, pc, pa, Mn);
        free(w->M[f]); w->M[f] = Mn; w->k[f] = k + 1;
        w->det.log += detrat.log; w->det.sign *= detrat.sign;
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        if (w->sector == 1) w->acc_add += 1; else w->wacc_add += 1;
        if (change_outer) w->acc_add_outer += 1;
        accpair_add(w, f, fabs(tau2 - tau1));
    } else {
        w->outer_sst = outer_old;
        list_remove(w, ia > ic ? ia : ic); list_remove(w, ia > ic ? ic : ia);
    }
    free(Q); free(R);
}

/* CTQMC.F90:3892-4150 StepRem */
static void step_rem(ora_walker *w, int change_outer)
{
    if (w->sector == 1) w->try_rem += 1; else w->wtry_rem += 1;
    if (change_outer) w->try_rem_outer += 1;
    const int outer_old = w->outer_sst;
    /* gen_OperRemove, Trace.F90:4851-4950 */
    int np = count_pairs(w, 0, NULL, NULL);
    if (np <= 0) { w->accqn_rem += 1; return; }
    int pick = RINT(w, 1, np);
    int i1 = -1, i2 = -1;
    count_pairs(w, pick, &i1, &i2);
    op_t o1 = w->ops[i1], o2 = w->ops[i2];
    int f = o1.flav;
    int ic = (o1.ca == 1) ? i1 : i2, ia = (o1.ca == 2) ? i1 : i2;
    int pc = frank(w, ic), pa = frank(w, ia);
    double ta = w->ops[ia].tau;
    double hybpairremfactor = (double)np / ((w->wmax - w->wmin)
        * (fmin(ta + w->taudiff_max, w->wmax) - fmax(ta - w->taudiff_max, w->wmin)));

    const int s_lo = change_outer ? sst_at(w, o1.tau, 0) : 0, s_hi = change_outer ? sst_at(w, o2.tau, 1) : 0;
    list_remove(w, i2); list_remove(w, i1);
    int seq_ok;
    if (change_outer) { const int o = outer_sequence(w, o1.tau, o2.tau, 0, s_lo, s_hi); seq_ok = o >= 0; if (seq_ok) w->outer_sst = o; }
    else seq_ok = sequence_valid(w, w->outer_sst);
    if (!seq_ok) {
        w->outer_sst = outer_old;
        list_insert(w, o1); list_insert(w, o2);
        w->accqn_rem += 1;
        return;
    }
    int k = w->k[f];
    /* get_LogDetRatPairRem, Trace.F90:2937-2963 */
    double S = w->M[f][pc * k + pa];
    double sig = ((abs(pc - pa) % 2) == 0) ? 1.0 : -1.0;
    logval detrat = { log(fabs(S)), (S < 0 ? -1.0 : 1.0) * sig };
    logval trnew = full_trace(w, w->outer_sst);
    double prob = fabs(hybpairremfactor / (double)(2 * w->p.nd) * exp((trnew.log - w->tr.log) + detrat.log));
    double rnd = U01(w);
    if (rnd < prob) {
        if (fabs(detrat.sign * exp(detrat.log)) < 1e-12) { fprintf(stderr, "DetRat too small in removal\n"); abort(); }
        double *Mn = (double *)malloc(sizeof(double) * (size_t)((k - 1) * (k - 1) + 1));
        ora_mat_rem(w->M[f], k, pc, pa, Mn);
        free(w->M[f]); w->M[f] = Mn; w->k[f] = k - 1;
        w->det.log += detrat.log; w->det.sign *= detrat.sign;
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        accpair_add(w, f, o2.tau - o1.tau);
        if (w->sector == 1) w->acc_rem += 1; else w->wacc_rem += 1;
        if (change_outer) w->acc_rem_outer += 1;
    } else {
        w->outer_sst = outer_old;
        list_insert(w, o1); list_insert(w, o2);
    }
}

/* Trace.F90:2006-2030 is_rempair (diagonal hybridisation) */
static int is_rempair(const ora_walker *w, const op_t *a, const op_t *b)
{
    return a->flav == b->flav && a->ca != b->ca && a->wid == 0 && b->wid == 0
        && a->tau >= w->wmin && a->tau <= w->wmax && b->tau >= w->wmin && b->tau <= w->wmax
        && fabs(a->tau - b->tau) <= w->taudiff_max;
}

static int list_find(const ora_walker *w, const op_t *o)
{
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].tau == o->tau && w->ops[i].flav == o->flav && w->ops[i].ca == o->ca && w->ops[i].wid == o->wid) return i;
    return -1;
}

/* removable pairs with the operators a and b taken out of the list for the count */
static int count_pairs_without(ora_walker *w, const op_t *a, const op_t *b)
{
    const op_t oa = *a, ob = *b;
    list_remove(w, list_find(w, &oa)); list_remove(w, list_find(w, &ob));
    const int n = count_pairs(w, 0, NULL, NULL);
    list_insert(w, oa); list_insert(w, ob);
    return n;
}

/* the sum over the ways a removal of two pairs can take the same four operators out (CTQMC.F90:3684-3772, 4207-4289): o[0] < o[1]
 * and o[2] < o[3] in time are the two pairs, all four in the list; n_all with all four, n12 / n34 with only the first / second pair */
static double rempair_factor4(ora_walker *w, const op_t *o, int n_all, int n12, int n34, double *tdf)
{
    const op_t *x = NULL, *y = NULL;           /* partner of o[0] and of o[1] in the other pairing */
    if (is_rempair(w, &o[0], &o[3]) && is_rempair(w, &o[1], &o[2])) { x = &o[3]; y = &o[2]; }
    else if (is_rempair(w, &o[0], &o[2]) && is_rempair(w, &o[1], &o[3])) { x = &o[2]; y = &o[3]; }
    if (x) {
        const int n23 = count_pairs_without(w, &o[0], x);      /* NPairs_only23 (only24) */
        const int n14 = count_pairs_without(w, &o[1], y);      /* NPairs_only14 (only13) */
        *tdf = *tdf / 4.0;
        return (1.0 / (double)n_all) * (1.0 / (double)n12 + 1.0 / (double)n34 + 1.0 / (double)n14 + 1.0 / (double)n23);
    }
    *tdf = *tdf / 2.0;
    return (1.0 / (double)n_all) * (1.0 / (double)n12 + 1.0 / (double)n34);
}

static void rebuild_minv(ora_walker *w, double **Mnew, int *knew, logval *det);
static logval minv_logdet(const ora_walker *w);

/* CTQMC.F90:3615-3889 StepAdd4: two pairs at once, the inverse matrices built anew */
static void step_add4(ora_walker *w)
{
    w->try_add4 += 1;
    op_t o[4]; double tdf = 1.0, tau_ref = 0.0;
    for (int i = 0; i < 4; ++i) {                        /* pair_OperAdd with N = 4, Trace.F90:3657-3698 */
        o[i].wid = 0; o[i].ca = (i % 2 == 0) ? 2 : 1;
        if (i % 2 == 0) {
            const double width = w->wmax - w->wmin;
            tau_ref = w->wmin + width * U01(w); o[i].tau = tau_ref; tdf *= width;
            const int orb = RINT(w, 1, w->p.nd), spin = RINT(w, 1, 2);
            o[i].flav = 2 * (orb - 1) + (spin - 1);
        } else {
            const double tmin = fmax(tau_ref - w->taudiff_max, w->wmin);
            const double width = fmin(tau_ref + w->taudiff_max, w->wmax) - tmin;
            o[i].tau = tmin + width * U01(w); tdf *= width;
            o[i].flav = o[i - 1].flav;
        }
    }
    int equal = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < i; ++j) if (o[i].tau == o[j].tau) equal = 1;
        for (int j = 0; j < w->nops; ++j) if (w->ops[j].tau == o[i].tau) equal = 1;
    }
    if (equal) return;                                   /* CTQMC.F90:3790-3803: rejected, no counter */
    for (int p = 0; p < 4; p += 2) if (o[p].tau > o[p + 1].tau) { op_t t = o[p]; o[p] = o[p + 1]; o[p + 1] = t; }   /* process_OperAdd sorts each pair in time */
    list_insert(w, o[0]); list_insert(w, o[1]);
    const int n12 = count_pairs(w, 0, NULL, NULL);
    list_insert(w, o[2]); list_insert(w, o[3]);
    const int n34 = count_pairs_without(w, &o[0], &o[1]);
    const int n_all = count_pairs(w, 0, NULL, NULL);
    const double rpf = rempair_factor4(w, o, n_all, n12, n34, &tdf);

    if ((w->sector != 1 && !check_equal_time(w, w->sector)) || !sequence_valid(w, w->outer_sst)) {
        for (int i = 0; i < 4; ++i) list_remove(w, list_find(w, &o[i]));
        return;
    }
    double *Mn[MAXFLAV]; int kn[MAXFLAV]; logval detnew;
    rebuild_minv(w, Mn, kn, &detnew);                    /* get_InvFullHybr */
    const logval dold = minv_logdet(w);                  /* get_FullLogDet of the present inverse matrices */
    const logval detrat = { detnew.log + dold.log, detnew.sign * dold.sign };
    const logval trnew = full_trace(w, w->outer_sst);
    const double rnd = U01(w);
    const double nf2 = (double)(2 * w->p.nd) * (double)(2 * w->p.nd);
    const double prob = fabs(tdf * nf2 * rpf * exp((trnew.log - w->tr.log) + detrat.log));
    if (rnd < prob) {
        for (int f = 0; f < w->nflav; ++f) { free(w->M[f]); w->M[f] = Mn[f]; w->k[f] = kn[f]; }
        w->det = detnew; w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_add4 += 1;
    } else {
        for (int f = 0; f < w->nflav; ++f) free(Mn[f]);
        for (int i = 0; i < 4; ++i) list_remove(w, list_find(w, &o[i]));
    }
}

/* CTQMC.F90:4153-4430 StepRem4 */
static void step_rem4(ora_walker *w)
{
    w->try_rem4 += 1;
    const int n_old = count_pairs(w, 0, NULL, NULL);
    if (n_old <= 0) return;
    op_t o[4]; int i1 = -1, i2 = -1;
    count_pairs(w, RINT(w, 1, n_old), &i1, &i2);
    o[0] = w->ops[i1]; o[1] = w->ops[i2];
    const double W = w->wmax - w->wmin;
    {
        const double ta = o[0].ca == 2 ? o[0].tau : o[1].tau;
        /* hybpairremfactor of the first pair, Trace.F90:4945-4948 */
        const double f1 = (double)n_old / (W * (fmin(ta + w->taudiff_max, w->wmax) - fmax(ta - w->taudiff_max, w->wmin)));
        list_remove(w, i2); list_remove(w, i1);
        const int n34 = count_pairs(w, 0, NULL, NULL);
        if (n34 <= 0) { list_insert(w, o[0]); list_insert(w, o[1]); return; }
        count_pairs(w, RINT(w, 1, n34), &i1, &i2);
        o[2] = w->ops[i1]; o[3] = w->ops[i2];
        const double tb = o[2].ca == 2 ? o[2].tau : o[3].tau;
        const double f2 = (double)n34 / (W * (fmin(tb + w->taudiff_max, w->wmax) - fmax(tb - w->taudiff_max, w->wmin)));
        double tdf = (double)n_old * (double)n34 / f1 / f2;                  /* CTQMC.F90:4195-4196 */
        list_insert(w, o[0]); list_insert(w, o[1]);
        const int n12 = count_pairs_without(w, &o[2], &o[3]);
        const double rpf = rempair_factor4(w, o, n_old, n12, n34, &tdf);
        for (int i = 0; i < 4; ++i) list_remove(w, list_find(w, &o[i]));
        if (!sequence_valid(w, w->outer_sst)) { for (int i = 0; i < 4; ++i) list_insert(w, o[i]); return; }
        double *Mn[MAXFLAV]; int kn[MAXFLAV]; logval detnew;
        rebuild_minv(w, Mn, kn, &detnew);
        const logval dold = minv_logdet(w);
        const logval detrat = { detnew.log + dold.log, detnew.sign * dold.sign };
        const logval trnew = full_trace(w, w->outer_sst);
        const double nf2 = (double)((2 * w->p.nd) * (2 * w->p.nd));
        const double prob = fabs(1.0 / nf2 / tdf / rpf * exp((trnew.log - w->tr.log) + detrat.log));
        const double rnd = U01(w);
        if (rnd < prob) {
            for (int f = 0; f < w->nflav; ++f) { free(w->M[f]); w->M[f] = Mn[f]; w->k[f] = kn[f]; }
            w->det = detnew; w->tr = trnew;
            w->cfgsign = -get_sign(w);
            w->acc_rem4 += 1;
        } else {
            for (int f = 0; f < w->nflav; ++f) free(Mn[f]);
            for (int i = 0; i < 4; ++i) list_insert(w, o[i]);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------- */
/* off-diagonal hybridisation (offdiag = 1, fast updates): one inverse matrix over all flavours              */
/* ------------------------------------------------------------------------------------------------------- */
static double hybr_full(const ora_walker *w, int f1, int f2, double dtau)      /* get_HybrF_full(tau, b1, s1, b2, s2), Trace.F90:1038-1057 */
{
    return ora_hybr(w->ffull + ((size_t)f1 * w->nflav + f2) * w->p.nftau, w->p.nftau, w->p.beta, dtau);
}

/* row (creator) or column (annihilator) of operator idx in the full matrix: flavours in ascending order, time inside (add_ftau_offset, Trace.F90:3002-3040) */
static int od_rank(const ora_walker *w, int idx)
{
    const op_t *o = &w->ops[idx]; int r = 0;
    for (int i = 0; i < w->nops; ++i) {
        const op_t *e = &w->ops[i];
        if (i == idx || e->wid || e->ca != o->ca) continue;
        if (e->flav < o->flav || (e->flav == o->flav && i < idx)) ++r;
    }
    return r;
}

/* get_FullHybr_offdiag + get_MatLogDetFull (Trace.F90:2690-2770): the inverse of F(tau_a - tau_c; flavour of a, flavour of c) */
static void rebuild_full(const ora_walker *w, double **Mnew, int *ktnew, logval *det)
{
    int kt = 0;
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid == 0 && w->ops[i].ca == 1) ++kt;
    *ktnew = kt; det->log = 0.0; det->sign = 1.0;
    *Mnew = (double *)malloc(sizeof(double) * (size_t)(kt * kt + 1));
    if (kt == 0) return;
    int *ic = (int *)malloc(sizeof(int) * kt), *ia = (int *)malloc(sizeof(int) * kt); int nc = 0, na = 0;
    for (int f = 0; f < w->nflav; ++f) for (int i = 0; i < w->nops; ++i) {
        if (w->ops[i].wid || w->ops[i].flav != f) continue;
        if (w->ops[i].ca == 1) ic[nc++] = i; else ia[na++] = i;
    }
    double *F = (double *)malloc(sizeof(double) * (size_t)kt * kt);
    for (int a = 0; a < kt; ++a) for (int c = 0; c < kt; ++c)
        F[a * kt + c] = hybr_full(w, w->ops[ia[a]].flav, w->ops[ic[c]].flav, w->ops[ia[a]].tau - w->ops[ic[c]].tau);
    ora_inverse_logdet(F, kt, &det->log, &det->sign);      /* F^-1 [creator][annihilator] */
    memcpy(*Mnew, F, sizeof(double) * (size_t)kt * kt);
    free(F); free(ic); free(ia);
}

/* recalc_sign, Trace.F90:3500-3545: parity of the permutation that sorts the hybridising operators from time order into
 * the order (creators first; flavour, time descending), times (-1)^(1 + 2 + ... + (n/2 - 1)) */
static double recalc_sign(const ora_walker *w)
{
    int n = 0; double *key = (double *)malloc(sizeof(double) * (size_t)(w->nops + 1));
    for (int i = 0; i < w->nops; ++i) {
        const op_t *e = &w->ops[i];
        if (e->wid) continue;
        key[n++] = (double)(3 - e->ca) * 100000.0 + (double)((e->flav >> 1) + 1) * 10000.0 + (double)((e->flav & 1) + 1) * 1000.0 + e->tau;
    }
    long perm = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
        if (key[i] > key[j]) { const double t = key[i]; key[i] = key[j]; key[j] = t; ++perm; }
    for (int i = 1; i <= n / 2 - 1; ++i) perm += i;
    free(key);
    return (perm % 2) ? -1.0 : 1.0;
}

/* StepAdd with offdiag (CTQMC.F90:3353-3612): the two flavours are drawn independently (pair_OperAdd, force_diagonal = false) */
static void step_add_od(ora_walker *w)
{
    w->try_add += 1;
    const double width1 = w->wmax - w->wmin;
    const double tau1 = w->wmin + width1 * U01(w);                 /* annihilator */
    const int orb1 = RINT(w, 1, w->p.nd), spin1 = RINT(w, 1, 2);
    const double tmin = fmax(tau1 - w->taudiff_max, w->wmin);
    const double width2 = fmin(tau1 + w->taudiff_max, w->wmax) - tmin;
    const double tau2 = tmin + width2 * U01(w);                    /* creator */
    const int orb2 = RINT(w, 1, w->p.nd), spin2 = RINT(w, 1, 2);
    const double taudiff_factor = width1 * width2;
    const int fa = 2 * (orb1 - 1) + (spin1 - 1), fc = 2 * (orb2 - 1) + (spin2 - 1);
    int equal = (tau1 == tau2);
    for (int i = 0; i < w->nops && !equal; ++i) if (w->ops[i].tau == tau1 || w->ops[i].tau == tau2) equal = 1;
    if (equal) { w->accqn_add += 1; return; }
    op_t oa = { tau1, fa, 2, 0 }, oc = { tau2, fc, 1, 0 };
    list_insert(w, oa); list_insert(w, oc);
    const int ia = list_find(w, &oa), ic = list_find(w, &oc);
    if (!sequence_valid(w, w->outer_sst)) {
        list_remove(w, ia > ic ? ia : ic); list_remove(w, ia > ic ? ic : ia);
        w->accqn_add += 1;
        return;
    }
    /* get_LogDetRatPairAdd_full, Trace.F90:2835-2932 (index order of the three look-ups as written there) */
    const int kt = w->kt;
    double *Q = (double *)malloc(sizeof(double) * (size_t)(kt + 1)), *R = (double *)malloc(sizeof(double) * (size_t)(kt + 1));
    const double S0 = hybr_full(w, fa, fc, tau1 - tau2);
    int iq = 0, ir = 0;
    for (int f = 0; f < w->nflav; ++f) {
        for (int i = 0; i < w->nops; ++i) {
            const op_t *e = &w->ops[i];
            if (e->wid || e->flav != f || e->ca != 2 || i == ia) continue;
            Q[iq++] = hybr_full(w, fc, f, e->tau - tau2);
        }
        for (int i = 0; i < w->nops; ++i) {
            const op_t *e = &w->ops[i];
            if (e->wid || e->flav != f || e->ca != 1 || i == ic) continue;
            R[ir++] = hybr_full(w, fa, f, tau1 - e->tau);
        }
    }
    const double S = ora_detrat_add(w->Mf, kt, Q, R, S0);
    const int pc = od_rank(w, ic), pa = od_rank(w, ia);
    const double sig = ((abs(pc - pa) % 2) == 0) ? 1.0 : -1.0;
    const logval detrat = { log(fabs(S)), (S < 0 ? -1.0 : 1.0) * sig };
    const logval trnew = full_trace(w, w->outer_sst);
    const double rnd = U01(w);
    const int npairs = count_pairs(w, 0, NULL, NULL);
    const double nf2 = (double)((2 * w->p.nd) * (2 * w->p.nd));
    const double prob = fabs(taudiff_factor * nf2 / (double)npairs * exp((trnew.log - w->tr.log) + detrat.log));
    if (rnd < prob) {
        double *Mn = (double *)malloc(sizeof(double) * (size_t)(kt + 1) * (kt + 1));
        ora_mat_add(w->Mf, kt, Q, R, S, pc, pa, Mn);
        free(w->Mf); w->Mf = Mn; w->kt = kt + 1;
        w->det.log += detrat.log; w->det.sign *= detrat.sign;
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_add += 1;
        accpair_add(w, tau1 < tau2 ? fa : fc, fabs(tau2 - tau1));          /* flavour of the earlier operator (CTQMC.F90:3579) */
    } else {
        list_remove(w, ia > ic ? ia : ic); list_remove(w, ia > ic ? ic : ia);
    }
    free(Q); free(R);
}

/* StepRem with offdiag (CTQMC.F90:3892-4150) */
static void step_rem_od(ora_walker *w)
{
    w->try_rem += 1;
    const int np = count_pairs(w, 0, NULL, NULL);
    if (np <= 0) { w->accqn_rem += 1; return; }
    int i1 = -1, i2 = -1;
    count_pairs(w, RINT(w, 1, np), &i1, &i2);
    const op_t o1 = w->ops[i1], o2 = w->ops[i2];
    const int ic = (o1.ca == 1) ? i1 : i2, ia = (o1.ca == 2) ? i1 : i2;
    const int pc = od_rank(w, ic), pa = od_rank(w, ia);
    const double ta = w->ops[ia].tau; const int fcre = w->ops[ic].flav;
    const double hybpairremfactor = (double)np / ((w->wmax - w->wmin)
        * (fmin(ta + w->taudiff_max, w->wmax) - fmax(ta - w->taudiff_max, w->wmin)));
    list_remove(w, i2); list_remove(w, i1);
    if (!sequence_valid(w, w->outer_sst)) { list_insert(w, o1); list_insert(w, o2); w->accqn_rem += 1; return; }
    const int kt = w->kt;
    const double S = w->Mf[pc * kt + pa];
    const double sig = ((abs(pc - pa) % 2) == 0) ? 1.0 : -1.0;
    const logval detrat = { log(fabs(S)), (S < 0 ? -1.0 : 1.0) * sig };
    const logval trnew = full_trace(w, w->outer_sst);
    const double nf2 = (double)((2 * w->p.nd) * (2 * w->p.nd));
    const double prob = fabs(hybpairremfactor / nf2 * exp((trnew.log - w->tr.log) + detrat.log));
    const double rnd = U01(w);
    if (rnd < prob) {
        if (fabs(detrat.sign * exp(detrat.log)) < 1e-12) { fprintf(stderr, "DetRat too small in removal\n"); abort(); }
        double *Mn = (double *)malloc(sizeof(double) * (size_t)((kt - 1) * (kt - 1) + 1));
        ora_mat_rem(w->Mf, kt, pc, pa, Mn);
        free(w->Mf); w->Mf = Mn; w->kt = kt - 1;
        w->det.log += detrat.log; w->det.sign *= detrat.sign;
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        accpair_add(w, fcre, o2.tau - o1.tau);                             /* flavour of the creator (CTQMC.F90:4114-4122) */
        w->acc_rem += 1;
    } else { list_insert(w, o1); list_insert(w, o2); }
}

/* CTQMC.F90:4838-4989 StepFlavourchange_general (flavourchange_moves = 1; off-diagonal hybridisation): two operators inside the
 * window get new random flavours (propose_flavourexchange_general, Trace.F90:2191-2274), the matrix is built anew */
static void step_flavc(ora_walker *w)
{
    w->try_flavc += 1;
    int jlo = 0, jhi = 0;
    for (int i = 0; i < w->nops; ++i) { if (w->ops[i].tau < w->wmin) ++jlo; if (w->ops[i].tau <= w->wmax) ++jhi; }
    const int N = jhi - jlo;
    if (N <= 1) return;
    const int pos1 = RINT(w, 1, N - 1), pos2 = RINT(w, 1, N - pos1);
    const int i1 = jlo + pos1 - 1, i2 = i1 + pos2;
    if (w->ops[i1].wid || w->ops[i2].wid) return;
    const int f1o = w->ops[i1].flav, f2o = w->ops[i2].flav;
    { const int orb = RINT(w, 1, w->p.nd), spin = RINT(w, 1, 2); w->ops[i1].flav = 2 * (orb - 1) + (spin - 1); }
    { const int orb = RINT(w, 1, w->p.nd), spin = RINT(w, 1, 2); w->ops[i2].flav = 2 * (orb - 1) + (spin - 1); }
    if (!sequence_valid(w, w->outer_sst)) { w->ops[i1].flav = f1o; w->ops[i2].flav = f2o; return; }
    const logval trnew = full_trace(w, w->outer_sst);
    double *Mfn; int ktn; logval detnew;
    rebuild_full(w, &Mfn, &ktn, &detnew);
    const double prob = fabs(exp((trnew.log - w->tr.log) + (detnew.log - w->det.log)));
    if (U01(w) < prob) {
        free(w->Mf); w->Mf = Mfn; w->kt = ktn; w->det = detnew; w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_flavc += 1;
    } else { free(Mfn); w->ops[i1].flav = f1o; w->ops[i2].flav = f2o; }
}

/* the pair move of one step: CTQMC.F90:8441-8449, 8456-8464 (two random numbers, used or not) */
static void pair_move(ora_walker *w, int add)
{
    const int four = !(U01(w) >= w->p.p_4op);
    if (!four && w->p.offdiag) { (void)U01(w); if (add) step_add_od(w); else step_rem_od(w); }
    else if (!four) { const int outer = U01(w) < w->p.p_outer; if (add) step_add(w, outer); else step_rem(w, outer); }
    else if (w->p.flavc) step_flavc(w);                /* b_exch: the flavour change takes the place of the two-pair moves */
    else if (add) step_add4(w); else step_rem4(w);
}

/* Trace.F90:2279-2333 get_InvFullHybr: rebuild every M block as the inverse of F(tau_a_i - tau_c_j), det = prod */
static void rebuild_minv(ora_walker *w, double **Mnew, int *knew, logval *det)
{
    det->log = 0.0; det->sign = 1.0;
    for (int f = 0; f < w->nflav; ++f) {
        int k = 0;
        for (int i = 0; i < w->nops; ++i) if (w->ops[i].flav == f && w->ops[i].ca == 1 && !w->ops[i].wid) ++k;     /* hybridising operators only */
        knew[f] = k;
        Mnew[f] = (double *)malloc(sizeof(double) * (size_t)(k * k + 1));
        if (k == 0) continue;
        const double *ft = w->ftab + (size_t)f * w->p.nftau;
        double *tc = (double *)malloc(sizeof(double) * k), *ta = (double *)malloc(sizeof(double) * k);
        int nc = 0, na = 0;
        for (int i = 0; i < w->nops; ++i) {
            if (w->ops[i].flav != f || w->ops[i].wid) continue;
            if (w->ops[i].ca == 1) tc[nc++] = w->ops[i].tau; else ta[na++] = w->ops[i].tau;
        }
        /* Fortran Mat(pos_a, pos_c) column major == row-major [c][a]: after inversion this array read as
         * row-major [c][a] is M(creator, annihilator) */
        for (int c = 0; c < k; ++c)
            for (int a = 0; a < k; ++a)
                Mnew[f][c * k + a] = ora_hybr(ft, w->p.nftau, w->p.beta, ta[a] - tc[c]);
        /* the array holds F^T in row-major terms; inverse of transpose = transpose of inverse, so inverting in
         * place gives (F^-1)^T[c][a] = F^-1[a]... keep explicit: build F[a][c], invert, transpose */
        double *F = (double *)malloc(sizeof(double) * (size_t)k * k);
        for (int a = 0; a < k; ++a) for (int c = 0; c < k; ++c) F[a * k + c] = Mnew[f][c * k + a];
        double ld, sg;
        ora_inverse_logdet(F, k, &ld, &sg);          /* F^-1[c][a] */
        for (int c = 0; c < k; ++c) for (int a = 0; a < k; ++a) Mnew[f][c * k + a] = F[c * k + a];
        det->log += ld; det->sign *= sg;
        free(F); free(tc); free(ta);
    }
}

/* log-determinant of the current M blocks: Trace.F90:2665-2685 get_FullLogDet */
static logval minv_logdet(const ora_walker *w)
{
    logval d = { 0.0, 1.0 };
    for (int f = 0; f < w->nflav; ++f) {
        int k = w->k[f];
        if (k == 0) continue;
        double *A = (double *)malloc(sizeof(double) * (size_t)k * k);
        memcpy(A, w->M[f], sizeof(double) * (size_t)k * k);
        double ld, sg;
        ora_inverse_logdet(A, k, &ld, &sg);
        d.log += ld; d.sign *= sg;
        free(A);
    }
    return d;
}

/* Trace.F90:2126-2188 globalUpdate: gu[0] = flip c<->c+, gu[i] (1-based, spin-major flavour numbers) = new flavour */
static void apply_global(ora_walker *w, const int *gu)
{
    int nd = w->p.nd;
    for (int i = 0; i < w->nops; ++i) {
        int f = w->ops[i].flav;
        int num = (f >> 1) + nd * (f & 1) + 1;
        int t = gu[num];
        int orb = (t - 1) % nd, spin = (t - 1) / nd;
        w->ops[i].flav = 2 * orb + spin;
        if (gu[0] != 0) w->ops[i].ca = 3 - w->ops[i].ca;
    }
}

/* Trace.F90:1402-1433 check_qn_all_ssts */
static int check_all_ssts(const ora_walker *w, int *ssts)
{
    int n = w->nsst, possible = n;
    for (int k = 0; k < n; ++k) ssts[k] = k;
    for (int i = 0; i < w->nops; ++i) {
        for (int k = 0; k < n; ++k) {
            if (ssts[k] == -1) continue;
            ssts[k] = connect(w, ssts[k], w->ops[i].flav, w->ops[i].ca);
            if (ssts[k] == -1) --possible;
        }
        if (possible <= 0) return 0;
    }
    for (int k = 0; k < n; ++k) {
        if (ssts[k] == -1) continue;
        if (ssts[k] != k) { ssts[k] = -1; --possible; }
    }
    return possible;
}

/* CTQMC.F90:4433-4554 StepGlob */
static void step_glob(ora_walker *w)
{
    int nf = w->nflav, nd = w->p.nd;
    int gu[MAXFLAV + 1], inv[MAXFLAV + 1];
    w->try_glob += 1;
    if (w->nops == 0) return;
    double r = U01(w);
    int spinflip = 0;
    if (r < 0.3) {                                   /* gen_SpinFlipUpdate, Trace.F90:1150-1166 */
        spinflip = 1; gu[0] = 0;
        for (int i = 1; i <= nd; ++i) gu[i] = i + nd;
        for (int i = nd + 1; i <= 2 * nd; ++i) gu[i] = i - nd;
    } else if (r < 0.6) {                            /* gen_CAFlip, Trace.F90:1186-1198 */
        gu[0] = 1; for (int i = 1; i <= nf; ++i) gu[i] = i;
    } else if (r < 0.8 && w->p.nsym > 0) {           /* gen_SymUpdate, Trace.F90:1170-1182: one of the listed moves or of their inverses */
        const int isym = RINT(w, 1, 2 * w->p.nsym);
        gu[0] = 0;
        if (isym <= w->p.nsym) for (int j = 1; j <= nf; ++j) gu[j] = w->p.symmoves[(isym - 1) * 16 + j - 1];
        else for (int j = 1; j <= nf; ++j) gu[w->p.symmoves[(isym - w->p.nsym - 1) * 16 + j - 1]] = j;      /* Trace.F90:475-477 */
    } else {                                         /* gen_GlobalUpdate_new + rand_perm, Trace.F90:1083-1123 */
        for (int i = 1; i <= nf; ++i) gu[i] = i;
        for (int j = 1; j <= nf; ++j) { int i = RINT(w, j, nf); int t = gu[j]; gu[j] = gu[i]; gu[i] = t; }
        gu[0] = RINT(w, 0, 1);
    }
    int outer_old = w->outer_sst;
    op_t *saved = (op_t *)malloc(sizeof(op_t) * (size_t)w->nops);
    memcpy(saved, w->ops, sizeof(op_t) * (size_t)w->nops);
    double sst_factor;
    if (spinflip && w->p.has_szt) {                  /* perform_spinflip_check_qn, Trace.F90:1306-1339 */
        int old_state = w->lp->states[w->lp->sst_off[outer_old]];        /* first Fock state of the block */
        int down = old_state & ((1 << nd) - 1), up = (old_state >> nd) & ((1 << nd) - 1);
        w->outer_sst = w->lp->sst_of_state[up | (down << nd)];
        apply_global(w, gu);
        sst_factor = sequence_valid(w, w->outer_sst) ? 1.0 : 0.0;
    } else {                                         /* perform_global_check_qn, Trace.F90:1345-1394 */
        int *so = (int *)malloc(sizeof(int) * (size_t)w->nsst), *sn = (int *)malloc(sizeof(int) * (size_t)w->nsst);
        int pold = check_all_ssts(w, so);
        if (pold <= 0) { fprintf(stderr, "impossible old configuration\n"); abort(); }
        apply_global(w, gu);
        int pnew = check_all_ssts(w, sn);
        if (pnew <= 0) sst_factor = 0.0;
        else {
            sst_factor = (double)pnew / (double)pold;
            int kk = RINT(w, 1, pnew);
            for (int i = 0; i < w->nsst; ++i) if (sn[i] != -1) { if (--kk <= 0) { w->outer_sst = sn[i]; break; } }
        }
        free(so); free(sn);
    }
    (void)inv;
    if (sst_factor == 0.0) {
        w->outer_sst = outer_old; memcpy(w->ops, saved, sizeof(op_t) * (size_t)w->nops); free(saved);
        w->accqn_glob += 1;
        return;
    }
    if (w->p.offdiag) {                                /* CTQMC.F90:4491-4494: DetRat = DetNew / Det */
        double *Mfn; int ktn; logval detnew;
        rebuild_full(w, &Mfn, &ktn, &detnew);
        const logval trnew = full_trace(w, w->outer_sst);
        const double rnd = U01(w);
        if (rnd < fabs(exp((trnew.log - w->tr.log) + (detnew.log - w->det.log)) * sst_factor)) {
            free(w->Mf); w->Mf = Mfn; w->kt = ktn; w->det = detnew; w->tr = trnew;
            w->cfgsign = -get_sign(w);
            w->acc_glob += 1;
        } else {
            free(Mfn);
            w->outer_sst = outer_old; memcpy(w->ops, saved, sizeof(op_t) * (size_t)w->nops);
        }
        free(saved);
        return;
    }
    double *Mn[MAXFLAV]; int kn[MAXFLAV]; logval detnew;
    rebuild_minv(w, Mn, kn, &detnew);
    logval dold = minv_logdet(w);
    logval detrat = { detnew.log + dold.log, detnew.sign * dold.sign };
    logval trnew = full_trace(w, w->outer_sst);
    double rnd = U01(w);
    if (rnd < fabs(exp((trnew.log - w->tr.log) + detrat.log) * sst_factor)) {
        for (int f = 0; f < nf; ++f) { free(w->M[f]); w->M[f] = Mn[f]; w->k[f] = kn[f]; }
        w->det = detnew; w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_glob += 1;
    } else {
        for (int f = 0; f < nf; ++f) free(Mn[f]);
        w->outer_sst = outer_old; memcpy(w->ops, saved, sizeof(op_t) * (size_t)w->nops);
    }
    free(saved);
}

/* CTQMC.F90:4570-4638 StepChangeOuter (no state sampling): draw the outer superstate of the empty configuration */
static int draw_outer(ora_walker *w)
{
    int n = w->nsst; double sum = 0.0;
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    for (int s = 0; s < n; ++s) { logval t = full_trace_part(w, s, NULL); v[s] = t.sign * exp(t.log); sum += v[s]; }
    for (int s = 0; s < n; ++s) v[s] /= sum;
    double r = U01(w), acc = 0.0;
    int sel = n - 1;
    for (int s = 0; s < n; ++s) { acc += v[s]; sel = s; if (acc >= r && v[s] != 0.0) break; }
    free(v);
    return sel;
}

static void step_change_outer(ora_walker *w)
{
    w->outer_sst = draw_outer(w);
    w->tr = full_trace(w, w->outer_sst);
    w->cfgsign = -get_sign(w);
}

/* CTQMC.F90:4641-4835 StepShiftTau_proper + Trace.F90:2355-2457 */
static void step_shift(ora_walker *w)
{
    if (w->nops == 0) { step_change_outer(w); return; }
    double beta = w->p.beta;
    double dt = U01(w) * beta;
    if (dt == 0.0 || dt == beta) return;
    w->try_shift += 1;
    int n = w->nops, nf = w->nflav;
    /* superstate in front of the first operator with tau >= dt (the new outer one); all wrap if none */
    int first = 0; while (first < n && w->ops[first].tau < dt) ++first;
    int outer_old = w->outer_sst, s = w->outer_sst;
    if (first < n) for (int i = 0; i < first; ++i) s = connect(w, s, w->ops[i].flav, w->ops[i].ca);
    int wrapped[MAXFLAV][3]; memset(wrapped, 0, sizeof(wrapped));
    op_t *saved = (op_t *)malloc(sizeof(op_t) * (size_t)n), *tmp = (op_t *)malloc(sizeof(op_t) * (size_t)n);
    memcpy(saved, w->ops, sizeof(op_t) * (size_t)n);
    for (int i = 0; i < n; ++i) if (w->ops[i].tau < dt) wrapped[w->ops[i].flav][w->ops[i].ca] += 1;
    int m = 0;
    int cut = (first < n) ? first : 0;
    for (int i = cut; i < n; ++i) { tmp[m] = saved[i]; tmp[m].tau = fmod(saved[i].tau - dt, beta); if (tmp[m].tau < 0) tmp[m].tau += beta; ++m; }
    for (int i = 0; i < cut; ++i) { tmp[m] = saved[i]; tmp[m].tau = fmod(saved[i].tau - dt, beta); if (tmp[m].tau < 0) tmp[m].tau += beta; ++m; }
    memcpy(w->ops, tmp, sizeof(op_t) * (size_t)n);
    w->outer_sst = s;
    int wpos_old = w->wpos; double wmin_old = w->wmin, wmax_old = w->wmax;
    {   /* CTQMC.F90:4744-4746 */
        int nw = w->num_windows;
        int a = (int)floor((dt / beta) * (double)(nw + 1)); a = ((a % nw) + nw) % nw;
        w->wpos = (((wpos_old - a) % nw) + nw) % nw;
        set_window(w);
    }
    logval trnew = full_trace(w, w->outer_sst);
    double rnd = U01(w);
    if (rnd < fabs(trnew.sign * w->tr.sign * exp(trnew.log - w->tr.log))) {
        w->tr = trnew;
        double detrat = 1.0;
        if (w->p.offdiag) {                            /* update_InvFullHybrShiftTau_offdiag, Trace.F90:2462-2601 */
            int nc[MAXFLAV], na[MAXFLAV], offc[MAXFLAV + 1], offa[MAXFLAV + 1];
            for (int f = 0; f < nf; ++f) nc[f] = na[f] = 0;
            for (int i = 0; i < n; ++i) if (w->ops[i].ca == 1) nc[w->ops[i].flav] += 1; else na[w->ops[i].flav] += 1;
            offc[0] = offa[0] = 0;
            for (int f = 0; f < nf; ++f) { offc[f + 1] = offc[f] + nc[f]; offa[f + 1] = offa[f] + na[f]; }
            const int kt = w->kt;
            double *Mn = (double *)malloc(sizeof(double) * (size_t)(kt * kt + 1));
            for (int fr = 0; fr < nf; ++fr) for (int fc = 0; fc < nf; ++fc) {
                const int sR = nc[fr], sC = na[fc], wr = wrapped[fr][1], wcol = wrapped[fc][2];
                for (int i = 0; i < sR; ++i) for (int j = 0; j < sC; ++j) {
                    const int io = (i + wr) % sR, jo = (j + wcol) % sC;
                    const double sg = ((i >= sR - wr) ? -1.0 : 1.0) * ((j >= sC - wcol) ? -1.0 : 1.0);
                    Mn[(offc[fr] + i) * kt + offa[fc] + j] = sg * w->Mf[(offc[fr] + io) * kt + offa[fc] + jo];
                }
            }
            free(w->Mf); w->Mf = Mn;
            for (int f = 0; f < nf; ++f) {
                if ((nc[f] * wrapped[f][1]) % 2) detrat = -detrat;
                if ((na[f] * wrapped[f][2]) % 2) detrat = -detrat;
            }
        }
        for (int f = 0; f < nf && !w->p.offdiag; ++f) {
            int k = w->k[f], wc = wrapped[f][1], wa = wrapped[f][2];
            if (!((wc == 0 && wa == 0) || (wc == k && wa == k))) {
                double *Mn = (double *)malloc(sizeof(double) * (size_t)k * k);
                for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
                    int io = (i + wc) % k, jo = (j + wa) % k;
                    double sg = ((i >= k - wc) ? -1.0 : 1.0) * ((j >= k - wa) ? -1.0 : 1.0);
                    Mn[i * k + j] = sg * w->M[f][io * k + jo];
                }
                free(w->M[f]); w->M[f] = Mn;
            }
            if ((k % 2) != 0 && ((wc + wa) % 2) != 0) detrat = -detrat;
        }
        w->det.sign *= detrat;
        w->cfgsign = -get_sign(w);
        w->acc_shift += 1;
    } else {
        memcpy(w->ops, saved, sizeof(op_t) * (size_t)n);
        w->outer_sst = outer_old; w->wpos = wpos_old; w->wmin = wmin_old; w->wmax = wmax_old;
    }
    free(saved); free(tmp);
}

/* ------------------------------------------------------------------------------------------------------- */
/* worm moves: sector 4 (two-particle Green's function), four operators c c+ c c+ without hybridisation lines,  */
/* and sector 2 (one-particle Green's function), two operators c c+ (NOperWorm, CTQMC.F90:5282-5285)             */
/* ------------------------------------------------------------------------------------------------------- */

/* NOperWorm, Parameters.F90:45 */
static int worm_nops(const ora_walker *w)
{
    static const int n[17] = {0, 0, 2, 4, 4, 6, 4, 4, 4, 4, 6, 12, 8, 8, 4, 4, 4};
    return (w->isector >= 2 && w->isector <= 16) ? n[w->isector] : 4;
}
static int worm_is_ie(int sector) { return sector == 3 || sector == 5 || (sector >= 10 && sector <= 16); }

/* get_Proposal (CTQMC.F90:5677-5796): SectorG / SectorG4 (beta Nd)^n eta; P2 / P2pp beta^2 Nd^4 eta; P3 / P3pp beta^3 Nd^4 eta;
 * the sectors with U-matrix factors (3, 5, 10 - 16) below */
#define UM(a, b, c, d) w->umat[(((size_t)(a) * nf + (b)) * nf + (c)) * nf + (d)]
#define UA1(a, b, c, d) (0.5 * (UM(a, b, c, d) - UM(b, a, c, d)))                  /* u_matrix_antisymm_1, CTQMC.F90:7930 */
#define UA2(a, b, c, d) (0.5 * (UM(a, b, c, d) - UM(a, b, d, c)))                  /* u_matrix_antisymm_2, 7932 */
#define UAB(a, b, c, d) (0.5 * (UA1(a, b, c, d) - UA1(a, b, d, c)))                /* u_matrix_antisymm_both, 7943 */
/* the U-matrix factor of the worm's proposal: g[i] = flavour of worm operator i (1-based), u[i] = i-th flavour index of the
 * component (u_g; two digits in sectors 3 and 10, four otherwise; set_IEOperFlavor, Trace.F90:3835-3883).  meas != 0: the
 * expression whose sign enters the estimator, which differs from the proposal's in sector 13 (CTQMC.F90:1919-1921 vs 5764-5766) */
static double worm_ufac_of(const ora_walker *w, int sector, const int *g, int meas)
{
    const int nf = w->nflav;
    const int nred = (sector == 3 || sector == 10) ? 2 : 4;
    int u[5] = {0, 0, 0, 0, 0}, rest = w->icomponent - 1;
    for (int i = nred; i >= 1; --i) { u[i] = rest % nf; rest /= nf; }
    switch (sector) {
    case 3: case 5: return 0.5 * (UM(u[1], g[2], g[1], g[3]) - UM(g[2], u[1], g[1], g[3]));
    case 10: return 0.5 * (UM(u[1], g[2], g[1], g[3]) - UM(g[2], u[1], g[1], g[3])) * 0.5 * (UM(g[6], g[4], u[2], g[5]) - UM(g[6], g[4], g[5], u[2]));
    case 11: return 0.5 * (UM(u[1], g[2], g[1], g[3]) - UM(g[2], u[1], g[1], g[3])) * 0.5 * (UM(g[6], g[4], u[2], g[5]) - UM(g[6], g[4], g[5], u[2]))
                  * 0.5 * (UM(u[3], g[8], g[7], g[9]) - UM(g[8], u[3], g[7], g[9])) * 0.5 * (UM(g[12], g[10], u[4], g[11]) - UM(g[12], g[10], g[11], u[4]));
    case 12: return UA1(u[1], g[2], g[1], g[3]) * UA2(g[4], g[6], u[2], g[5]) * UAB(u[3], g[8], u[4], g[7]);
    case 13: return meas ? UA1(u[1], g[2], g[1], g[3]) * UA1(u[3], g[4], g[5], g[7]) * UA2(g[6], g[8], u[2], u[4])
                         : UA1(u[1], g[2], g[1], g[3]) * UA1(u[3], g[4], g[6], g[5]) * UA2(g[8], g[7], u[2], u[4]);
    case 14: return UAB(u[1], g[2], u[2], g[1]) * UAB(u[3], g[4], u[4], g[3]);
    case 15: return UA1(u[1], u[3], g[1], g[3]) * UA2(g[2], g[4], u[2], u[4]);       /* the proposal carries a minus sign on top (5791) */
    case 16: return UA1(u[1], g[2], g[1], g[3]) * UAB(u[3], g[4], u[2], u[4]);
    default: return 1.0;
    }
}
static double worm_ufac(const ora_walker *w, int meas)
{
    int g[13] = {0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) g[w->ops[i].wid] = w->ops[i].flav;
    return worm_ufac_of(w, w->isector, g, meas);
}
static double gsigma_ufac(const ora_walker *w) { return worm_ufac(w, 0); }

static double worm_prefact_of(const ora_walker *w, double ufac)
{
    const double b = w->p.beta, nd = (double)w->p.nd, nd4 = nd * nd * nd * nd, b2 = b * b;
    switch (w->isector) {
    case 3: return ufac * b2 * (nd4 * nd) * w->eta;                                  /* beta^(N-2) Nd^(N+1) eta, N = 4 (CTQMC.F90:5696) */
    case 5: return ufac * (b2 * b2) * (nd4 * nd * nd * nd) * w->eta;                 /* N = 6 */
    case 6: case 7: return b2 * nd4 * w->eta;
    case 8: case 9: return b2 * b * nd4 * w->eta;
    case 10: return ufac * b2 * (nd4 * nd4) * w->eta;                                /* beta^(N-4) Nd^(N+2) eta, N = 6 (5731) */
    case 11: return ufac * (b2 * b2) * (nd4 * nd4 * nd4) * w->eta;                   /* 5745 */
    case 12: case 13: return ufac * w->eta * (b2 * b) * (nd4 * nd4);                 /* 5757, 5768 */
    case 14: return ufac * w->eta * b2 * nd4;                                        /* 5778-5782 */
    case 15: return -ufac * w->eta * b2 * nd4;                                       /* 5791-5795 */
    case 16: return ufac * b2 * nd4 * w->eta;                                        /* 5706-5709 */
    default: return pow(b * nd, (double)worm_nops(w)) * w->eta;
    }
}
static double worm_prefact(const ora_walker *w) { return worm_prefact_of(w, worm_is_ie(w->isector) ? worm_ufac(w, 0) : 1.0); }

/* check_EqualTime (Trace.F90:4184-4269): the operators of an equal-time object must be neighbours in the list, the one named
 * second right in front of the one named first.  wid = 1-based index in wormContainer */
static int worm_index(const ora_walker *w, int wid)
{
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid == wid) return i;
    return -1;
}
static int prev_is(const ora_walker *w, int wid, int wid_prev) { const int i = worm_index(w, wid); return i > 0 && w->ops[i - 1].wid == wid_prev; }
static int check_equal_time(const ora_walker *w, int sector)
{
    switch (sector) {
    case 3: case 5: case 16: return prev_is(w, 2, 3) && prev_is(w, 3, 1);
    case 6: return prev_is(w, 1, 2) && prev_is(w, 3, 4);
    case 7: return prev_is(w, 1, 3) && prev_is(w, 2, 4);
    case 8: return prev_is(w, 3, 4);
    case 9: return prev_is(w, 2, 4);
    case 10: return prev_is(w, 2, 3) && prev_is(w, 3, 1) && prev_is(w, 6, 4) && prev_is(w, 4, 5);
    case 11: return prev_is(w, 2, 3) && prev_is(w, 3, 1) && prev_is(w, 6, 4) && prev_is(w, 4, 5)
                 && prev_is(w, 8, 9) && prev_is(w, 9, 7) && prev_is(w, 12, 10) && prev_is(w, 10, 11);
    case 12: return prev_is(w, 2, 3) && prev_is(w, 3, 1) && prev_is(w, 4, 6) && prev_is(w, 6, 5) && prev_is(w, 8, 7);
    case 13: return prev_is(w, 2, 3) && prev_is(w, 3, 1) && prev_is(w, 4, 7) && prev_is(w, 7, 5) && prev_is(w, 6, 8);
    case 14: return prev_is(w, 2, 1) && prev_is(w, 4, 3);
    case 15: return prev_is(w, 2, 4) && prev_is(w, 3, 1);
    default: return 1;
    }
}

/* StepWormAdd, CTQMC.F90:4992-5140 */
static void step_worm_add(ora_walker *w)
{
    const int nf = w->nflav, sec = w->isector;
    const int nw = worm_nops(w);
    const int ie = worm_is_ie(sec);
    int comp = w->icomponent, ncomp = 1;
    for (int i = 0; i < nw; ++i) ncomp *= nf;
    if (!ie && comp == 0) comp = RINT(w, 1, ncomp);
    w->try_wadd += 1;
    /* set_OperFlavor (Trace.F90:3804-3826): digits of comp-1 in base 2Nd, most significant first; CA = mod(i,2)+1 */
    op_t o[12]; int rest = comp - 1, base = ncomp / nf;
    for (int i = 0; i < nw; ++i) {
        o[i].ca = ((i + 1) % 2) + 1; o[i].wid = i + 1; o[i].flav = 0;
        if (!ie) { int bs = rest / base; rest -= bs * base; base /= nf; o[i].flav = bs; }
    }
    if (ie) {
        /* set_IEOperFlavor (Trace.F90:3835-3919): the operators of the equal-time objects get random flavours (orbital, then spin) until
         * the U-matrix factor of the proposal does not vanish (CTQMC.F90:5047-5054); sectors 3 and 5: the operators after the third
         * carry the component's flavours after the first */
        int dig[4] = {0, 0, 0, 0}, r2 = comp - 1;
        for (int i = 3; i >= 0; --i) { dig[i] = r2 % nf; r2 /= nf; }                /* four digits (sector 5) */
        for (;;) {
            int g[13] = {0};
            for (int i = 0; i < nw; ++i) {
                if ((sec == 3 || sec == 5) && i >= 3) o[i].flav = sec == 3 ? (comp - 1) % nf : dig[i - 2];
                else { const int orb = RINT(w, 1, w->p.nd), spin = RINT(w, 1, 2); o[i].flav = 2 * (orb - 1) + (spin - 1); }
                g[i + 1] = o[i].flav;
            }
            if (worm_prefact_of(w, worm_ufac_of(w, sec, g, 0)) != 0.0) break;
        }
    }
    /* set_OperTime (Trace.F90:3942-4164) */
    const double beta = w->p.beta, eqoff = beta * 1e-14;       /* equal_time_offset, Trace.F90:398 */
#define T(i) o[(i) - 1].tau
    if (sec == 3 || sec == 5 || sec == 10 || sec == 11) {      /* a random time for every operator, then the equal-time objects */
        for (int i = 0; i < nw; ++i) o[i].tau = U01(w) * beta;
        if (T(2) < 2.0 * eqoff) T(2) = 2.0 * eqoff;
        if (sec >= 10 && T(6) < 2.0 * eqoff) T(6) = 2.0 * eqoff;
        if (sec == 11) { if (T(8) < 2.0 * eqoff) T(8) = 2.0 * eqoff; if (T(12) < 2.0 * eqoff) T(12) = 2.0 * eqoff; }
        T(3) = T(2) - eqoff; T(1) = T(2) - 2.0 * eqoff;
        if (sec >= 10) { T(4) = T(6) - eqoff; T(5) = T(6) - 2.0 * eqoff; }
        if (sec == 11) { T(9) = T(8) - eqoff; T(7) = T(8) - 2.0 * eqoff; T(10) = T(12) - eqoff; T(11) = T(12) - 2.0 * eqoff; }
    } else if (sec == 12 || sec == 13) {
        const int third = sec == 12 ? 8 : 6;
        T(2) = U01(w) * beta; T(4) = U01(w) * beta; T(third) = U01(w) * beta;
        if (T(2) < 2.0 * eqoff) T(2) = 2.0 * eqoff;
        if (T(4) < 2.0 * eqoff) T(4) = 2.0 * eqoff;
        if (T(third) < eqoff) T(third) = eqoff;
        T(3) = T(2) - eqoff; T(1) = T(2) - 2.0 * eqoff;
        if (sec == 12) { T(6) = T(4) - eqoff; T(5) = T(4) - 2.0 * eqoff; T(7) = T(8) - eqoff; }
        else { T(7) = T(4) - eqoff; T(5) = T(4) - 2.0 * eqoff; T(8) = T(6) - eqoff; }
    } else if (sec == 14 || sec == 15) {
        const int second = sec == 14 ? 4 : 3;
        T(2) = U01(w) * beta; T(second) = U01(w) * beta;
        if (T(2) < eqoff) T(2) = eqoff;
        if (T(second) < eqoff) T(second) = eqoff;
        if (sec == 14) { T(1) = T(2) - eqoff; T(3) = T(4) - eqoff; }
        else { T(1) = T(3) - eqoff; T(4) = T(2) - eqoff; }
    } else if (sec == 16) {
        T(2) = U01(w) * beta; T(4) = U01(w) * beta;
        if (T(2) < 2.0 * eqoff) T(2) = 2.0 * eqoff;
        if (T(4) < eqoff) T(4) = eqoff;
        T(3) = T(2) - eqoff; T(1) = T(2) - 2.0 * eqoff;
    } else if (sec >= 6 && sec <= 9) {
        o[0].tau = U01(w) * beta;
        o[3].tau = U01(w) * beta;
        if (sec <= 7) {                                         /* two random times */
            if (o[0].tau < eqoff) o[0].tau = eqoff;
            if (o[3].tau > beta - eqoff) o[3].tau = beta - eqoff;
            if (sec == 6) { o[1].tau = o[0].tau - eqoff; o[2].tau = o[3].tau + eqoff; }
            else { o[1].tau = o[3].tau + eqoff; o[2].tau = o[0].tau - eqoff; }
        } else {                                                /* three random times */
            if (o[3].tau > beta - eqoff) o[3].tau = beta - eqoff;
            if (sec == 8) { o[1].tau = U01(w) * beta; o[2].tau = o[3].tau + eqoff; }
            else { o[1].tau = o[3].tau + eqoff; o[2].tau = U01(w) * beta; }
        }
    } else
    for (int i = 0; i < nw; ++i) {
        o[i].tau = U01(w) * w->p.beta;
        if (i > 0 && o[i - 1].tau == o[i].tau) o[i].tau += w->p.beta * 1e-14;
    }
#undef T
    for (int i = 0; i < nw; ++i) list_insert(w, o[i]);
    double prefact = worm_prefact(w);
    if (!check_equal_time(w, sec) || prefact == 0.0 || !sequence_valid(w, w->outer_sst)) {
        for (int i = w->nops - 1; i >= 0; --i) if (w->ops[i].wid) list_remove(w, i);
        w->accqn_wadd += 1; w->sector = 1;
        return;
    }
    logval trnew = full_trace(w, w->outer_sst);
    double rnd = U01(w);
    if (rnd < fabs(prefact * trnew.sign * w->tr.sign * exp(trnew.log - w->tr.log))) {
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_wadd += 1;
    } else {
        for (int i = w->nops - 1; i >= 0; --i) if (w->ops[i].wid) list_remove(w, i);
        w->sector = 1;
    }
}

/* StepWormRem, CTQMC.F90:5144-5249 */
static void step_worm_rem(ora_walker *w)
{
    w->try_wrem += 1;
    const double prefact = worm_prefact(w);       /* get_Proposal reads the worm operators' flavours (sector 3): before they leave the list */
    op_t saved[12]; int ns = 0;
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) saved[ns++] = w->ops[i];
    for (int i = w->nops - 1; i >= 0; --i) if (w->ops[i].wid) list_remove(w, i);
    if (!sequence_valid(w, w->outer_sst)) {
        for (int i = 0; i < ns; ++i) list_insert(w, saved[i]);
        w->accqn_wrem += 1;
        return;
    }
    logval trnew = full_trace(w, w->outer_sst);
    double rnd = U01(w);
    if (rnd < fabs(1.0 / prefact * trnew.sign * w->tr.sign * exp(trnew.log - w->tr.log))) {
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_wrem += 1;
        w->sector = 1;
    } else {
        for (int i = 0; i < ns; ++i) list_insert(w, saved[i]);
    }
}

/* StepWormReplace (CTQMC.F90:5809-6297) with propose_WormReplace / do_WormReplace (Trace.F90:3054-3478): a worm operator
 * and a hybridising operator of the same flavour and type exchange their roles.  If the worm operator belongs to an
 * equal-time pair (sectors 6 - 9), its partner moves along to the hybridising operator's time (`multimove`, 6068-6127,
 * 6183-6249): the pair then sits at (tau_h, tau_h + offset) in the order it had, the local trace changes. */
static void step_worm_replace(ora_walker *w)
{
    w->try_wrep += 1;
    /* which worm operator becomes a hybridising one (CTQMC.F90:5834-5946) */
    int r;
    {
        int g[13] = {0};
        for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) g[w->ops[i].wid] = w->ops[i].flav;
        switch (w->sector) {
        case 5: case 16: r = RINT(w, 4, worm_nops(w)); break;       /* only the operators outside the equal-time object */
        case 10:                                                   /* a three-operator object that holds a density moves the density */
            if (g[2] == g[3]) { if (g[4] == g[5]) { r = RINT(w, 1, 2); r = (r - 1) * 5 + 1; } else r = 1; }
            else { if (g[4] == g[5]) r = 6; else r = RINT(w, 1, 6); }
            break;
        case 12:
            if (g[2] == g[3]) { if (g[6] == g[5]) { r = RINT(w, 1, 2); r = (r - 1) * 3 + 1; } else r = 1; }
            else { if (g[6] == g[5]) r = 4; else r = RINT(w, 1, 6); }
            break;
        case 11: {
            int nden = (g[2] == g[3]) + (g[4] == g[5]) + (g[8] == g[9]) + (g[10] == g[11]);
            if (nden > 0) {
                int pick = RINT(w, 1, nden); r = 0;
                for (int q = 0; q < 2 && !r; ++q) {
                    if (g[6 * q + 2] == g[6 * q + 3]) { if (pick <= 1) { r = 6 * q + 1; break; } pick -= 1; }
                    if (g[6 * q + 4] == g[6 * q + 5]) { if (pick <= 1) { r = 6 * q + 6; break; } pick -= 1; }
                }
            } else r = RINT(w, 1, 12);
            break; }
        case 13: return;                                           /* no replacement move in this sector (5944-5946) */
        default: r = RINT(w, 1, worm_nops(w)); break;
        }
    }
    /* partners that move along, and the places (1 = first, 2 = second, 3 = third in time) of the partners / of the new worm operator in
     * the equal-time object at the hybridising operator's time (qrepind / qtimeind, CTQMC.F90:5997-6178) */
    int npart = 0, partner[2] = {0, 0}, pos_partner[2] = {0, 0}, pos_new = 0;
    {
        static const int p2[4] = {2, 1, 4, 3}, p2pp[4] = {3, 4, 1, 2};
        /* three-operator objects: rows = the operator that stays (1..6 of a QQ pair of objects), {partner, place, partner, place, place of the new one} */
        static const int q3[6][5] = {{3, 2, 2, 3, 1}, {1, 1, 3, 2, 3}, {1, 1, 2, 3, 2}, {5, 1, 6, 3, 2}, {4, 2, 6, 3, 1}, {5, 1, 4, 2, 3}};
        static const int q3n[6][5] = {{3, 2, 2, 3, 1}, {1, 1, 3, 2, 3}, {1, 1, 2, 3, 2}, {5, 1, 6, 2, 3}, {6, 2, 4, 3, 1}, {5, 1, 4, 3, 2}};   /* sector 12 */
        const int (*tab)[5] = NULL; int row = 0, shift = 0;
        switch (w->sector) {
        case 3: if (r <= 3) { tab = q3; row = r - 1; } break;
        case 10: tab = q3; row = r - 1; break;
        case 11: tab = q3; row = (r - 1) % 6; shift = 6 * ((r - 1) / 6); break;
        case 12: tab = q3n; row = r - 1; break;
        case 6: npart = 1; partner[0] = p2[r - 1]; pos_partner[0] = (r & 1) ? 1 : 2; break;
        case 7: npart = 1; partner[0] = p2pp[r - 1]; pos_partner[0] = (r <= 2) ? 1 : 2; break;
        case 8: if (r > 2) { npart = 1; partner[0] = p2[r - 1]; pos_partner[0] = (r == 3) ? 1 : 2; } break;
        case 9: if ((r & 1) == 0) { npart = 1; partner[0] = p2pp[r - 1]; pos_partner[0] = (r == 2) ? 1 : 2; } break;
        case 14: npart = 1; partner[0] = p2[r - 1]; pos_partner[0] = (r & 1) ? 2 : 1; break;          /* 6151-6162 */
        case 15: npart = 1; partner[0] = p2pp[r - 1]; pos_partner[0] = (r == 1 || r == 4) ? 2 : 1; break;   /* 6133-6144 */
        default: break;
        }
        if (tab) {
            npart = 2; partner[0] = tab[row][0] + shift; pos_partner[0] = tab[row][1]; partner[1] = tab[row][2] + shift; pos_partner[1] = tab[row][3];
            pos_new = tab[row][4];
        }
        if (npart == 1) pos_new = 3 - pos_partner[0];
    }
    int iw = -1;
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid == r) iw = i;
    const int f = w->ops[iw].flav, ca = w->ops[iw].ca, k = w->k[f];
    if (k == 0) return;
    int pos = RINT(w, 1, k);                   /* operPos-th hybridising operator of that flavour and type */
    int ih = -1, cnt = 0;
    for (int i = 0; i < w->nops; ++i)
        if (w->ops[i].wid == 0 && w->ops[i].flav == f && w->ops[i].ca == ca) { if (++cnt == pos) { ih = i; break; } }
    const double tw = w->ops[iw].tau, th = w->ops[ih].tau;
    const double *ft = w->ftab + (size_t)f * w->p.nftau;
    int ncnt = 0;
    for (int i = 0; i < w->nops; ++i) {
        const op_t *e = &w->ops[i];
        if (e->wid == 0 && e->flav == f && e->ca == ca && e->tau > fmin(th, tw) && e->tau < fmax(th, tw)) ++ncnt;
    }
    if (tw < th) ncnt = -ncnt;
    double *u = (double *)calloc((size_t)k, sizeof(double)), *v = (double *)calloc((size_t)k, sizeof(double));
    double *x = (double *)calloc((size_t)k, sizeof(double)), *y = (double *)calloc((size_t)k, sizeof(double));
    const double *M = w->M[f];
    if (ca == 1) {          /* creator: a column of F(tau_a - tau_c) changes */
        v[pos - 1] = 1.0;
        int q = 0;
        for (int i = 0; i < w->nops; ++i) {
            const op_t *e = &w->ops[i];
            if (e->wid || e->flav != f || e->ca != 2) continue;
            u[q++] = -ora_hybr(ft, w->p.nftau, w->p.beta, e->tau - th) + ora_hybr(ft, w->p.nftau, w->p.beta, e->tau - tw);
        }
    } else {                /* annihilator: a row changes */
        u[pos - 1] = 1.0;
        int q = 0;
        for (int i = 0; i < w->nops; ++i) {
            const op_t *e = &w->ops[i];
            if (e->wid || e->flav != f || e->ca != 1) continue;
            v[q++] = -ora_hybr(ft, w->p.nftau, w->p.beta, th - e->tau) + ora_hybr(ft, w->p.nftau, w->p.beta, tw - e->tau);
        }
    }
    /* Mat(creator, annihilator): res = Mat u (u over annihilators), detRat = 1 + vdag . res (vdag over creators) */
    for (int c = 0; c < k; ++c) { double a = 0.0; for (int j = 0; j < k; ++j) a += M[c * k + j] * u[j]; x[c] = a; }
    double detrat = 1.0;
    for (int c = 0; c < k; ++c) detrat += v[c] * x[c];
    double prob = fabs(detrat);
    logval trnew = w->tr;
    op_t old_partner[2], old_hyb;
    if (npart) {
        const double eqoff = w->p.beta * 1e-14;
        old_hyb = w->ops[ih];
        for (int q = 0; q < npart; ++q) old_partner[q] = w->ops[worm_index(w, partner[q])];
        for (int q = 0; q < npart; ++q) list_remove(w, worm_index(w, partner[q]));
        for (int i = 0; i < w->nops; ++i)
            if (w->ops[i].wid == 0 && w->ops[i].flav == f && w->ops[i].ca == ca && w->ops[i].tau == th) { list_remove(w, i); break; }
        w->ops[worm_index(w, r)].wid = 0;                       /* stays where it is, as a hybridising operator if accepted */
        op_t nn[3]; int npos[3];
        for (int q = 0; q < npart; ++q) { nn[q] = old_partner[q]; npos[q] = pos_partner[q]; }
        nn[npart] = old_hyb; nn[npart].wid = r; npos[npart] = pos_new;
        for (int pos = 1; pos <= npart + 1; ++pos)               /* in time order: tau_h, tau_h + offset, tau_h + 2 offset */
            for (int q = 0; q <= npart; ++q) if (npos[q] == pos) { nn[q].tau = th + (double)(pos - 1) * eqoff; list_insert(w, nn[q]); }
        int ok = check_equal_time(w, w->sector) && sequence_valid(w, w->outer_sst);
        if (ok) {
            trnew = full_trace(w, w->outer_sst);
            prob = fabs(detrat * trnew.sign * w->tr.sign * exp(trnew.log - w->tr.log));
        }
        if (!ok || !(U01(w) < prob)) {
            /* back to the old operators (6224-6234, 6281-6290); a failed check consumes no random number */
            for (int q = 0; q < npart; ++q) list_remove(w, worm_index(w, partner[q]));
            list_remove(w, worm_index(w, r));
            for (int i = 0; i < w->nops; ++i)
                if (w->ops[i].wid == 0 && w->ops[i].flav == f && w->ops[i].ca == ca && w->ops[i].tau == tw) w->ops[i].wid = r;
            list_insert(w, old_hyb);
            for (int q = 0; q < npart; ++q) list_insert(w, old_partner[q]);
            free(u); free(v); free(x); free(y);
            return;
        }
    }
    if (npart || U01(w) < prob) {
        for (int j = 0; j < k; ++j) { double a = 0.0; for (int c = 0; c < k; ++c) a += M[c * k + j] * v[c]; y[j] = a; }
        double *Mn = (double *)malloc(sizeof(double) * (size_t)k * k);
        for (int c = 0; c < k; ++c) for (int j = 0; j < k; ++j) Mn[c * k + j] = M[c * k + j] - x[c] * y[j] / detrat;
        /* the exchanged operator sits at another rank inside its flavour: move its row (creator) / column (annihilator) */
        int from = pos - 1, to = pos - 1 + ncnt;
        if (from != to) {
            double *T = (double *)malloc(sizeof(double) * (size_t)k * k);
            for (int c = 0; c < k; ++c) for (int j = 0; j < k; ++j) {
                int src = (ca == 1) ? c : j, dst;
                if (src == from) dst = to;
                else if (from < to) dst = (src > from && src <= to) ? src - 1 : src;
                else dst = (src >= to && src < from) ? src + 1 : src;
                if (ca == 1) T[dst * k + j] = Mn[c * k + j]; else T[c * k + dst] = Mn[c * k + j];
            }
            free(Mn); Mn = T;
        }
        free(w->M[f]); w->M[f] = Mn;
        w->det.log += log(fabs(detrat));
        w->det.sign *= (detrat < 0.0 ? -1.0 : 1.0) * ((ncnt & 1) ? -1.0 : 1.0);
        if (!npart) { w->ops[ih].wid = r; w->ops[iw].wid = 0; }
        w->tr = trnew;
        w->cfgsign = -get_sign(w);
        w->acc_wrep += 1;
    }
    free(u); free(v); free(x); free(y);
}

/* one iteration of the sampling loop: Z space CTQMC.F90:8431-8488 (== warm-up 8213-8257 without worm insertion),
 * worm space 8489-8528; window shift 8530-8532 */
static void one_step(ora_walker *w)
{
    if (w->sector == 1) {
        w->cnt_sampling += 1;
        double r = U01(w);
        double pg = w->p.p_glob, ps = w->p.p_shift, pw = (w->isector != 1) ? w->p.p_worm_ins : 0.0;
        if (r < (1.0 - ps - pg - pw) / 2.0) {
            pair_move(w, 1);
        } else if (r <= 1.0 - ps - pg - pw) {
            pair_move(w, 0);
        } else if (r <= 1.0 - ps - pg) {
            w->sector = w->isector;
            step_worm_add(w);
        } else if (r <= 1.0 - ps) {
            step_glob(w);
        } else {
            step_shift(w);
        }
    } else {
        w->cnt_sampling_w += 1;
        double r = U01(w);
        double prep = w->p.p_worm_rep, pins = w->p.p_worm_ins;
        if (r < (1.0 - prep - pins) / 2.0) pair_move(w, 1);
        else if (r < 1.0 - prep - pins) pair_move(w, 0);
        else if (r < 1.0 - pins) step_worm_replace(w);
        else step_worm_rem(w);
    }
    long tries = (long)(w->try_add + w->try_rem + w->wtry_add + w->wtry_rem);
    if (tries % w->nslide == 0) shift_window(w);
}

/* ------------------------------------------------------------------------------------------------------- */
/* driver                                                                                                    */
/* ------------------------------------------------------------------------------------------------------- */

static void init_counters(ora_walker *w) /* CTQMC.F90:847-896 */
{
    w->try_add = w->acc_add = w->accqn_add = w->try_rem = w->acc_rem = w->accqn_rem = 0;
    w->try_glob = w->acc_glob = w->accqn_glob = w->try_shift = w->acc_shift = 0;
    w->cnt_sampling = w->cnt_meas = 0;
    w->try_add4 = w->acc_add4 = w->try_rem4 = w->acc_rem4 = 0; w->try_flavc = w->acc_flavc = 0;
    w->try_add_outer = w->acc_add_outer = w->try_rem_outer = w->acc_rem_outer = 0;
    w->wtry_add = w->wacc_add = w->wtry_rem = w->wacc_rem = 0;
    w->try_wadd = w->acc_wadd = w->accqn_wadd = w->try_wrem = w->acc_wrem = w->accqn_wrem = w->try_wrep = w->acc_wrep = 0;
    w->cnt_sampling_w = w->cnt_meas_w = 0;
    memset(w->accpair, 0, sizeof(int64_t) * (size_t)w->nflav * w->naccpair);
    w->apt_index = 0;
}

/* AccPair back to Ntau bins over beta (ctqmc_warmup / ctqmc_measure, CTQMC.F90:8199-8205, 8387-8393) */
static void accpair_default(ora_walker *w)
{
    if (w->naccpair != w->p.ntau) {
        free(w->accpair);
        w->accpair = (int64_t *)calloc((size_t)w->nflav * w->p.ntau, sizeof(int64_t));
    }
    w->naccpair = w->p.ntau; w->accpair_max = w->p.beta;
}

ora_walker *ora_create(const ora_params *p, const double *U, const double *ftau_full, const double *mu_full, ora_rng *rng)
{
    ora_walker *w = (ora_walker *)calloc(1, sizeof(ora_walker));
    w->p = *p; w->rng = rng;
    w->nflav = 2 * p->nd; w->nstates = 1 << (2 * p->nd);
    if (w->nflav > MAXFLAV) { free(w); return NULL; }
    { size_t n4 = (size_t)w->nflav * w->nflav * w->nflav * w->nflav; w->umat = (double *)malloc(sizeof(double) * n4); memcpy(w->umat, U, sizeof(double) * n4); }
    w->lp = ora_locprob_build(p->nd, U, mu_full, p->qns, p->epsevec, p->epsblock);
    if (!w->lp || w->lp->nsmax > 64) { ora_locprob_free(w->lp); free(w); return NULL; }
    w->nsst = w->lp->nsst;
    w->E = w->lp->eval;
    w->Egs = w->lp->egs;                                                            /* Trace.F90:576-583 */
    w->ptr = (logval *)calloc((size_t)w->lp->nsmax, sizeof(logval));
    w->dmoff = (int *)calloc((size_t)w->nsst + 1, sizeof(int));
    for (int b = 0; b < w->nsst; ++b) w->dmoff[b + 1] = w->dmoff[b] + w->lp->sst_n[b] * w->lp->sst_n[b];
    w->dmsize = w->dmoff[w->nsst];
    w->dmblk = (double *)calloc((size_t)w->dmsize, sizeof(double));
    int nd = p->nd, nft = p->nftau;
    w->ftab = (double *)malloc(sizeof(double) * (size_t)w->nflav * nft);
    for (int b = 0; b < nd; ++b) for (int s = 0; s < 2; ++s)                        /* CTQMC.F90:7950-7955 */
        memcpy(w->ftab + (size_t)(2 * b + s) * nft,
               ftau_full + ((((size_t)b * 2 + s) * nd + b) * 2 + s) * nft, sizeof(double) * (size_t)nft);
    for (int f = 0; f < w->nflav; ++f) { w->M[f] = (double *)malloc(sizeof(double)); w->k[f] = 0; }
    w->Mf = (double *)malloc(sizeof(double)); w->kt = 0;
    if (p->offdiag) {      /* HybrF_full(b1,s1,b2,s2,:) as handed over (CTQMC.F90:7950-7955) */
        const size_t nn = (size_t)w->nflav * w->nflav * nft;
        w->ffull = (double *)malloc(sizeof(double) * nn); memcpy(w->ffull, ftau_full, sizeof(double) * nn);
        w->gleg_full = (double *)calloc((size_t)w->nflav * w->nflav * p->nlegmax, sizeof(double));
        w->gtau_full = (double *)calloc((size_t)w->nflav * w->nflav * p->ntau, sizeof(double));
    }
    w->accpair = (int64_t *)calloc((size_t)w->nflav * p->ntau, sizeof(int64_t));
    w->naccpair = p->ntau; w->accpair_max = p->beta;
    w->gleg = (double *)calloc((size_t)w->nflav * p->nlegmax, sizeof(double));
    w->gtau = (double *)calloc((size_t)w->nflav * p->ntau, sizeof(double));
    w->ntau_n0 = p->meas_susz ? (double *)calloc((size_t)w->nflav * w->nflav * (p->ntau + 1), sizeof(double)) : NULL;
    w->gms = w->gmid = w->sstep = NULL;
    if (p->step_series && p->nmeas > 0) {
        w->gms = (double *)calloc((size_t)w->nflav * p->nmeas, sizeof(double)); w->gmid = (double *)calloc((size_t)w->nflav * p->nmeas, sizeof(double));
        w->sstep = (double *)calloc((size_t)p->nmeas, sizeof(double));
    }
    w->giw = (double *)calloc((size_t)w->nflav * 2 * p->niw * 2, sizeof(double));
    w->histo = (double *)calloc((size_t)w->nflav * (p->maxhisto + 1), sizeof(double));
    w->dm = (double *)calloc((size_t)w->nstates, sizeof(double));
    w->outer_histo = (double *)calloc((size_t)w->nstates, sizeof(double));
    w->sign_sst = (double *)calloc((size_t)w->nstates, sizeof(double)); w->tcontrib_sst = (double *)calloc((size_t)w->nstates, sizeof(double));
    w->outer_state_histo = (double *)calloc((size_t)w->nstates, sizeof(double));
    w->sign_state = (double *)calloc((size_t)w->nstates, sizeof(double)); w->tcontrib_state = (double *)calloc((size_t)w->nstates, sizeof(double));
    if (p->fourpnt & 4) {                                                           /* CTQMC.F90:552-559, 838-840 */
        size_t n2 = 2 * (size_t)(p->n4iwf + p->n4iwb), n4 = (size_t)w->nflav * w->nflav * 4 * p->n4iwf * p->n4iwf * (2 * p->n4iwb + 1) * 2;
        w->g4iw = (double *)calloc(n4, sizeof(double));
        if (p->meas_g4iwpp) w->g4iwpp = (double *)calloc(n4, sizeof(double));
        if (p->meas_g2iw) w->g2iw = (double *)calloc((size_t)w->nflav * n2 * n2 * 2, sizeof(double));
    }
    w->sector = w->isector = 1; w->icomponent = 0; w->eta = p->worm_eta;
    if (p->accum_gtau == 2) {                                                       /* CTQMC.F90:497-500, ilog2 Accumulators.F90:89-101 */
        int lev = -1; for (int64_t t = p->nmeas; t > 0; t /= 2) ++lev;
        if (lev < 1) { fprintf(stderr, "oracle: AccumGtau = 2 needs Nmeas >= 2\n"); abort(); }
        w->aclev = lev; w->acn = 0;
        w->acbuf = (double *)calloc((size_t)w->nflav * p->ntau, sizeof(double));
        w->acdat = (double *)calloc((size_t)w->nflav * p->ntau * lev * 3, sizeof(double));
    }
    if (p->fourpnt & 1) w->g4tau = (double *)calloc((size_t)w->nflav * w->nflav * p->n4tau * p->n4tau * p->n4tau, sizeof(double));   /* CTQMC.F90:544-547 */
    if (p->fourpnt & 2) {                                                           /* 548-551: allocated with N4tau, filled up to N4leg */
        if (p->n4leg > p->n4tau || p->n4leg < 2) { fprintf(stderr, "oracle: FourPnt & 2 needs 2 <= N4leg <= N4tau\n"); abort(); }
        w->g4leg = (double *)calloc((size_t)w->nflav * w->nflav * p->n4tau * p->n4tau * (2 * p->n4iwb + 1) * 2, sizeof(double));
    }
    if (p->worm_meas_giw) w->giww = (double *)calloc((size_t)4 * p->niw, sizeof(double));      /* CTQMC.F90:597-599 */
    if (p->worm_meas_gtau) w->gtauw = (double *)calloc((size_t)p->ntau, sizeof(double));        /* CTQMC.F90:609-612 */
    if (p->worm_meas_gsigma) w->gsw = (double *)calloc((size_t)4 * p->niw, sizeof(double));     /* CTQMC.F90:619-623 */
    for (int q = 0; q < 6; ++q) {                                                   /* CTQMC.F90:654-726 */
        w->weq[q] = NULL; w->weq_n[q] = 0;
        if (!p->worm_meas_eq[q]) continue;
        w->weq_n[q] = q < 2 ? (size_t)2 * (2 * p->n2iwb + 1) : (q < 4 ? (size_t)p->ntau : (size_t)2 * 2 * p->n3iwf * (2 * p->n3iwb + 1));
        w->weq[q] = (double *)calloc(w->weq_n[q], sizeof(double));
    }
    for (int q = 0; q < 11; ++q) {                                                  /* CTQMC.F90:379-430, 624-652 */
        const size_t n4 = (size_t)4 * p->n4iwf * p->n4iwf * (2 * p->n4iwb + 1) * 2;
        const size_t sizes[11] = {n4, (size_t)4 * p->niw, (size_t)p->ntau, n4, (size_t)2 * 2 * p->n3iwf * (2 * p->n3iwb + 1), (size_t)2 * 4 * p->n3iwf * p->n3iwf,
                                  (size_t)2 * (2 * p->n2iwb + 1), (size_t)p->ntau, (size_t)2 * (2 * p->n2iwb + 1), (size_t)p->ntau, (size_t)4 * p->niw};
        w->wmore[q] = NULL; w->wmore_n[q] = 0;
        if (!p->worm_meas_more[q] || ((q == 0 || q == 3) && !(p->fourpnt & 8))) continue;
        w->wmore_n[q] = sizes[q];
        w->wmore[q] = (double *)calloc(sizes[q], sizeof(double));
    }
    if ((p->fourpnt & 8) && p->worm_meas_g4iw) {                                    /* CTQMC.F90:339-361 */
        size_t n = (size_t)4 * p->n4iwf * p->n4iwf * (2 * p->n4iwb + 1) * 2;
        if (p->g4ph) w->g4w = (double *)calloc(n, sizeof(double));
        if (p->g4pp) w->g4wpp = (double *)calloc(n, sizeof(double));
    }
    w->det.log = 0.0; w->det.sign = 1.0;
    w->cfgsign = -1.0;                                                              /* Trace.F90:129 */
    init_window(w, p->taudiffmax);                                                  /* Trace.F90:605 */
    /* initial outer superstate ~ empty-trace weight, Trace.F90:644-665 */
    w->outer_sst = draw_outer(w);
    w->tr = full_trace(w, w->outer_sst);
    {   /* CTQMC.F90:8069 */
        int a = 2 * nd, b = (int)floor((double)p->ncorr / (double)w->num_windows);
        w->nslide = a > b ? a : b;
    }
    init_counters(w);
    return w;
}

void ora_destroy(ora_walker *w)
{
    if (!w) return;
    for (int f = 0; f < w->nflav; ++f) free(w->M[f]);
    free(w->Mf); free(w->ffull); free(w->gleg_full); free(w->gtau_full);
    free(w->ops); free(w->ftab); ora_locprob_free(w->lp); free(w->ptr); free(w->dmblk); free(w->dmoff); free(w->accpair); free(w->accpairtau); free(w->gleg); free(w->gtau); free(w->giw); free(w->ntau_n0); free(w->gms); free(w->gmid); free(w->sstep);
    free(w->histo); free(w->dm); free(w->outer_histo); free(w->sign_sst); free(w->tcontrib_sst); free(w->outer_state_histo); free(w->sign_state); free(w->tcontrib_state); free(w->g2iw); free(w->g4iw); free(w->g4iwpp); free(w->g4w); free(w->g4wpp); free(w->g4tau); free(w->g4leg); free(w->giww); free(w->gtauw); free(w->gsw); free(w->umat); for (int q = 0; q < 6; ++q) free(w->weq[q]); for (int q = 0; q < 11; ++q) free(w->wmore[q]); free(w->acbuf); free(w->acdat); free(w);
}

void ora_set_taudiffmax(ora_walker *w, double x) { init_window(w, x); }

void ora_warmup(ora_walker *w, int64_t n)
{
    int isec = w->isector;
    accpair_default(w);
    init_counters(w);
    w->isector = 1;                       /* ctqmc_warmup never inserts worms (CTQMC.F90:8213-8257) */
    for (int64_t i = 0; i < n; ++i) one_step(w);
    w->cnt_sampling = 0;                  /* the warm-up loop does not count sampling steps */
    w->isector = isec;
}

void ora_measure_begin(ora_walker *w) { accpair_default(w); init_counters(w); }

void ora_init_counters(ora_walker *w) { init_counters(w); }

/* ctqmc_calibrate (CTQMC.F90:8088-8178): the Z-space move loop without window shifts and without step counting.
 * Phase 1 (list != 0) ends at the first multiple of min(1000, nprogress) steps at which phase1pairnum accepted pair
 * distances are on the list; phase 2 at the first such multiple at which every flavour has phase2pairnum accepted
 * pairs in the histogram of phase2taugrid bins over [0, taudiff_max].  Returns the number of steps. */
int64_t ora_calibrate(ora_walker *w, int list, int phase1pairnum, int phase2taugrid, int phase2pairnum, int nprogress)
{
    if (list) {
        if (!w->accpairtau) { w->accpairtau = (double *)calloc((size_t)phase1pairnum, sizeof(double)); w->napt = phase1pairnum; }
    } else {
        free(w->accpairtau); w->accpairtau = NULL; w->napt = 0;
        free(w->accpair);
        w->accpair = (int64_t *)calloc((size_t)w->nflav * phase2taugrid, sizeof(int64_t));
        w->naccpair = phase2taugrid; w->accpair_max = w->taudiff_max;
    }
    const int chunk = nprogress < 1000 ? nprogress : 1000;
    const double pg = w->p.p_glob, ps = w->p.p_shift;
    int64_t i = 0;
    for (;;) {
        ++i;
        double r = U01(w);
        if (r < (1.0 - ps - pg) / 2.0) pair_move(w, 1);
        else if (r <= 1.0 - ps - pg) pair_move(w, 0);
        else if (r <= 1.0 - ps) step_glob(w);
        else step_shift(w);
        if (i % chunk == 0) {
            if (list) { if (w->apt_index >= w->napt) break; }
            else {
                int done = 1;
                for (int f = 0; f < w->nflav && done; ++f) {
                    int64_t sum = 0;
                    for (int b = 0; b < w->naccpair; ++b) sum += w->accpair[(size_t)f * w->naccpair + b];
                    if (sum < phase2pairnum) done = 0;
                }
                if (done) break;
            }
        }
    }
    return i;
}

int ora_get_accpairtau(const ora_walker *w, double *o) { if (!w->accpairtau) return 0; memcpy(o, w->accpairtau, sizeof(double) * (size_t)w->napt); return w->napt; }
int ora_naccpair(const ora_walker *w) { return w->naccpair; }

void ora_sweep(ora_walker *w, int64_t nsteps)
{
    for (int64_t i = 0; i < nsteps; ++i) one_step(w);
}

/* measure_g4iw_nfft with the exact-sum transform (CTQMC.F90:2130-2234, Nfft_z.F90:416-453) */
static void meas_g4iw(ora_walker *w)
{
    const int nf = w->nflav, nwf = w->p.n4iwf, nwb = w->p.n4iwb, n2 = 2 * (nwf + nwb), h = n2 / 2;
    const double beta = w->p.beta;
    double *g2 = (double *)calloc((size_t)nf * n2 * n2 * 2, sizeof(double));     /* [f][wa][wc] */
    for (int f = 0; f < nf; ++f) {
        int k = w->k[f];
        if (k == 0) continue;
        double *tc = (double *)malloc(sizeof(double) * k), *ta = (double *)malloc(sizeof(double) * k);
        int nc = 0, na = 0;
        for (int i = 0; i < w->nops; ++i) {
            if (w->ops[i].flav != f || w->ops[i].wid) continue;
            if (w->ops[i].ca == 1) tc[nc++] = w->ops[i].tau; else ta[na++] = w->ops[i].tau;
        }
        for (int ifc = 0; ifc < n2; ++ifc) for (int ifa = 0; ifa < n2; ++ifa) {
            double fc = M_PI / beta * (double)(2 * (ifc + 1) - n2 - 1), fa = M_PI / beta * (double)(2 * (ifa + 1) - n2 - 1);
            double re = 0.0, im = 0.0;
            for (int ic = 0; ic < k; ++ic) for (int ia = 0; ia < k; ++ia) {
                double m = w->M[f][ic * k + ia];
                double cr = cos(-fc * tc[ic]), ci = sin(-fc * tc[ic]), ar = cos(fa * ta[ia]), ai = sin(fa * ta[ia]);
                double pr = m * cr, pi = m * ci;
                re += pr * ar - pi * ai; im += pr * ai + pi * ar;
            }
            g2[(((size_t)f * n2 + ifa) * n2 + ifc) * 2] = re; g2[(((size_t)f * n2 + ifa) * n2 + ifc) * 2 + 1] = im;
        }
        free(tc); free(ta);
    }
    if (w->g2iw) for (size_t i = 0; i < (size_t)nf * n2 * n2 * 2; ++i) w->g2iw[i] += g2[i];
    /* this_g2iw(o,s,a,c) with frequency indices -n2/2 .. n2/2-1 */
#define G2(f, a, c) (&g2[(((size_t)(f) * n2 + ((a) + h)) * n2 + ((c) + h)) * 2])
    const int nb = 2 * nwb + 1, nw = 2 * nwf;
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2)
        for (int w1 = -nwf; w1 < nwf; ++w1) for (int w2 = -nwf; w2 < nwf; ++w2) for (int wb = -nwb; wb <= nwb; ++wb) {
            size_t idx = (((((size_t)f1 * nf + f2) * nw + (w1 + nwf)) * nw + (w2 + nwf)) * nb + (wb + nwb)) * 2;
            if (w->g4iw) {
                const double *a, *b;
                if (w->p.zphconv == 0) { a = G2(f1, w1 + wb, w1); b = G2(f2, w2, w2 + wb); }
                else { a = G2(f1, w1, w1 - wb); b = G2(f2, w2 - wb, w2); }
                double re = a[0] * b[0] - a[1] * b[1], im = a[0] * b[1] + a[1] * b[0];
                if (f1 == f2) {
                    const double *c, *d;
                    if (w->p.zphconv == 0) { c = G2(f1, w2, w1); d = G2(f1, w1 + wb, w2 + wb); }
                    else { c = G2(f1, w1, w2); d = G2(f1, w2 - wb, w1 - wb); }
                    re -= c[0] * d[0] - c[1] * d[1]; im -= c[0] * d[1] + c[1] * d[0];
                }
                w->g4iw[idx] += re; w->g4iw[idx + 1] += im;
            }
            if (w->g4iwpp) {
                const double *a = G2(f1, wb - w2 - 1, w1), *b = G2(f2, w2, wb - w1 - 1);
                double re = a[0] * b[0] - a[1] * b[1], im = a[0] * b[1] + a[1] * b[0];
                if (f1 == f2) {
                    const double *c = G2(f1, w2, w1), *d = G2(f1, wb - w2 - 1, wb - w1 - 1);
                    re -= c[0] * d[0] - c[1] * d[1]; im -= c[0] * d[1] + c[1] * d[0];
                }
                w->g4iwpp[idx] += re; w->g4iwpp[idx + 1] += im;
            }
        }
#undef G2
    free(g2);
}

/* MeasDensityMatrix_beta_half, CTQMC.F90:3088-3151 with update_b2_states_eb, Trace.F90:6213-6470: for every state of
 * the outer block the ket is propagated from 0 to beta/2 through the operators with tau <= beta/2, the bra from beta
 * down to beta/2; rho(bra index, ket index) += -cfgsign * sign(tr) * bra x ket * exp(lk + lb - log|tr|) */
static void meas_densitymatrix(ora_walker *w)
{
    const ora_locprob *lp = w->lp;
    const double b2 = w->p.beta / 2.0;
    int outer = w->outer_sst, nout = lp->sst_n[outer];
    if (w->nops == 0) {
        full_trace_part(w, outer, w->ptr);           /* parttrace of the current (accepted) configuration */
        for (int st = 0; st < nout; ++st) {
            double r = w->ptr[st].sign * w->tr.sign * exp(w->ptr[st].log - w->tr.log);
            w->dmblk[w->dmoff[outer] + st * nout + st] -= w->cfgsign * r;
        }
        return;
    }
    int nk = 0; while (nk < w->nops && w->ops[nk].tau <= b2) ++nk;
    for (int st = 0; st < nout; ++st) {
        double ket[64], bra[64], lk = 0.0, lb = 0.0;
        for (int i = 0; i < nout; ++i) ket[i] = bra[i] = (i == st) ? 1.0 : 0.0;
        int s = outer; double tp = 0.0;
        for (int i = 0; i < nk; ++i) { s = propagate(w, s, w->ops[i].tau - tp, w->ops[i].flav, w->ops[i].ca, ket, &lk); tp = w->ops[i].tau; }
        {
            int n = lp->sst_n[s]; const double *E = lp->eval + lp->sst_off[s];
            for (int i = 0; i < n; ++i) ket[i] *= exp(-(E[i] - E[0]) * (b2 - tp));
            split_log(ket, n, &lk);
            lk += -(E[0] - w->Egs) * (b2 - tp);
        }
        int sk = s;
        s = outer; tp = w->p.beta;
        for (int i = w->nops - 1; i >= nk; --i) {     /* bra: the operator acts with the other type (Trace.F90:5927) */
            s = propagate(w, s, tp - w->ops[i].tau, w->ops[i].flav, 3 - w->ops[i].ca, bra, &lb); tp = w->ops[i].tau;
        }
        {
            int n = lp->sst_n[s]; const double *E = lp->eval + lp->sst_off[s];
            for (int i = 0; i < n; ++i) bra[i] *= exp(-(E[i] - E[0]) * (tp - b2));
            split_log(bra, n, &lb);
            lb += -(E[0] - w->Egs) * (tp - b2);
        }
        if (s != sk) { fprintf(stderr, "beta/2 blocks differ\n"); abort(); }
        int n = lp->sst_n[s];
        double fac = -w->cfgsign * w->tr.sign * exp(lk + lb - w->tr.log);
        for (int i1 = 0; i1 < n; ++i1) for (int i2 = 0; i2 < n; ++i2)
            w->dmblk[w->dmoff[s] + i1 * n + i2] += bra[i1] * fac * ket[i2];
    }
}

/* MeasGtau4PntLeg (CTQMC.F90:2609-2763): for every annihilator a1 and creator c1 of one flavour and every creator c2 and
 * annihilator a2 of a (possibly different) flavour, M = M1(c1,a1) M2(c2,a2) - [same flavour] M(c2,a1) M(c1,a2) with the
 * fermionic sign of the two time differences goes to the tau bin (tau12, tau34, tau14) and, times P_l(x12) P_l'(x34)
 * exp(i w_m tau14), to the mixed Legendre / Matsubara representation (Boehnke 2011).  No configuration sign. */
static void meas_g4leg(ora_walker *w)
{
    const int nf = w->nflav, nt = w->p.n4tau, nl = w->p.n4leg, nwb = w->p.n4iwb, nw = 2 * nwb + 1;
    const double beta = w->p.beta;
    double *tc[MAXFLAV], *ta[MAXFLAV];
    for (int f = 0; f < nf; ++f) {
        int k = w->k[f], nc = 0, na = 0;
        tc[f] = (double *)malloc(sizeof(double) * (size_t)(k + 1)); ta[f] = (double *)malloc(sizeof(double) * (size_t)(k + 1));
        for (int i = 0; i < w->nops; ++i) {
            if (w->ops[i].flav != f || w->ops[i].wid) continue;
            if (w->ops[i].ca == 1) tc[f][nc++] = w->ops[i].tau; else ta[f][na++] = w->ops[i].tau;
        }
    }
    double *leg12 = (double *)malloc(sizeof(double) * (size_t)(nl + 2)), *leg34 = (double *)malloc(sizeof(double) * (size_t)(nl + 2));
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) {
        const int k1 = w->k[f1], k2 = w->k[f2];
        const double *M1 = w->M[f1], *M2 = w->M[f2];
        for (int a1 = 0; a1 < k1; ++a1) for (int c1 = 0; c1 < k1; ++c1) {
            double tau12 = ta[f1][a1] - tc[f1][c1];
            const double sgn12 = tau12 < 0.0 ? -1.0 : 1.0;
            if (tau12 < 0.0) tau12 += beta;
            if (w->g4leg) legendre_p(2.0 / beta * tau12 - 1.0, nl, leg12);
            for (int c2 = 0; c2 < k2; ++c2) {
                double tau14 = ta[f1][a1] - tc[f2][c2];
                if (tau14 < 0.0) tau14 += beta;
                for (int a2 = 0; a2 < k2; ++a2) {
                    double M = M1[c1 * k1 + a1] * M2[c2 * k2 + a2];
                    if (f1 == f2) M -= M1[c2 * k1 + a1] * M2[c1 * k2 + a2];
                    double tau34 = ta[f2][a2] - tc[f2][c2];
                    const double sgnfull = sgn12 * (tau34 < 0.0 ? -1.0 : 1.0);
                    if (tau34 < 0.0) tau34 += beta;
                    M = -sgnfull * M;
                    if (w->g4leg) {
                        legendre_p(2.0 / beta * tau34 - 1.0, nl, leg34);
                        for (int l = 0; l < nl; ++l) for (int lp = 0; lp < nl; ++lp) for (int wm = -nwb; wm <= nwb; ++wm) {
                            const double ph = 2.0 * (double)wm * M_PI * tau14 / beta, v = leg34[lp] * M * leg12[l];
                            double *g = w->g4leg + (((((size_t)f1 * nf + f2) * nt + l) * nt + lp) * nw + (wm + nwb)) * 2;
                            g[0] += v * cos(ph); g[1] += v * sin(ph);
                        }
                    }
                    if (w->g4tau) {
                        const int b12 = (int)(nt * tau12 / beta), b14 = (int)(nt * tau14 / beta), b34 = (int)(nt * tau34 / beta);
                        w->g4tau[((((size_t)f1 * nf + f2) * nt + b12) * nt + b34) * nt + b14] += M;
                    }
                }
            }
        }
    }
    for (int f = 0; f < nf; ++f) { free(tc[f]); free(ta[f]); }
    free(leg12); free(leg34);
}

/* measure_g4iw_worm with the exact transform (CTQMC.F90:1618-1701, Nfft_worm.F90:94-137): every measurement adds
 * cfgsign/eta * exp(i[nu1 (t1-t2) + nu2 (t3-t4) + omega (t1-t4)]) (ph; WormPHConvention != 0: omega (t2-t3)) */
static void meas_g4iw_worm(ora_walker *w)
{
    const int nwf = w->p.n4iwf, nwb = w->p.n4iwb, nw = 2 * nwf, nb = 2 * nwb + 1;
    double t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) t[w->ops[i].wid] = w->ops[i].tau;
    const double val = w->cfgsign / w->eta, pib = M_PI / w->p.beta;
    for (int pp = 0; pp < 2; ++pp) {
        double *acc = pp ? w->g4wpp : w->g4w;
        if (!acc) continue;
        double x1, x2, x3;
        if (!pp) { x1 = t[1] - t[2]; x2 = t[3] - t[4]; x3 = w->p.wormphconv ? t[2] - t[3] : t[1] - t[4]; }
        else { x1 = t[1] - t[3]; x2 = t[2] - t[4]; x3 = t[3] - t[2]; }
        for (int i1 = -nwf; i1 < nwf; ++i1) for (int i2 = -nwf; i2 < nwf; ++i2) for (int i3 = -nwb; i3 <= nwb; ++i3) {
            double ph = pib * ((double)(2 * i1 + 1) * x1 + (double)(2 * i2 + 1) * x2 + (double)(2 * i3) * x3);
            size_t idx = ((((size_t)(i1 + nwf)) * nw + (i2 + nwf)) * nb + (i3 + nwb)) * 2;
            acc[idx] += val * cos(ph); acc[idx + 1] += val * sin(ph);
        }
    }
}

/* measure_giw_worm / measure_gtau_worm (CTQMC.F90:1061-1120): wormContainer = (c, c+); giw_worm(n) gets
 * cfgsign/eta * exp(i w_n (t1 - t2)), n = -Niw..Niw-1 (every second entry of the 4Niw-point transform of
 * x = (t1-t2)/(2 beta), Nfft_worm.F90:94-137); gtau_worm gets +-cfgsign/eta in the bin of t1 - t2 (+ beta, sign -1) */
static void meas_g_worm(ora_walker *w)
{
    double t[3] = {0, 0, 0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) t[w->ops[i].wid] = w->ops[i].tau;
    const double val = w->cfgsign / w->eta, beta = w->p.beta;
    if (w->giww) {
        const int niw = w->p.niw;
        for (int n = -niw; n < niw; ++n) {
            double ph = M_PI / beta * (double)(2 * n + 1) * (t[1] - t[2]);
            w->giww[2 * (n + niw)] += val * cos(ph); w->giww[2 * (n + niw) + 1] += val * sin(ph);
        }
    }
    if (w->gtauw) {
        double tau = t[1] - t[2], sgn = 1.0;
        if (tau < 0.0) { tau += beta; sgn = -1.0; }
        int itau = (int)((double)w->p.ntau * tau / beta) + 1;
        if (itau > w->p.ntau) itau = w->p.ntau;
        w->gtauw[itau - 1] += sgn * w->cfgsign / w->eta;
    }
}

/* autocorracc_add (Accumulators.F90:283-331): Welford update of level 0 with the buffer; whenever the lowest bits of the
 * count are ones, the mean of the two last lower-level values rolls over into the next level */
/* measure_p2iw_worm / p2tau / p2iwpp / p2taupp / p3iw / p3iwpp (CTQMC.F90:1297-1507) with the exact transform (Nfft_worm.F90:94-137):
 * the picked entries of the (4 N + 3)-point transform of x = dtau / (2 beta) are the bosonic frequencies 2 pi j / beta, j = -N..N,
 * those of the 4 N3iwf-point one the fermionic frequencies (2 i + 1) pi / beta, i = -N3iwf..N3iwf-1 */
static void meas_eq_worm(ora_walker *w)
{
    double t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) t[w->ops[i].wid] = w->ops[i].tau;
    const double val = w->cfgsign / w->eta, beta = w->p.beta, pib = M_PI / beta;
    const int sec = w->sector;
    if (sec == 6 || sec == 7) {
        const double dt = sec == 6 ? t[1] - t[3] : t[1] - t[2];
        double *giw = w->weq[sec - 6], *gt = w->weq[sec - 4];
        if (giw) {
            const int nb = w->p.n2iwb;
            for (int j = -nb; j <= nb; ++j) { const double ph = pib * (double)(2 * j) * dt; giw[2 * (j + nb)] += val * cos(ph); giw[2 * (j + nb) + 1] += val * sin(ph); }
        }
        if (gt) {                                   /* no sign change when the difference is wrapped (1340-1344) */
            double tau = dt; if (tau < 0.0) tau += beta;
            int itau = (int)((double)w->p.ntau * tau / beta) + 1;
            if (itau > w->p.ntau) itau = w->p.ntau;
            gt[itau - 1] += val;
        }
    } else if (sec == 8 || sec == 9) {
        double *g = w->weq[sec - 4];
        if (!g) return;
        const double x1 = sec == 8 ? t[1] - t[2] : t[1] - t[3], x2 = sec == 8 ? t[2] - t[3] : t[3] - t[2];
        const int nf3 = w->p.n3iwf, nb3 = w->p.n3iwb, nb = 2 * nb3 + 1;
        for (int i = -nf3; i < nf3; ++i)
            for (int j = -nb3; j <= nb3; ++j) {
                const double ph = pib * ((double)(2 * i + 1) * x1 + (double)(2 * j) * x2);
                double *o = g + 2 * ((size_t)(i + nf3) * nb + (j + nb3));
                o[0] += val * cos(ph); o[1] += val * sin(ph);
            }
    }
}

/* measure_ie_sigma_worm (CTQMC.F90:1123-1164): cfgsign/eta times the sign of the proposal's U-matrix factor times exp(i w_n (t1 - t4)),
 * n = -Niw..Niw-1 */
static void meas_gsigma_worm(ora_walker *w)
{
    double t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) t[w->ops[i].wid] = w->ops[i].tau;
    const double uf = gsigma_ufac(w);
    const double val = w->cfgsign / w->eta * (uf < 0.0 ? -1.0 : 1.0), dt = t[1] - t[4], pib = M_PI / w->p.beta;
    const int niw = w->p.niw;
    for (int n = -niw; n < niw; ++n) {
        const double ph = pib * (double)(2 * n + 1) * dt;
        w->gsw[2 * (n + niw)] += val * cos(ph); w->gsw[2 * (n + niw) + 1] += val * sin(ph);
    }
}

/* estimators of the worm sectors 5 and 10 - 16 (CTQMC.F90:1166-1294, 1708-2128) with the exact transform (Nfft_worm.F90:94-137): every
 * measurement adds cfgsign/eta times the sign of the proposal's U-matrix factor times a phase (or one count in a tau bin) */
static void acc_phase3(double *acc, int nwf, int nwb, double pib, double val, double x1, double x2, double x3)
{
    const int nw = 2 * nwf, nb = 2 * nwb + 1;
    for (int i1 = -nwf; i1 < nwf; ++i1) for (int i2 = -nwf; i2 < nwf; ++i2) for (int i3 = -nwb; i3 <= nwb; ++i3) {
        const double ph = pib * ((double)(2 * i1 + 1) * x1 + (double)(2 * i2 + 1) * x2 + (double)(2 * i3) * x3);
        const size_t idx = ((((size_t)(i1 + nwf)) * nw + (i2 + nwf)) * nb + (i3 + nwb)) * 2;
        acc[idx] += val * cos(ph); acc[idx + 1] += val * sin(ph);
    }
}
static void acc_taubin(const ora_walker *w, double *acc, double tau, double val, int antiperiodic)
{
    if (tau < 0.0) { tau += w->p.beta; if (antiperiodic) val = -val; }
    int itau = (int)((double)w->p.ntau * tau / w->p.beta) + 1;
    if (itau > w->p.ntau) itau = w->p.ntau;
    acc[itau - 1] += val;
}
static void meas_more_worm(ora_walker *w)
{
    double t[13] = {0};
    for (int i = 0; i < w->nops; ++i) if (w->ops[i].wid) t[w->ops[i].wid] = w->ops[i].tau;
    const double beta = w->p.beta, pib = M_PI / beta;
    const int sec = w->sector, niw = w->p.niw, nb2 = w->p.n2iwb;
    const double uf = worm_ufac(w, 1);
    const double val = w->cfgsign / w->eta * (uf < 0.0 ? -1.0 : 1.0);
    switch (sec) {
    case 5:                                             /* measure_ie_chi_worm, 1708-1772 */
        acc_phase3(w->wmore[0], w->p.n4iwf, w->p.n4iwb, pib, val, t[1] - t[4], t[5] - t[6], w->p.wormphconv ? t[4] - t[5] : t[1] - t[6]);
        break;
    case 10:                                            /* measure_qq_worm (minus sign, 1230), measure_qqtau_worm, 1215-1294 */
        for (int n = -niw; n < niw; ++n) {
            const double ph = pib * (double)(2 * n + 1) * (t[1] - t[4]);
            w->wmore[1][2 * (n + niw)] -= val * cos(ph); w->wmore[1][2 * (n + niw) + 1] -= val * sin(ph);
        }
        if (w->wmore[2]) acc_taubin(w, w->wmore[2], t[2] - t[6], val, 1);
        break;
    case 11:                                            /* measure_qqqq_worm, 1774-1840 */
        acc_phase3(w->wmore[3], w->p.n4iwf, w->p.n4iwb, pib, val, t[2] - t[6], t[8] - t[12], w->p.wormphconv ? t[6] - t[8] : t[2] - t[12]);
        break;
    case 12: case 13: {                                 /* measure_nqqdag_worm 1842-1902 (fermionic, bosonic), measure_qqdd_worm 1904-1964 (two fermionic) */
        const int nf3 = w->p.n3iwf, nb3 = w->p.n3iwb, n2 = sec == 12 ? 2 * nb3 + 1 : 2 * nf3;
        const double x1 = sec == 12 ? t[2] - t[4] : t[2] - t[6], x2 = sec == 12 ? t[4] - t[8] : t[4] - t[6];
        double *g = w->wmore[sec == 12 ? 4 : 5];
        for (int i = -nf3; i < nf3; ++i)
            for (int j = 0; j < n2; ++j) {
                const double m2 = sec == 12 ? (double)(2 * (j - nb3)) : (double)(2 * (j - nf3) + 1);
                const double ph = pib * ((double)(2 * i + 1) * x1 + m2 * x2);
                double *o = g + 2 * ((size_t)(i + nf3) * n2 + j);
                o[0] += val * cos(ph); o[1] += val * sin(ph);
            }
        break; }
    case 14: case 15: {                                 /* measure_ucaca_worm / ucacatau / uccaa / uccaatau, 1966-2128 */
        const double dt = sec == 14 ? t[1] - t[3] : t[3] - t[2];
        double *giw = w->wmore[sec == 14 ? 6 : 8], *gt = w->wmore[sec == 14 ? 7 : 9];
        if (giw) for (int j = -nb2; j <= nb2; ++j) { const double ph = pib * (double)(2 * j) * dt; giw[2 * (j + nb2)] += val * cos(ph); giw[2 * (j + nb2) + 1] += val * sin(ph); }
        if (gt) acc_taubin(w, gt, dt, val, 0);
        break; }
    case 16:                                            /* measure_quddag_worm, 1166-1211 */
        for (int n = -niw; n < niw; ++n) {
            const double ph = pib * (double)(2 * n + 1) * (t[1] - t[4]);
            w->wmore[10][2 * (n + niw)] += val * cos(ph); w->wmore[10][2 * (n + niw) + 1] += val * sin(ph);
        }
        break;
    default: break;
    }
}

static void autocorr_add(ora_walker *w)
{
    const size_t nc = (size_t)w->nflav * w->p.ntau;
    const int L = w->aclev;
    for (size_t i = 0; i < nc; ++i) {
        double *d = w->acdat + i * (size_t)L * 3;        /* [lvl][spill, mean, sqdiff] */
        int64_t n = w->acn; int lvl = 0;
        double val = w->acbuf[i];
        for (;;) {
            d[lvl * 3] += val;
            double incr1 = val - d[lvl * 3 + 1];
            d[lvl * 3 + 1] += incr1 / (double)(n + 1);
            double incr2 = val - d[lvl * 3 + 1];
            d[lvl * 3 + 2] += incr1 * incr2;
            if (!(n & 1) || lvl == L - 1) break;
            n /= 2; ++lvl;
            val = d[(lvl - 1) * 3] / 2; d[(lvl - 1) * 3] = 0.0;
        }
        w->acbuf[i] = 0.0;
    }
    w->acn += 1;
}

/* MeasSusz on the matrix solver: meas_ntau_n0 (Trace.F90:6472-6835).  For every flavour pair and every tau point t = 0 ..
 * Ntau-1 the reference inserts the density n_1 = c+_1 c_1 at the reference time tauref = upper edge of the window and n_2
 * at tauref + t beta/Ntau (points beyond beta: tauref - (Ntau - t) beta/Ntau) into the stored bra/ket vectors of the
 * outer states and adds  -(bra n_2 ket) cfgsign / Trace.  Restated with the operator list itself: the four operators
 * join the sequence and the local trace is taken from scratch (sum over the outer states as there). */
static void meas_ntau_n0(ora_walker *w)
{
    const int nf = w->nflav, ntau = w->p.ntau;
    const double beta = w->p.beta, tauD = beta / (double)ntau, tauref = w->wmax;
    const logval tr0 = full_trace(w, w->outer_sst);
    if (tr0.sign == 0.0) return;
    for (int f1 = 0; f1 < nf; ++f1) {
        for (int t = 0; t < ntau; ++t) {
            double tauint = tauref + (double)t * tauD;
            if (tauint > beta) tauint = tauref - (double)(ntau - t) * tauD;
            for (int f2 = 0; f2 < nf; ++f2) {
                /* annihilator first, then the creator at the same time: n = c+ c acting on the ket (6546-6555, 6613-6622);
                 * n_1 enters the list before n_2, so that it stays the earlier one if both times coincide */
                op_t o; memset(&o, 0, sizeof o);
                int idx[4];
                o.wid = 99;
                o.tau = tauref; o.flav = f1; o.ca = 2; list_insert(w, o); o.ca = 1; list_insert(w, o);
                o.tau = tauint; o.flav = f2; o.ca = 2; list_insert(w, o); o.ca = 1; list_insert(w, o);
                const logval tr = full_trace(w, w->outer_sst);
                int n = 0;
                for (int i = w->nops - 1; i >= 0 && n < 4; --i) if (w->ops[i].wid == 99) idx[n++] = i;
                for (int i = 0; i < n; ++i) list_remove(w, idx[i]);            /* descending indices */
                if (tr.sign != 0.0)
                    w->ntau_n0[((size_t)f1 * nf + f2) * (ntau + 1) + t] -= tr.sign * tr0.sign * exp(tr.log - tr0.log) * w->cfgsign;
            }
        }
    }
}

void ora_measure_once(ora_walker *w)
{
    int nf = w->nflav;
    if (w->sector != 1) {                                               /* CTQMC.F90:8586-8593 */
        if (w->sector == 2) { w->cnt_meas_w += 1; meas_g_worm(w); }
        else if (w->sector == 3) { if (w->gsw) { w->cnt_meas_w += 1; meas_gsigma_worm(w); } }              /* 8597-8599 */
        else if (w->sector == 6 || w->sector == 7) { w->cnt_meas_w += 1; meas_eq_worm(w); }              /* 8603-8610 */
        else if (w->sector == 8 || w->sector == 9) { if (w->weq[w->sector - 4]) { w->cnt_meas_w += 1; meas_eq_worm(w); } }   /* 8611-8616 */
        else if (w->sector == 5 || (w->sector >= 10 && w->sector <= 16)) {                            /* 8600-8602, 8617-8640 */
            static const int need[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 4, 5, -1, -1, -1};
            const int q = need[w->sector];
            if (w->sector == 5 ? w->wmore[0] != NULL : (q < 0 || w->wmore[q] != NULL)) {
                w->cnt_meas_w += 1;
                if (w->sector != 16 || w->wmore[10]) meas_more_worm(w);
            }
        }
        else if (w->g4w || w->g4wpp) { w->cnt_meas_w += 1; meas_g4iw_worm(w); }
        return;
    }
    w->cnt_meas += 1;
    w->meas_sign += w->cfgsign; w->cnt_sign += 1;                       /* MeasSign, CTQMC.F90:3343-3350 */
    {                                                                   /* MeasOuterHistograms, 3285-3339 (no state sampling) */
        const ora_locprob *lp = w->lp; const int o = w->outer_sst, n = lp->sst_n[o], off = lp->sst_off[o];
        full_trace_part(w, o, w->ptr);                                  /* parttrace of the current configuration (Trace.F90:6118) */
        w->outer_histo[o] += 1.0;
        w->sign_sst[o] += w->cfgsign;
        for (int i = 0; i < n; ++i) {
            const double c = w->ptr[i].sign != 0.0 ? exp(w->ptr[i].log) : 0.0;
            w->outer_state_histo[off + i] += 1.0;
            /* a vanishing partial trace carries sign(1.0, 0.0) = +1 in the reference (Trace.F90:6029-6030) */
            w->sign_state[off + i] += w->tr.sign * (w->ptr[i].sign < 0.0 ? -1.0 : 1.0) * w->cfgsign;
            w->tcontrib_sst[o] += c; w->tcontrib_state[off + i] += c;
        }
    }
    for (int f = 0; f < nf; ++f) {                                      /* MeasExpOrder, 3264-3282 */
        int k = w->k[f], last = w->p.maxhisto;
        if (w->p.offdiag) { k = 0; for (int i = 0; i < w->nops; ++i) if (w->ops[i].flav == f && w->ops[i].wid == 0) ++k; k /= 2; }      /* NOSOper / 2 */
        w->histo[(size_t)f * (last + 1) + (k < last ? k : last)] += 1.0;
    }
    meas_densitymatrix(w);
    if (w->isector != 1) return;                                        /* component sampling: no Z estimators (8554) */
    if (w->p.offdiag) {                                                 /* MeasGtauRem_full, CTQMC.F90:2475-2584 */
        const int kt = w->kt, nleg = w->p.nlegmax, ntau = w->p.ntau; const double beta = w->p.beta;
        int pc = 0;
        for (int f1 = 0; f1 < nf; ++f1) for (int i = 0; i < w->nops; ++i) {
            const op_t *c = &w->ops[i];
            if (c->wid || c->flav != f1 || c->ca != 1) continue;
            int pa = 0;
            for (int f2 = 0; f2 < nf; ++f2) for (int j = 0; j < w->nops; ++j) {
                const op_t *a = &w->ops[j];
                if (a->wid || a->flav != f2 || a->ca != 2) continue;
                double m = w->Mf[pc * kt + pa], x; int idt;
                if (a->tau - c->tau > 0.0) { idt = (int)((a->tau - c->tau) / beta * (double)ntau) + 1; x = 2.0 / beta * (a->tau - c->tau) - 1.0; }
                else { idt = (int)((beta + a->tau - c->tau) / beta * (double)ntau) + 1; x = 2.0 / beta * (beta + a->tau - c->tau) - 1.0; m = -m; }
                m = m * w->cfgsign;
                if (idt > ntau) idt = ntau;
                w->gtau_full[((size_t)f2 * nf + f1) * ntau + idt - 1] += m;
                double plx = 0.0, plm1 = 0.0, plp1 = 0.0;
                double *gl = w->gleg_full + ((size_t)f2 * nf + f1) * nleg;
                for (int il = 0; il < nleg; ++il) {
                    if (x == -1.0 || x == 1.0) plp1 = pow(x, il);
                    if (il == 0) plp1 = 1.0;
                    else if (il == 1) plp1 = x;
                    else { const int l = il - 1; plp1 = (2.0 * (double)l + 1.0) / (double)(l + 1) * x * plx - (double)l / (double)(l + 1) * plm1; }
                    gl[il] += m * plp1;
                    plm1 = plx; plx = plp1;
                }
                ++pa;
            }
            ++pc;
        }
        return;
    }
    for (int f = 0; f < nf; ++f) {                                      /* MeasGtauRem + measure_giw_nfft */
        int k = w->k[f];
        if (k == 0) continue;
        double *tc = (double *)malloc(sizeof(double) * k), *ta = (double *)malloc(sizeof(double) * k);
        int nc = 0, na = 0;
        for (int i = 0; i < w->nops; ++i) {
            if (w->ops[i].flav != f || w->ops[i].wid) continue;
            if (w->ops[i].ca == 1) tc[nc++] = w->ops[i].tau; else ta[na++] = w->ops[i].tau;
        }
        /* AccumGtau = 0: MeasGtauRem returns at once (CTQMC.F90:2267), neither G(l) nor G(tau); 2: per-measurement buffer */
        const int acc = w->p.accum_gtau;
        ora_meas_flavour(w->M[f], k, tc, ta, w->p.beta, w->cfgsign, w->p.nlegmax, acc ? w->gleg + (size_t)f * w->p.nlegmax : NULL,
                         w->p.ntau, acc == 0 ? NULL : (acc == 2 ? w->acbuf : w->gtau) + (size_t)f * w->p.ntau, 2 * w->p.niw,
                         w->p.meas_giw ? w->giw + (size_t)f * 4 * w->p.niw : NULL);
        /* per-measurement series (CTQMC.F90:2394-2402, inside MeasGtauRem: not with AccumGtau = 0): the element as it went into
         * G(tau), times the configuration sign once more; entry g_inmeas = number of this measurement */
        if (w->gms && acc && (long long)w->cnt_meas <= (long long)w->p.nmeas) {
            const size_t at = (size_t)f * w->p.nmeas + (size_t)w->cnt_meas - 1;
            for (int ia = 0; ia < k; ++ia) for (int ic = 0; ic < k; ++ic) {
                double m = w->M[f][ic * k + ia];
                if (!(ta[ia] - tc[ic] > 0.0)) m = -m;
                m = m * w->cfgsign;
                if (w->p.step_series & 1) w->gms[at] += m * w->cfgsign;
                const double r = fabs(ta[ia] - tc[ic]) / w->p.beta;
                if ((w->p.step_series & 2) && r >= 0.4 && r <= 0.6) w->gmid[at] += m * w->cfgsign;
                if (w->p.step_series & 4) w->sstep[(size_t)w->cnt_meas - 1] = w->cfgsign;
            }
        }
        free(tc); free(ta);
    }
    if (w->p.accum_gtau == 2) autocorr_add(w);                          /* gtau_accum%add(), CTQMC.F90:2439 */
    if (w->ntau_n0) meas_ntau_n0(w);                                    /* CTQMC.F90:8556-8560 */
    if (w->g4tau || w->g4leg) meas_g4leg(w);                            /* CTQMC.F90:8578 */
    if (w->g4iw || w->g4iwpp) meas_g4iw(w);                             /* CTQMC.F90:8579 */
}

/* Transform_DensityMatrix_EB_to_OCCB (CTQMC.F90:3163-3179): Evec rho Evec^T per block; keeps the diagonal in w->dm */
static void dm_occbasis(const ora_walker *w, int b, double *out)
{
    const ora_locprob *lp = w->lp; int n = lp->sst_n[b];
    const double *R = w->dmblk + w->dmoff[b], *V = lp->evec[b];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        double a = 0.0;
        for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) a += V[i + n * k] * R[k * n + l] * V[j + n * l];
        out[i * n + j] = a;
    }
}

static void finish_densitymatrix(ora_walker *w)
{
    const ora_locprob *lp = w->lp;
    double tmp[64 * 64];
    for (int b = 0; b < lp->nsst; ++b) {
        int n = lp->sst_n[b];
        dm_occbasis(w, b, tmp);
        for (int i = 0; i < n; ++i) w->dm[lp->sst_off[b] + i] = tmp[i * n + i];
    }
}

void ora_measure_finish(ora_walker *w)
{
    int nf = w->nflav; double beta = w->p.beta;
    if (w->cnt_meas == 0 && w->cnt_meas_w == 0) { w->aborted = 2; return; }
    w->mean_sign = w->meas_sign / (double)w->cnt_sign;                                  /* CTQMC.F90:8670 */
    double zs = -1.0 / (w->cnt_meas * w->mean_sign);                                     /* 8697 */
    for (int i = 0; i < w->dmsize; ++i) w->dmblk[i] *= zs;                              /* 8706-8708 */
    finish_densitymatrix(w);
    if (w->p.accum_gtau == 2)                                                           /* autocorracc_mean: level-0 mean */
        for (size_t i = 0; i < (size_t)nf * w->p.ntau; ++i) w->gtau[i] = w->acdat[i * (size_t)w->aclev * 3 + 1] * w->cnt_meas;
    for (size_t i = 0; i < (size_t)nf * w->p.ntau; ++i)                                 /* 8726-8731 */
        w->gtau[i] = -(w->gtau[i] / w->cnt_meas) / (-w->mean_sign) * (double)w->p.ntau / (beta * beta);
    for (size_t i = 0; i < (size_t)nf * (w->p.maxhisto + 1); ++i) w->histo[i] /= w->cnt_meas;   /* 8755 */
    if (w->gms) for (size_t i = 0; i < (size_t)nf * w->p.nmeas; ++i) { w->gms[i] /= beta * beta; w->gmid[i] /= beta * beta; }   /* 8741-8742 */
    if (w->ntau_n0)                                                                     /* 8744-8746 */
        for (int ff = 0; ff < nf * nf; ++ff) {
            double *r = w->ntau_n0 + (size_t)ff * (w->p.ntau + 1);
            for (int t = 0; t < w->p.ntau; ++t) r[t] *= zs;
            r[w->p.ntau] = r[0];
        }
    for (int s = 0; s < w->nstates; ++s) {                                              /* 8757-8764 (0/0 = NaN for blocks never visited) */
        w->sign_sst[s] /= w->outer_histo[s]; w->sign_state[s] /= w->outer_state_histo[s];
        w->outer_histo[s] /= w->cnt_meas; w->outer_state_histo[s] /= w->cnt_meas;
        w->tcontrib_sst[s] /= w->cnt_meas; w->tcontrib_state[s] /= w->cnt_meas;
    }
    for (int f = 0; f < nf; ++f)
        for (int l = 0; l < w->p.nlegmax; ++l) {                                        /* 8766-8771 */
            double *g = &w->gleg[(size_t)f * w->p.nlegmax + l];
            *g = *g * sqrt((double)(2 * l) + 1.0) / beta;
            *g = -*g * zs;
        }
    if (w->gleg_full) {                                                                 /* 8752-8753, 8768, 8771 */
        for (size_t i = 0; i < (size_t)nf * nf * w->p.ntau; ++i) w->gtau_full[i] = -w->gtau_full[i] * zs * (double)w->p.ntau / (beta * beta);
        for (size_t ff = 0; ff < (size_t)nf * nf; ++ff) for (int l = 0; l < w->p.nlegmax; ++l) {
            double *g = &w->gleg_full[ff * w->p.nlegmax + l];
            *g = *g * sqrt((double)(2 * l) + 1.0) / beta;
            *g = -*g * zs;
        }
    }
    if (w->p.meas_giw)
        for (size_t i = 0; i < (size_t)nf * 4 * w->p.niw; ++i) w->giw[i] = w->giw[i] * zs / beta;   /* 8815-8817 */
    {   /* 8790-8800 */
        size_t n2 = 2 * (size_t)(w->p.n4iwf + w->p.n4iwb), n4 = (size_t)nf * nf * 4 * w->p.n4iwf * w->p.n4iwf * (2 * w->p.n4iwb + 1) * 2;
        if (w->g4iw) for (size_t i = 0; i < n4; ++i) w->g4iw[i] *= zs / (beta * beta);
        if (w->g4iwpp) for (size_t i = 0; i < n4; ++i) w->g4iwpp[i] *= zs / (beta * beta);
        if (w->g2iw) for (size_t i = 0; i < (size_t)nf * n2 * n2 * 2; ++i) w->g2iw[i] *= zs / beta;
    }
    if (w->g4leg) {                                                                     /* 8775-8780 */
        const int nt = w->p.n4tau, nl = w->p.n4leg, nw = 2 * w->p.n4iwb + 1;
        for (int ff = 0; ff < nf * nf; ++ff) for (int l = 0; l < nl; ++l) for (int lp = 0; lp < nl; ++lp) for (int m = 0; m < nw; ++m) {
            double *g = w->g4leg + ((((size_t)ff * nt + l) * nt + lp) * nw + m) * 2;
            const double fac = zs * sqrt(2.0 * l + 1.0) * sqrt(2.0 * lp + 1.0) * (((lp + 1) & 1) ? -1.0 : 1.0) / beta;
            g[0] *= fac; g[1] *= fac;
        }
    }
    if (w->g4tau) {                                                                     /* 8783-8786 */
        const double nt = (double)w->p.n4tau, fac = zs * (nt / beta) * (nt / beta) * (nt / beta) / beta;
        for (size_t i = 0; i < (size_t)nf * nf * w->p.n4tau * w->p.n4tau * w->p.n4tau; ++i) w->g4tau[i] *= fac;
    }
    if (w->isector == 2 && (w->giww || w->gtauw)) {   /* wormScaling(SectorG), CTQMC.F90:8825, 8843-8876 */
        double scal = (w->cnt_meas_w != 0) ? w->cnt_sampling_w / (w->cnt_meas_w * w->cnt_sampling) * 4.0 / beta / w->mean_sign : 0.0;
        double nf2 = (double)nf * (double)nf;
        if (w->giww) for (int i = 0; i < 4 * w->p.niw; ++i) w->giww[i] = -w->giww[i] * scal / nf2;
        if (w->gtauw) for (int i = 0; i < w->p.ntau; ++i) w->gtauw[i] = w->gtauw[i] * scal * ((double)w->p.ntau / beta) / nf2;
    }
    if (w->isector == 3 && w->gsw) {                 /* wormNorm 32 / beta, result / (-4 (2 Nd)^2) (CTQMC.F90:8826, 8987-8998) */
        const double scal = (w->cnt_meas_w != 0) ? w->cnt_sampling_w / (w->cnt_meas_w * w->cnt_sampling) * 32.0 / beta / w->mean_sign : 0.0;
        for (int i = 0; i < 4 * w->p.niw; ++i) w->gsw[i] *= scal / (-4.0 * (double)nf * (double)nf);
    }
    if (w->isector >= 6 && w->isector <= 9) {        /* wormNorm 16 / beta (P2) or 16 / beta^2 (P3), CTQMC.F90:8830-8849, 9116-9130, 9215-9224, 9229-9313 */
        const double norm = w->isector <= 7 ? 16.0 / beta : 16.0 / (beta * beta);
        const double scal = (w->cnt_meas_w != 0) ? w->cnt_sampling_w / (w->cnt_meas_w * w->cnt_sampling) * norm / w->mean_sign : 0.0;
        for (int q = 0; q < 6; ++q) {
            if (!w->weq[q]) continue;
            const double fac = (q == 2 || q == 3) ? scal * ((double)w->p.ntau / beta) : scal;
            for (size_t i = 0; i < w->weq_n[q]; ++i) w->weq[q][i] *= fac;
        }
    }
    if (w->isector == 5 || (w->isector >= 10 && w->isector <= 16)) {      /* wormNorm 8829-8841; u_sparsity and the other factors 9003-9211 */
        static const double num[17] = {0, 0, 0, 0, 0, 128.0, 0, 0, 0, 0, 256.0, 65536.0, 4096.0, 8192.0, 256.0, 256.0, 256.0};
        const int sec = w->isector;
        const double norm = sec == 10 ? num[sec] / beta : num[sec] / (beta * beta);
        const double scal = (w->cnt_meas_w != 0) ? w->cnt_sampling_w / (w->cnt_meas_w * w->cnt_sampling) * norm / w->mean_sign : 0.0;
        /* u_sparsity: numbers of non-vanishing elements of slices of the antisymmetrised U matrices at the component's indices */
        int u[5] = {0, 0, 0, 0, 0}, rest = w->icomponent - 1;
        const int nred = sec == 10 ? 2 : 4;
        for (int i = nred; i >= 1; --i) { u[i] = rest % nf; rest /= nf; }
        double c1 = 0, c2 = 0, c3 = 0, c4 = 0;            /* counts */
        for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) {
            for (int c = 0; c < nf; ++c) {
                if (UA1(u[1], a, b, c) != 0.0) c1 += 1;                               /* u_matrix_antisymm_1(u_g(1),:,:,:) */
                if (UA2(a, b, u[2], c) != 0.0) c2 += 1;                               /* u_matrix_antisymm_2(:,:,u_g(2),:) */
                if (sec == 11) { if (UA1(u[3], a, b, c) != 0.0) c3 += 1; if (UA2(a, b, u[4], c) != 0.0) c4 += 1; }
            }
        }
        const double nfd = (double)nf, nf2 = nfd * nfd, nf4 = nf2 * nf2;
        double spars = 1.0;
        if (sec == 10) spars = c1 * c2 / (nf4 * nf2);
        else if (sec == 11) spars = c1 * c2 * c3 * c4 / (nf4 * nf4 * nf4);
        else if (sec == 14) {
            double a1 = 0, a2 = 0;
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) { if (UAB(u[1], a, u[2], b) != 0.0) a1 += 1; if (UAB(u[3], a, u[4], b) != 0.0) a2 += 1; }
            spars = a1 * a2 / nf4;
        } else if (sec == 15) {
            double a1 = 0, a2 = 0;
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) { if (UA2(a, b, u[2], u[4]) != 0.0) a1 += 1; if (UA1(u[1], u[3], a, b) != 0.0) a2 += 1; }
            spars = a1 * a2 / nf4;
        } else if (sec == 12) {                               /* 9278-9282 */
            double a3 = 0;
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) if (UAB(u[3], a, u[4], b) != 0.0) a3 += 1;
            spars = c1 * c2 * a3 / (nf4 * nf4);
        } else if (sec == 13) {                               /* 9339-9343 */
            double a2 = 0, a3 = 0;
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) {
                for (int c = 0; c < nf; ++c) if (UA2(a, u[3], b, c) != 0.0) a2 += 1;
                if (UAB(a, b, u[2], u[4]) != 0.0) a3 += 1;
            }
            spars = c1 * a2 * a3 / (nf4 * nf4);
        } else if (sec == 16) {
            double a2 = 0;
            for (int a = 0; a < nf; ++a) if (UAB(u[3], a, u[2], u[4]) != 0.0) a2 += 1;
            spars = c1 * a2 / nf4;
        }
        const double tb = (double)w->p.ntau / beta;
        const double fac[11] = {scal, scal * spars / nf2, scal * spars * tb / nf2, scal * spars, scal * spars, scal * spars, scal * spars, scal * tb * spars, scal * spars, scal * tb * spars, scal * spars};
        for (int q = 0; q < 11; ++q) if (w->wmore[q]) for (size_t i = 0; i < w->wmore_n[q]; ++i) w->wmore[q][i] *= fac[q];
    }
    if ((w->isector == 4 || w->isector == 1) && (w->g4w || w->g4wpp)) {   /* wormScaling(SectorG4), CTQMC.F90:8833-8849, 8883-8915 */
        double scal = (w->cnt_meas_w != 0) ? w->cnt_sampling_w / (w->cnt_meas_w * w->cnt_sampling) * 16.0 / (beta * beta) / w->mean_sign : 0.0;
        size_t n = (size_t)4 * w->p.n4iwf * w->p.n4iwf * (2 * w->p.n4iwb + 1) * 2;
        if (w->g4w) for (size_t i = 0; i < n; ++i) w->g4w[i] *= scal;
        if (w->g4wpp) for (size_t i = 0; i < n; ++i) w->g4wpp[i] *= scal;
    }
}

int ora_measure_sector(ora_walker *w, int isector, int icomponent)
{
    w->isector = isector; w->icomponent = icomponent; w->sector = 1;
    int rc = ora_measure(w);
    w->isector = 1;
    return rc;
}

void ora_set_wormeta(ora_walker *w, double eta) { w->eta = eta; }
/* sector and component of the following ora_sweep calls (step-by-step checks); every chain starts in Z space */
void ora_set_sector(ora_walker *w, int isector, int icomponent) { w->isector = isector; w->icomponent = icomponent; }

void ora_count_steps(ora_walker *w, int isector, int icomponent, int64_t nfix, int64_t *sw, int64_t *sz)
{
    int n = w->nops;
    op_t *saved = (op_t *)malloc(sizeof(op_t) * (size_t)(n + 1));
    memcpy(saved, w->ops, sizeof(op_t) * (size_t)n);
    int outer = w->outer_sst;
    w->isector = isector; w->icomponent = icomponent; w->sector = 1;
    for (int64_t j = 0; j < nfix; ++j) one_step(w);
    *sw = (int64_t)w->cnt_sampling_w; *sz = (int64_t)w->cnt_sampling;
    w->sector = 1; w->isector = 1;
    init_counters(w);
    /* clear_Trace + set_mc_config: back to the configuration the search started from (CTQMC.F90:6436-6440) */
    double *taus = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    int *orbs = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *spins = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *cas = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) { taus[i] = saved[i].tau; orbs[i] = (saved[i].flav >> 1) + 1; spins[i] = (saved[i].flav & 1) + 1; cas[i] = saved[i].ca; }
    ora_set_config(w, n, taus, orbs, spins, cas, outer);
    free(saved); free(taus); free(orbs); free(spins); free(cas);
}

int ora_measure(ora_walker *w)
{
    ora_measure_begin(w);
    for (int64_t m = 0; m < w->p.nmeas; ++m) {
        ora_sweep(w, w->p.ncorr);
        ora_measure_once(w);
    }
    ora_measure_finish(w);
    return w->aborted;
}

/* ------------------------------------------------------------------------------------------------------- */
/* accessors                                                                                                 */
/* ------------------------------------------------------------------------------------------------------- */

int ora_noper(const ora_walker *w) { return w->nops; }

void ora_get_config(const ora_walker *w, double *taus, int *orbs, int *spins, int *cas, int *outer_sst)
{
    for (int i = 0; i < w->nops; ++i) {
        taus[i] = w->ops[i].tau; orbs[i] = (w->ops[i].flav >> 1) + 1; spins[i] = (w->ops[i].flav & 1) + 1; cas[i] = w->ops[i].ca;
    }
    *outer_sst = w->outer_sst;
}

/* hashybs of get_mc_config: -1 = hybridising operator, 0 = worm operator (Trace.F90:6861-6865) */
void ora_get_hashybs(const ora_walker *w, int *hashybs)
{
    for (int i = 0; i < w->nops; ++i) hashybs[i] = w->ops[i].wid ? 0 : -1;
}

/* Trace.F90:6872-7000 set_mc_config_from_arrays */
int ora_set_config(ora_walker *w, int n, const double *taus, const int *orbs, const int *spins, const int *cas, int outer_sst)
{
    ensure_cap(w, n);
    w->nops = n;
    for (int i = 0; i < n; ++i) { w->ops[i].tau = taus[i]; w->ops[i].flav = 2 * (orbs[i] - 1) + (spins[i] - 1); w->ops[i].ca = cas[i]; w->ops[i].wid = 0; }
    w->outer_sst = outer_sst;
    w->wpos = 0; set_window(w);
    w->tr = full_trace(w, outer_sst);
    if (w->p.offdiag) {
        double *Mfn; int ktn;
        rebuild_full(w, &Mfn, &ktn, &w->det);
        free(w->Mf); w->Mf = Mfn; w->kt = ktn;
        w->cfgsign = -get_sign(w);
        return w->tr.sign == 0.0 ? 1 : 0;
    }
    double *Mn[MAXFLAV]; int kn[MAXFLAV];
    rebuild_minv(w, Mn, kn, &w->det);
    for (int f = 0; f < w->nflav; ++f) { free(w->M[f]); w->M[f] = Mn[f]; w->k[f] = kn[f]; }
    w->cfgsign = -get_sign(w);
    return w->tr.sign == 0.0 ? 1 : 0;
}

int ora_get_minv_full(const ora_walker *w, double *out)
{
    if (out) memcpy(out, w->Mf, sizeof(double) * (size_t)w->kt * w->kt);
    return w->kt;
}
int ora_get_full(const ora_walker *w, double *gleg_full, double *gtau_full)
{
    if (!w->gleg_full) return 0;
    memcpy(gleg_full, w->gleg_full, sizeof(double) * (size_t)w->nflav * w->nflav * w->p.nlegmax);
    memcpy(gtau_full, w->gtau_full, sizeof(double) * (size_t)w->nflav * w->nflav * w->p.ntau);
    return 1;
}
int ora_get_minv(const ora_walker *w, int orb, int spin, double *out)
{
    int f = 2 * (orb - 1) + (spin - 1), k = w->k[f];
    if (out) memcpy(out, w->M[f], sizeof(double) * (size_t)k * k);
    return k;
}

void ora_get_state(const ora_walker *w, double *dl, double *ds, double *tl, double *ts, double *cs, int *wpos, int *np)
{
    *dl = w->det.log; *ds = w->det.sign; *tl = w->tr.log; *ts = w->tr.sign; *cs = w->cfgsign; *wpos = w->wpos;
    *np = count_pairs(w, 0, NULL, NULL);
}

void ora_get_gleg(const ora_walker *w, double *o) { memcpy(o, w->gleg, sizeof(double) * (size_t)w->nflav * w->p.nlegmax); }
void ora_get_gtau(const ora_walker *w, double *o) { memcpy(o, w->gtau, sizeof(double) * (size_t)w->nflav * w->p.ntau); }
void ora_get_giw(const ora_walker *w, double *o) { memcpy(o, w->giw, sizeof(double) * (size_t)w->nflav * 4 * w->p.niw); }
void ora_get_histo(const ora_walker *w, double *o) { memcpy(o, w->histo, sizeof(double) * (size_t)w->nflav * (w->p.maxhisto + 1)); }
void ora_get_densitymatrix_diag(const ora_walker *w, double *o) { memcpy(o, w->dm, sizeof(double) * (size_t)w->nstates); }
void ora_get_outer_histo(const ora_walker *w, double *o) { memcpy(o, w->outer_histo, sizeof(double) * (size_t)w->nstates); }
void ora_get_outer_histograms(const ora_walker *w, double *sign_sst, double *tcontrib_sst, double *outer_state, double *sign_state, double *tcontrib_state)
{
    const size_t n = sizeof(double) * (size_t)w->nstates;
    memcpy(sign_sst, w->sign_sst, n); memcpy(tcontrib_sst, w->tcontrib_sst, n); memcpy(outer_state, w->outer_state_histo, n);
    memcpy(sign_state, w->sign_state, n); memcpy(tcontrib_state, w->tcontrib_state, n);
}
void ora_get_accpair(const ora_walker *w, int64_t *o) { memcpy(o, w->accpair, sizeof(int64_t) * (size_t)w->nflav * w->naccpair); }
void ora_get_energies(const ora_walker *w, double *o) { memcpy(o, w->E, sizeof(double) * (size_t)w->nstates); }
int ora_get_g2iw(const ora_walker *w, double *o)
{
    size_t n2 = 2 * (size_t)(w->p.n4iwf + w->p.n4iwb);
    if (!w->g2iw) return 0;
    memcpy(o, w->g2iw, sizeof(double) * (size_t)w->nflav * n2 * n2 * 2); return 1;
}
int ora_get_g4iw(const ora_walker *w, double *o, int pp)
{
    size_t n4 = (size_t)w->nflav * w->nflav * 4 * w->p.n4iwf * w->p.n4iwf * (2 * w->p.n4iwb + 1) * 2;
    const double *src = pp ? w->g4iwpp : w->g4iw;
    if (!src) return 0;
    memcpy(o, src, sizeof(double) * n4); return 1;
}
int ora_get_g4tau(const ora_walker *w, double *o)
{
    if (!w->g4tau) return 0;
    memcpy(o, w->g4tau, sizeof(double) * (size_t)w->nflav * w->nflav * w->p.n4tau * w->p.n4tau * w->p.n4tau); return 1;
}
int ora_get_g4leg(const ora_walker *w, double *o)
{
    if (!w->g4leg) return 0;
    memcpy(o, w->g4leg, sizeof(double) * (size_t)w->nflav * w->nflav * w->p.n4tau * w->p.n4tau * (2 * w->p.n4iwb + 1) * 2); return 1;
}
int ora_get_g4iw_worm(const ora_walker *w, double *o, int pp)
{
    const double *src = pp ? w->g4wpp : w->g4w;
    if (!src) return 0;
    memcpy(o, src, sizeof(double) * (size_t)4 * w->p.n4iwf * w->p.n4iwf * (2 * w->p.n4iwb + 1) * 2); return 1;
}
/* autocorracc_blocks (Accumulators.F90:343-361): sqdiff / (n - 1) per level, n halved from level to level;
 * out (Nd,2,Ntau,nlevels) C order; returns nlevels (0 if AccumGtau != 2); out may be NULL */
int ora_get_step_series(const ora_walker *w, double *mean, double *mid, double *sign)
{
    if (!w->gms) return 0;
    memcpy(mean, w->gms, sizeof(double) * (size_t)w->nflav * w->p.nmeas); memcpy(mid, w->gmid, sizeof(double) * (size_t)w->nflav * w->p.nmeas);
    memcpy(sign, w->sstep, sizeof(double) * (size_t)w->p.nmeas);
    return w->p.nmeas;
}
int ora_get_ntau_n0(const ora_walker *w, double *o)
{
    if (!w->ntau_n0) return 0;
    /* (Nd,2,Nd,2,Ntau+1) from flavour order f = 2 band + spin */
    memcpy(o, w->ntau_n0, sizeof(double) * (size_t)w->nflav * w->nflav * (w->p.ntau + 1));
    return 1;
}

int ora_get_gtau_blocks(const ora_walker *w, double *o)
{
    if (!w->acdat) return 0;
    const size_t nc = (size_t)w->nflav * w->p.ntau; const int L = w->aclev;
    if (o) for (size_t i = 0; i < nc; ++i) {
        int64_t n = w->acn;
        for (int l = 0; l < L; ++l) { o[i * L + l] = w->acdat[(i * L + l) * 3 + 2] / (n > 1 ? (double)(n - 1) : 0.0); n /= 2; }
    }
    return L;
}
int ora_get_giw_worm(const ora_walker *w, double *o)
{
    if (!w->giww) return 0;
    memcpy(o, w->giww, sizeof(double) * (size_t)4 * w->p.niw);
    return 1;
}
int ora_get_worm_eq(const ora_walker *w, int which, double *o)
{
    if (which < 0 || which > 5 || !w->weq[which]) return 0;
    memcpy(o, w->weq[which], sizeof(double) * w->weq_n[which]);
    return (int)w->weq_n[which];
}
int ora_get_worm_more(const ora_walker *w, int which, double *o)
{
    if (which < 0 || which > 10 || !w->wmore[which]) return 0;
    memcpy(o, w->wmore[which], sizeof(double) * w->wmore_n[which]);
    return (int)w->wmore_n[which];
}
int ora_get_gsigmaiw_worm(const ora_walker *w, double *o)
{
    if (!w->gsw) return 0;
    memcpy(o, w->gsw, sizeof(double) * (size_t)4 * w->p.niw);
    return 1;
}
int ora_get_gtau_worm(const ora_walker *w, double *o)
{
    if (!w->gtauw) return 0;
    memcpy(o, w->gtauw, sizeof(double) * (size_t)w->p.ntau);
    return 1;
}
void ora_get_worm_counters(const ora_walker *w, double *o)
{
    o[0] = w->try_wadd; o[1] = w->acc_wadd; o[2] = w->accqn_wadd; o[3] = w->try_wrem; o[4] = w->acc_wrem; o[5] = w->accqn_wrem;
    o[6] = w->try_wrep; o[7] = w->acc_wrep; o[8] = w->cnt_sampling_w; o[9] = w->cnt_meas_w;
    o[10] = w->wtry_add; o[11] = w->wacc_add; o[12] = w->wtry_rem; o[13] = w->wacc_rem; o[14] = w->sector;
}
int ora_nsst(const ora_walker *w) { return w->nsst; }
int ora_nsmax(const ora_walker *w) { return w->lp->nsmax; }
void ora_get_blocks(const ora_walker *w, int *n, int *off, int *st)
{
    memcpy(n, w->lp->sst_n, sizeof(int) * (size_t)w->nsst); memcpy(off, w->lp->sst_off, sizeof(int) * (size_t)w->nsst);
    memcpy(st, w->lp->states, sizeof(int) * (size_t)w->nstates);
}
void ora_get_densitymatrix(const ora_walker *w, double *o)
{
    const ora_locprob *lp = w->lp; size_t N = (size_t)w->nstates;
    double tmp[64 * 64];
    memset(o, 0, sizeof(double) * N * N);
    for (int b = 0; b < lp->nsst; ++b) {
        int n = lp->sst_n[b], off = lp->sst_off[b];
        dm_occbasis(w, b, tmp);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) o[(size_t)(off + i) * N + off + j] = tmp[i * n + j];
    }
}

/* <A> = sum_blocks tr(rho_b A_bb) for operator strings given as (flavour, ca) applied right to left, evaluated in the
 * eigenbasis like Meas*_from_Densitymatrix (CTQMC.F90:2842-3060) */
static double dm_expect(const ora_walker *w, int nop, const int *fl, const int *ca)
{
    const ora_locprob *lp = w->lp; double tot = 0.0;
    for (int b = 0; b < lp->nsst; ++b) {
        int n = lp->sst_n[b];
        for (int col = 0; col < n; ++col) {            /* A |col> */
            double v[64], out[64]; int s = b, ok = 1;
            for (int i = 0; i < n; ++i) v[i] = (i == col) ? 1.0 : 0.0;
            for (int q = 0; q < nop && ok; ++q) {
                int t = connect(w, s, fl[q], ca[q]);
                if (t < 0) { ok = 0; break; }
                int ns = lp->sst_n[s], nt = lp->sst_n[t]; const double *O = ora_lp_op(lp, s, fl[q], ca[q]);
                for (int i = 0; i < nt; ++i) { double a = 0.0; for (int j = 0; j < ns; ++j) a += O[i + nt * j] * v[j]; out[i] = a; }
                for (int i = 0; i < nt; ++i) v[i] = out[i];
                s = t;
            }
            if (!ok || s != b) continue;
            for (int row = 0; row < n; ++row) tot += w->dmblk[w->dmoff[b] + col * n + row] * v[row];   /* rho(col,row) A(row,col) */
        }
    }
    return tot;
}

/* MeasSsquare: S^2 = 1/2 (S+S- + S-S+) + Sz^2 written as the reference writes it (init_S2alt2, Operator.F90:548-633): a sum
 * of strings c+_{b3 s4} c_{b3 s3} c+_{b1 s2} c_{b1 s1} with the coefficients sm1(s2,s1) sm2(s4,s3) of the three terms
 * (Spm / PauliSM, AngularMomentum.F90:46-50), here applied with the operators of the eigenbasis: S[row * n + col] of block b. */
static void s2_block_eig(const ora_walker *w, int b, double *S)
{
    const ora_locprob *lp = w->lp; const int n = lp->sst_n[b], nd = w->p.nd;
    for (int i = 0; i < n * n; ++i) S[i] = 0.0;
    for (int xyz = 0; xyz < 3; ++xyz)
        for (int b1 = 0; b1 < nd; ++b1) for (int s1 = 0; s1 < 2; ++s1) for (int s2 = 0; s2 < 2; ++s2) {
            double c1 = xyz == 0 ? ((s2 == 1 && s1 == 0) ? 1.0 : 0.0) : (xyz == 1 ? ((s2 == 0 && s1 == 1) ? 1.0 : 0.0) : (s1 == s2 ? (s1 == 0 ? 1.0 : -1.0) : 0.0));
            if (c1 == 0.0) continue;
            for (int b3 = 0; b3 < nd; ++b3) for (int s3 = 0; s3 < 2; ++s3) for (int s4 = 0; s4 < 2; ++s4) {
                double c2 = xyz == 0 ? ((s4 == 0 && s3 == 1) ? 0.5 : 0.0) : (xyz == 1 ? ((s4 == 1 && s3 == 0) ? 0.5 : 0.0) : (s3 == s4 ? (s3 == 0 ? 0.25 : -0.25) : 0.0));
                if (c2 == 0.0) continue;
                const int fl[4] = {2 * b1 + s1, 2 * b1 + s2, 2 * b3 + s3, 2 * b3 + s4}, ca[4] = {2, 1, 2, 1};
                for (int col = 0; col < n; ++col) {
                    double v[64], out[64]; int s = b, ok = 1;
                    for (int i = 0; i < n; ++i) v[i] = (i == col) ? 1.0 : 0.0;
                    for (int q = 0; q < 4 && ok; ++q) {
                        int t = connect(w, s, fl[q], ca[q]);
                        if (t < 0) { ok = 0; break; }
                        int ns = lp->sst_n[s], nt = lp->sst_n[t]; const double *O = ora_lp_op(lp, s, fl[q], ca[q]);
                        for (int i = 0; i < nt; ++i) { double a = 0.0; for (int j = 0; j < ns; ++j) a += O[i + nt * j] * v[j]; out[i] = a; }
                        for (int i = 0; i < nt; ++i) v[i] = out[i];
                        s = t;
                    }
                    if (!ok || s != b) continue;           /* a string that leaves the block: undefined in the reference, left out */
                    for (int row = 0; row < n; ++row) S[row * n + col] += c1 * c2 * v[row];
                }
            }
        }
}

/* ssq[0 .. 2 Nd]: tr(rho S^2) over the blocks with N electrons (MeasureS2_from_densitymatrix, CTQMC.F90:2810-2840);
 * sop (4^Nd)^2 row major: S^2 in the occupation basis, block order (create_ssquare_full, CTQMC.F90:2769-2806) */
void ora_get_ssquare(const ora_walker *w, double *ssq, double *sop)
{
    const ora_locprob *lp = w->lp; size_t N = (size_t)w->nstates;
    double S[64 * 64];
    for (int i = 0; i <= 2 * w->p.nd; ++i) ssq[i] = 0.0;
    memset(sop, 0, sizeof(double) * N * N);
    for (int b = 0; b < lp->nsst; ++b) {
        int n = lp->sst_n[b], off = lp->sst_off[b]; const double *V = lp->evec[b];
        int nel = __builtin_popcount((unsigned)lp->states[off]);
        s2_block_eig(w, b, S);
        for (int col = 0; col < n; ++col) for (int row = 0; row < n; ++row) ssq[nel] += w->dmblk[w->dmoff[b] + col * n + row] * S[row * n + col];
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            double a = 0.0;
            for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) a += V[i + n * k] * S[k * n + l] * V[j + n * l];
            sop[(size_t)(off + i) * N + off + j] = a;
        }
    }
}

void ora_get_occ(const ora_walker *w, double *occ, double *rho1, double *rho2)
{
    int nf = w->nflav;
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) {
        {   /* occ(b1,s1,b2,s2) = <c+_1 c_1 c+_2 c_2> (2884-2942) */
            int fl[4] = { f2, f2, f1, f1 }, ca[4] = { 2, 1, 2, 1 };
            occ[f1 * nf + f2] = dm_expect(w, 4, fl, ca);
        }
        {   /* rho1(f2, f1) = <c+_2 c_1> (3020-3060) */
            int fl[2] = { f1, f2 }, ca[2] = { 2, 1 };
            rho1[f2 * nf + f1] = dm_expect(w, 2, fl, ca);
        }
    }
    if (rho2)
        for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) for (int f3 = 0; f3 < nf; ++f3) for (int f4 = 0; f4 < nf; ++f4) {
            /* rho2(f4,f3,f2,f1) = <c+_4 c+_3 c_2 c_1> (2945-3017) */
            int fl[4] = { f1, f2, f3, f4 }, ca[4] = { 2, 2, 1, 1 };
            rho2[((f4 * nf + f3) * nf + f2) * nf + f1] = dm_expect(w, 4, fl, ca);
        }
}
double ora_mean_sign(const ora_walker *w) { return w->mean_sign; }
void ora_get_counters_more(const ora_walker *w, double *o)
{
    o[0] = w->try_add4; o[1] = w->acc_add4; o[2] = w->try_rem4; o[3] = w->acc_rem4;
    o[4] = w->try_add_outer; o[5] = w->acc_add_outer; o[6] = w->try_rem_outer; o[7] = w->acc_rem_outer;
    o[8] = w->try_flavc; o[9] = w->acc_flavc;
}
void ora_get_counters(const ora_walker *w, double *o)
{
    o[0] = w->try_add; o[1] = w->acc_add; o[2] = w->accqn_add; o[3] = w->try_rem; o[4] = w->acc_rem; o[5] = w->accqn_rem;
    o[6] = w->try_glob; o[7] = w->acc_glob; o[8] = w->accqn_glob; o[9] = w->try_shift; o[10] = w->acc_shift;
    o[11] = w->cnt_sampling; o[12] = w->cnt_meas;
}
