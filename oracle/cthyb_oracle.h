/* TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference CT-HYB hot path.
 *
 * Single Markov chain, diagonal hybridisation.  Local problem: blocks of H_loc from the QuantumNumbers string
 * (locprob.h); one-state blocks use the reference's "scalar" trace path (Trace.F90:5556-5727), larger blocks the
 * matrix path (Trace.F90:5734-6033).  Plain arrays instead of the
 * reference's linked lists; the move logic, proposal volumes, acceptance ratios, random-number consumption
 * order and estimator arithmetic follow the reference function by function (citations in cthyb_oracle.c).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef CTHYB_ORACLE_H
#define CTHYB_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Random source: the oracle never generates numbers itself, so that the same chain can be driven by
 * std::mt19937 (to pin it against the reference) or by the product's counter-based generator (to check the GPU). */
typedef struct ora_rng {
    void *state;
    double (*uniform)(void *state);               /* [0,1)      : Random.F90:49 grnd()     */
    int (*randint)(void *state, int lo, int hi);  /* [lo,hi]    : Random.F90:54 randint()  */
} ora_rng;

typedef struct ora_params {
    int nd;                 /* Nd */
    double beta;
    int nftau;              /* Nftau: points of F(tau) incl. both ends */
    int ntau;               /* Ntau: G(tau) bins */
    int nlegmax;            /* NLegMax */
    int niw;                /* Niw (giw has 2*Niw frequencies) */
    int meas_giw;           /* MeasGiw */
    int ncorr;              /* NCorr */
    int64_t nmeas;          /* Nmeas */
    int maxhisto;           /* MaxHisto */
    double p_glob;          /* PercentageGlobalMove */
    double p_shift;         /* PercentageTauShiftMove */
    double taudiffmax;      /* TaudiffMax from the parameter string (window at init_solver, NSlide) */
    int has_szt;            /* "Szt" in QuantumNumbers (CTQMC.F90 glob_has_Szt) */
    char qns[64];           /* QuantumNumbers string (States.F90:164) */
    double epsevec;         /* EPSEVEC (Operator.F90:784) */
    int fourpnt;            /* module variable fourpnt (impurity.py:515); bit 4 = Matsubara G4 in Z space (CTQMC.F90:226) */
    int n4iwf, n4iwb;       /* N4iwf, N4iwb */
    int meas_g2iw, meas_g4iwpp, zphconv;   /* MeasG2iw, MeasG4iwPP, ZPHConvention */
    int n4tau, n4leg;       /* N4tau, N4leg: fourpnt bit 1 = imaginary-time G4 (g4tau), bit 2 = Legendre G4 (g4leg), CTQMC.F90:226, 544-551 */
    /* worm sampling, sector 4 = two-particle Green's function (CTQMC.F90:335-378, 434) */
    double p_worm_ins, p_worm_rep;         /* PercentageWormInsert, PercentageWormReplace */
    double worm_eta;                       /* WormEta */
    int worm_meas_g4iw, g4ph, g4pp, wormphconv;   /* WormMeasG4iw, G4ph, G4pp, WormPHConvention */
    /* worm sampling, sector 2 = one-particle Green's function (CTQMC.F90:597-612) */
    int worm_meas_giw, worm_meas_gtau;            /* WormMeasGiw, WormMeasGtau */
    double epsblock;        /* EPSBLOCK: threshold of the automatic block partitioning, QuantumNumbers = All (CTQMC.F90:7972) */
    int accum_gtau;         /* AccumGtau: 0 = G(tau)/G(l) not measured, 1 = running mean, 2 = autocorrelation accumulator (CTQMC.F90:492-500) */
    int meas_susz;          /* MeasSusz (matrix solver): <n(tau) n(0)>, meas_ntau_n0 (CTQMC.F90:534-540, Trace.F90:6472-6835) */
    int step_series;        /* bit 0: Gtau_mean_step, bit 1: Gtau_mid_step, bit 2: sign_step - per-measurement series (CTQMC.F90:43-45, 2394-2402) */
    /* worm sampling with equal-time pairs, sectors 6 / 7 (P2 ph / pp) and 8 / 9 (P3 ph / pp) (CTQMC.F90:654-726) */
    int n2iwb, n3iwf, n3iwb;                       /* N2iwb, N3iwf, N3iwb */
    int worm_meas_eq[6];                           /* WormMeasP2iwPH, WormMeasP2iwPP, WormMeasP2tauPH, WormMeasP2tauPP, WormMeasP3iwPH, WormMeasP3iwPP */
    /* user symmetry moves of the global move (NSymMove, SymMoveNN; Trace.F90:403-404, 471-478): nsym rows of 2Nd entries, values
     * 1-based in the order band + Nd * spin; the inverses are made by the oracle */
    int nsym; int symmoves[8 * 16];
    int worm_meas_gsigma;                          /* WormMeasGSigmaiw: sector 3, improved estimator of the one-particle Green's function (CTQMC.F90:619-623) */
    int offdiag;                                   /* offdiag = 1: hybridisation function with both flavour indices, one inverse matrix over all flavours (CTQMC.F90:461-467) */
    int flavc;                                     /* flavourchange_moves (b_exch, CTQMC.F90:481-485): StepFlavourchange_general in place of the two-pair moves */
    double p_4op, p_outer;                         /* Percentage4OperatorMove, PercentageOuterMove (CTQMC.F90:332-333, 8441-8442) */
    /* worm sectors 5, 10 - 16 (CTQMC.F90:379-430, 624-652): WormMeasH4iw, WormMeasQQ, WormMeasQQtau, WormMeasQQQQ, WormMeasNQQdag, WormMeasQQdd,
     * WormMeasUcaca, WormMeasUcacatau, WormMeasUccaa, WormMeasUccaatau, WormMeasQUDdag */
    int worm_meas_more[11];
} ora_params;

typedef struct ora_walker ora_walker;

/* init_solver (CTQMC.F90:7894-8085).
 * umatrix (2Nd)^4, ftau_full (Nd,2,Nd,2,Nftau), muimp_full (Nd,2,Nd,2) [already negated], C order.
 * Consumes one uniform() for the initial outer superstate (Trace.F90:657-663). Returns NULL on failure. */
ora_walker *ora_create(const ora_params *p, const double *umatrix, const double *ftau_full,
                       const double *muimp_full, ora_rng *rng);
void ora_destroy(ora_walker *w);

void ora_set_taudiffmax(ora_walker *w, double x);        /* CTQMC.F90:8180 */
void ora_warmup(ora_walker *w, int64_t n);               /* CTQMC.F90:8188 */
int ora_measure(ora_walker *w);                          /* CTQMC.F90:8355 sector Z; returns `aborted` */
/* ctqmc_measure(isector, icomponent) with isector = 2 (G worm), 4 (G4 worm), 6 - 9 (P2 / P2pp / P3 / P3pp worms) or 1 (CTQMC.F90:8355-8849); returns `aborted` */
int ora_measure_sector(ora_walker *w, int isector, int icomponent);
void ora_set_wormeta(ora_walker *w, double eta);
void ora_set_sector(ora_walker *w, int isector, int icomponent);   /* sector of the following ora_sweep calls */         /* module variable wormeta(isector) (impurity.py:719) */
/* count_steps (CTQMC.F90:6299-6444): nfix steps without measuring, configuration restored afterwards */
void ora_count_steps(ora_walker *w, int isector, int icomponent, int64_t nfix, int64_t *steps_worm, int64_t *steps_z);
/* worm results: g4iw_worm / g4iwpp_worm (2N4iwf,2N4iwf,2N4iwb+1) complex interleaved; returns 0 if not measured */
int ora_get_g4iw_worm(const ora_walker *w, double *out_reim, int pp);
/* giw_worm (2Niw) complex interleaved / gtau_worm (Ntau) of the sampled component (CTQMC.F90:1061-1120, 8858-8876);
 * return 0 if not measured */
int ora_get_ntau_n0(const ora_walker *w, double *out);       /* ntau_n0 (Nd,2,Nd,2,Ntau+1), MeasSusz = 1; returns 0 if not measured */
int ora_get_gtau_blocks(const ora_walker *w, double *out);   /* gtau_blocks (Nd,2,Ntau,nlevels), AccumGtau = 2; returns nlevels */
int ora_get_giw_worm(const ora_walker *w, double *out_reim);
/* which = 0 p2iw_worm, 1 p2iwpp_worm (2N2iwb+1 complex interleaved), 2 p2tau_worm, 3 p2taupp_worm (Ntau),
 * 4 p3iw_worm, 5 p3iwpp_worm (2N3iwf, 2N3iwb+1 complex interleaved); returns the number of doubles copied, 0 if not measured */
int ora_get_worm_eq(const ora_walker *w, int which, double *out);
int ora_get_gtau_worm(const ora_walker *w, double *out);
/* which = 0 h4iw_worm (2N4iwf,2N4iwf,2N4iwb+1), 1 qq_worm (2Niw), 2 qqtau_worm (Ntau, real), 3 qqqq_worm (as h4iw), 4 nqqdag_worm (2N3iwf,2N3iwb+1),
 * 5 qqdd_worm (2N3iwf,2N3iwf), 6 ucaca_worm (2N2iwb+1), 7 ucacatau_worm (Ntau, real), 8 uccaa_worm, 9 uccaatau_worm, 10 quddag_worm (2Niw);
 * complex arrays interleaved; returns the number of doubles copied, 0 if not measured */
int ora_get_worm_more(const ora_walker *w, int which, double *out);
int ora_get_gsigmaiw_worm(const ora_walker *w, double *out_reim);   /* GSigmaiw_worm (2Niw) complex interleaved, sector 3 */
/* [0]TryWormAdd [1]AccWormAdd [2]AccQNWormAdd [3]TryWormRem [4]AccWormRem [5]AccQNWormRem [6]TryWormRep [7]AccWormRep
 * [8]CntSampling(worm) [9]CntMeas(worm) [10]TryAdd(worm) [11]AccAdd(worm) [12]TryRem(worm) [13]AccRem(worm) [14]sector now */
void ora_get_worm_counters(const ora_walker *w, double *out15);
/* the two halves of ora_measure, for step-by-step checks */
void ora_init_counters(ora_walker *w);                   /* CTQMC.F90:847 */
/* ctqmc_calibrate(list, phase1pairnum, phase2taugrid, phase2pairnum, nprogress), CTQMC.F90:8088; returns the steps done */
int64_t ora_calibrate(ora_walker *w, int list, int phase1pairnum, int phase2taugrid, int phase2pairnum, int nprogress);
int ora_get_accpairtau(const ora_walker *w, double *o);  /* AccPairTau (unsorted), returns its length */
int ora_naccpair(const ora_walker *w);                   /* current number of AccPair bins */
void ora_measure_begin(ora_walker *w);                   /* init_counters etc. */
void ora_sweep(ora_walker *w, int64_t nsteps);           /* nsteps Monte Carlo steps of the Z-space loop */
void ora_measure_once(ora_walker *w);                    /* one pass of all Z-space estimators */
void ora_measure_finish(ora_walker *w);                  /* normalisation block, CTQMC.F90:8663-8817 */

/* configuration access (CTQMC.F90:9358-9400; Trace.F90:6837-7000) */
int ora_noper(const ora_walker *w);
void ora_get_config(const ora_walker *w, double *taus, int *orbs, int *spins, int *cas, int *outer_sst);
void ora_get_hashybs(const ora_walker *w, int *hashybs);   /* -1 hybridising, 0 worm operator */
int ora_set_config(ora_walker *w, int n, const double *taus, const int *orbs, const int *spins, const int *cas,
                   int outer_sst);
/* M block of flavour (orb,spin) [1-based like the reference], time ordered, row=creator, col=annihilator;
 * returns k and copies k*k doubles (row major) if out != NULL */
int ora_get_minv(const ora_walker *w, int orb, int spin, double *out);
/* offdiag = 1: the one inverse matrix over all flavours (rows = creators, columns = annihilators, by flavour then time); returns its order */
int ora_get_minv_full(const ora_walker *w, double *out);
/* gleg_full (Nd,2,Nd,2,NLegMax), gtau_full (Nd,2,Nd,2,Ntau): first index pair = annihilator flavour (CTQMC.F90:2531, 2549); 0 if not measured */
int ora_get_full(const ora_walker *w, double *gleg_full, double *gtau_full);
void ora_get_state(const ora_walker *w, double *det_log, double *det_sign, double *tr_log, double *tr_sign,
                   double *cfgsign, int *window_position, int *npairs);

/* results (valid after ora_measure / ora_measure_finish); shapes as the reference's module arrays, C order */
void ora_get_gleg(const ora_walker *w, double *out);      /* (Nd,2,NLegMax) */
void ora_get_gtau(const ora_walker *w, double *out);      /* (Nd,2,Ntau) */
void ora_get_giw(const ora_walker *w, double *out_reim);  /* (Nd,2,2Niw) complex interleaved */
void ora_get_histo(const ora_walker *w, double *out);     /* (Nd,2,MaxHisto+1) */
/* two-particle Z-space estimators (CTQMC.F90:2130-2234), complex interleaved, C order of the reference's axes:
 * g2iw (Nd,2,n2,n2) with n2 = 2(N4iwf+N4iwb), axes (annihilator frequency, creator frequency);
 * g4iw / g4iw_pp (Nd,2,Nd,2,2N4iwf,2N4iwf,2N4iwb+1).  Return 0 if the quantity was not measured. */
int ora_get_g2iw(const ora_walker *w, double *out_reim);
int ora_get_g4iw(const ora_walker *w, double *out_reim, int pp);
/* g4tau (Nd,2,Nd,2,N4tau,N4tau,N4tau) and g4leg (Nd,2,Nd,2,N4tau,N4tau,2N4iwb+1) complex interleaved (MeasGtau4PntLeg) */
int ora_get_g4tau(const ora_walker *w, double *out);
int ora_get_g4leg(const ora_walker *w, double *out_reim);
void ora_get_densitymatrix_diag(const ora_walker *w, double *out);  /* 4^Nd, occupation basis, block order */
void ora_get_densitymatrix(const ora_walker *w, double *out);       /* (4^Nd)^2 row major, occupation basis, block order
                                                                       (after Transform_DensityMatrix_EB_to_OCCB, CTQMC.F90:3163) */
void ora_get_outer_histo(const ora_walker *w, double *out);         /* first nsst entries: OuterSuperstateHisto */
/* the other arrays of MeasOuterHistograms (CTQMC.F90:3285-3339, normalised 8757-8764), 4^Nd doubles each: SignHistoSuperstates,
 * TraceContribSuperstates (first nsst entries), OuterStateHisto, SignHistoStates, TraceContribStates (eigenstates, block order) */
void ora_get_outer_histograms(const ora_walker *w, double *sign_sst, double *tcontrib_sst, double *outer_state, double *sign_state, double *tcontrib_state);
int ora_nsst(const ora_walker *w);
int ora_nsmax(const ora_walker *w);
void ora_get_blocks(const ora_walker *w, int *sst_n, int *sst_off, int *states);  /* [nsst],[nsst],[4^Nd] */
int ora_get_step_series(const ora_walker *w, double *mean, double *mid, double *sign);   /* (2Nd, Nmeas) x2, (Nmeas); returns Nmeas or 0 */
void ora_get_ssquare(const ora_walker *w, double *ssq, double *sop);   /* (2Nd+1), (4^Nd)^2: MeasSsquare, CTQMC.F90:2769-2840 */
void ora_get_occ(const ora_walker *w, double *occ, double *rho1, double *rho2);   /* (Nd,2,Nd,2), (2Nd)^2, (2Nd)^4: CTQMC.F90:2884-3060 */
void ora_get_accpair(const ora_walker *w, int64_t *out);  /* (Nd,2,Ntau) */
void ora_get_energies(const ora_walker *w, double *out);  /* 4^Nd eigenstate energies */
double ora_mean_sign(const ora_walker *w);
/* counters: [0]TryAdd [1]AccAdd [2]AccQNAdd [3]TryRem [4]AccRem [5]AccQNRem [6]TryGlob [7]AccGlob [8]AccQNGlob
 * [9]TryShift [10]AccShift [11]CntSampling [12]CntMeas */
void ora_get_counters(const ora_walker *w, double *out13);
/* [0]TryAdd4 [1]AccAdd4 [2]TryRem4 [3]AccRem4 [4]TryAddOuter [5]AccAddOuter [6]TryRemOuter [7]AccRemOuter [8]TryFlavc [9]AccFlavc */
void ora_get_counters_more(const ora_walker *w, double *out10);

/* ---- pure pieces, exposed for kernel-level parity tests ---- */
double ora_hybr(const double *ftab, int nftau, double beta, double dtau);          /* Trace.F90:1014-1033 */
double ora_detrat_add(const double *M, int k, const double *Q, const double *R, double S);  /* MatrixUpdate.F90:392 */
void ora_mat_add(const double *M, int k, const double *Q, const double *R, double S, int pos_c, int pos_a,
                 double *out);                                                      /* MatrixUpdate.F90:410 */
void ora_mat_rem(const double *M, int k, int pos_c, int pos_a, double *out);        /* MatrixUpdate.F90:455 */
/* LU inverse + log-determinant, own restatement of what get_MatLogDetFull (MatrixUpdate.F90:41-99) obtains
 * from LAPACK dgetrf/dgetri; A (k x k, row major) is overwritten by its inverse. returns 0 on success */
int ora_inverse_logdet(double *A, int k, double *logdet, double *sign);
/* one configuration's contribution to the Legendre / tau-bin / Matsubara estimators of one flavour
 * (CTQMC.F90:2285-2424, Nfft_z.F90:376-414); accumulates into the outputs */
void ora_meas_flavour(const double *M, int k, const double *tau_c, const double *tau_a, double beta, double cfgsign,
                      int nleg, double *gleg, int ntau, double *gtau, int niw2, double *giw_reim);

#ifdef __cplusplus
}
#endif
#endif
