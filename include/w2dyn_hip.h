/* C ABI of libw2dyn_hip.so - the MI355X (gfx950) CT-HYB hot path.
 *
 * Every entry point replaces one call (or one group of module-variable reads) of the reference's f2py module
 * `w2dyn.auxiliaries.CTQMC.mctqmc`, which is the only interface the reference's Python driver uses for this path
 * (w2dyn/dmft/impurity.py:18, 387-388, 514-515, 578-598, 650-651).  Citations are to /root/reference.
 *
 * Conventions: plain pointers and sizes, C (row-major) order for every array with the axis order of the reference's
 * Fortran arrays as numpy sees them, e.g. gleg(Nd,2,NLegMax).  All functions return 0 on success, a negative
 * W2DYN_E* code otherwise; w2dyn_last_error() gives the message.  One solver context owns one GPU; it is not
 * re-entrant (the reference module is a process-wide singleton, impurity.py:370-372).
 *
 * The reference runs ONE Markov chain per MPI rank.  Here one context runs `nwalkers` independent chains on its
 * GPU; chain w draws from the stream (seed, first_walker + w) (reference: seed + rank, DMFT.py:259).  Nwarmups,
 * Nmeas and NCorr keep their per-chain meaning.  Result arrays are means over all chains of the context.
 */
#ifndef W2DYN_HIP_H
#define W2DYN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2dyn_solver w2dyn_solver;

enum {
    W2DYN_OK = 0,
    W2DYN_EINVAL = -1,       /* bad argument / call order                                    */
    W2DYN_EUNSUPPORTED = -2, /* a mode of the reference this build refuses loudly             */
    W2DYN_EHIP = -3,         /* HIP runtime error                                            */
    W2DYN_EOVERFLOW = -4,    /* a walker exceeded the configured expansion-order capacity    */
    W2DYN_ENUMERIC = -5,     /* reference would `stop` (e.g. "DetRat too small in removal")  */
    W2DYN_EINTERRUPTED = -6  /* stepping call after SIGINT / SIGTERM / SIGUSR1: nothing launched (the ctqmc_* loops end softly instead: `aborted`) */
};

/* lifetime -------------------------------------------------------------------------------------------------- */
int w2dyn_create(w2dyn_solver **out, int device);            /* replaces importing the f2py module           */
int w2dyn_destroy(w2dyn_solver *s);
const char *w2dyn_last_error(const w2dyn_solver *s);

/* chains per GPU, index of this context's first chain (for multi-GPU sharding), expansion-order capacity per
 * flavour (0 = choose from beta), chains per accumulator bin.  Must precede init_solver. No reference counterpart:
 * the reference has exactly one chain per process. */
int w2dyn_set_layout(w2dyn_solver *s, int nwalkers, int64_t first_walker, int kmax, int walkers_per_bin);

/* init_rnd(nseed)                                  CTQMC.F90:9404 */
int w2dyn_init_rnd(w2dyn_solver *s, int64_t nseed);
/* init_paras(parastr): "key = value #" lines      CTQMC.F90:9412, Parameters.F90:281-320 (prefix lookup 77-105) */
int w2dyn_init_paras(w2dyn_solver *s, const char *parastr);
/* dest_paras()                                     CTQMC.F90:9420 */
int w2dyn_dest_paras(w2dyn_solver *s);
/* module variable `fourpnt` (impurity.py:515)                       */
int w2dyn_set_fourpnt(w2dyn_solver *s, int fourpnt);
/* init_solver(u_matrix_in, ftau_full, muimp_full, screening_function, iter_no, nftau, nbands)
 *                                                  CTQMC.F90:7894-8085
 * u_matrix (2Nd)^4, ftau_full (Nd,2,Nd,2,Nftau), muimp_full (Nd,2,Nd,2) already negated (impurity.py:529),
 * screening_function (Nd,2,Nd,2,Nftau) (must be zero: Phonon/Screening/Uw are refused). Host pointers.
 * offdiag = 0 (parameter string) uses the flavour-diagonal elements of ftau_full only (CTQMC.F90:7950-7955); offdiag = 1 all of
 * it: one hybridisation matrix over all flavours, pair moves with independent flavours (Trace.F90:2690-2770, 2835-2997), results
 * gleg_full / gtau_full (MeasGtauRem_full, CTQMC.F90:2475-2584); Z-space runs only.
 * Percentage4OperatorMove / PercentageOuterMove > 0: the two-pair moves StepAdd4 / StepRem4 (CTQMC.F90:3615-3889, 4153-4430) and the
 * pair moves that change the outer block (CTQMC.F90:3422-3449, 3935-3962), results accadd4 / accrem4; Z-space runs only. */
int w2dyn_init_solver(w2dyn_solver *s, const double *u_matrix, const double *ftau_full, const double *muimp_full,
                      const double *screening_function, int iter_no, int nftau, int nbands);
/* set_taudiffmax(x)                                CTQMC.F90:8180 */
int w2dyn_set_taudiffmax(w2dyn_solver *s, double taudiffmax);
/* ctqmc_calibrate(list, phase1pairnum, phase2taugrid, phase2pairnum, nprogress)      CTQMC.F90:8088-8178
 * Window calibration (impurity.py:1150-1172).  list != 0: phase 1, every chain steps until phase1pairnum accepted pair
 * distances are on its list (tested every min(1000, nprogress) steps); result "accpairtau" (phase1pairnum) is the mean
 * over the chains of the sorted lists, "accpairtau_chains" (nwalkers, phase1pairnum) the lists themselves.
 * list == 0: phase 2, histogram "accpair" (Nd,2,phase2taugrid) of accepted pair distances over [0, taudiff_max], summed
 * over chains; every chain steps until each of its flavours has phase2pairnum accepted pairs.  No window shifts. */
int w2dyn_ctqmc_calibrate(w2dyn_solver *s, int list, int phase1pairnum, int phase2taugrid, int phase2pairnum, int nprogress);
/* init_counters()                                  CTQMC.F90:847 */
int w2dyn_init_counters(w2dyn_solver *s);
/* ctqmc_warmup(n)                                  CTQMC.F90:8188 */
int w2dyn_ctqmc_warmup(w2dyn_solver *s, int64_t nwarmups);
/* ctqmc_measure(isector, icomponent)               CTQMC.F90:8355
 * isector 1 = Z space; isector 2 = worm sampling of the one-particle Green's function component icomponent
 * (giw_worm / gtau_worm, CTQMC.F90:1061-1120); isector 4 = worm sampling of the two-particle Green's function
 * component icomponent (0 = all components); isector 3 = GSigma worm (improved estimator: three-operator equal-time object with random
 * flavours and the U-matrix factor in the proposal, Trace.F90:3835-3919, CTQMC.F90:5687-5696; GSigmaiw_worm, CTQMC.F90:1123-1164;
 * icomponent = 1 + u (2Nd) + g4, not 0); isector 6 / 7 / 8 / 9 = worm sampling with equal-time pairs (P2 ph / pp: two random
 * times, P3 ph / pp: three; set_OperTime Trace.F90:3985-4020, replacement with the partner moved along CTQMC.F90:6068-6249;
 * estimators p2iw / p2tau / p2iwpp / p2taupp / p3iw / p3iwpp, CTQMC.F90:1297-1507); both trace paths.
 * isector 5 (H4, six worm operators), 10 (QQ, six), 11 (Q4, twelve), 12 (NQQdag, eight), 13 (QQdd, eight), 14 (Ucaca), 15 (Uccaa), 16 (QUDdag; four
 * each): worms with U-matrix factors in the proposal and random flavours of the operators in the equal-time objects (set_IEOperFlavor
 * Trace.F90:3835-3919, set_OperTime 4022-4158, get_Proposal CTQMC.F90:5698-5796, replacement 5841-5946 and 5997-6162, estimators 1166-1294 and
 * 1708-2128, u_sparsity 9003-9343); icomponent = the component's two (sector 10) or four flavour indices, not 0.  All of them run on
 * both trace paths. */
int w2dyn_ctqmc_measure(w2dyn_solver *s, int isector, int icomponent);
/* module variable `wormeta` (impurity.py:719, 755, 784): the same value for every worm sector, CTQMC.F90:194, 434 */
int w2dyn_set_wormeta(w2dyn_solver *s, double eta);
int w2dyn_get_wormeta(w2dyn_solver *s, double *eta);
/* ctqmc_worm_warmup(nwarmups2, isector, icomponent)    CTQMC.F90:8266 (allocates estimator buffers, samples nothing) */
int w2dyn_ctqmc_worm_warmup(w2dyn_solver *s, int64_t nwarmups2, int isector, int icomponent);
/* count_steps(isector, icomponent, nfix, steps_worm, steps_z)      CTQMC.F90:6296-6444
 * nfix steps of the mixed Z / worm loop for every chain, then counters reset and the starting configuration restored
 * (the random streams move on, as in the reference).  steps_worm / steps_z are the means over the chains of this
 * context, rounded so that they add up to nfix (what impurity.py:725 checks); the per-chain counts are the result
 * "steps_worm_chains" (int64, nwalkers). */
int w2dyn_count_steps(w2dyn_solver *s, int isector, int icomponent, int64_t nfix, int64_t *steps_worm, int64_t *steps_z);
/* dest_ctqmc()                                     CTQMC.F90:901 */
int w2dyn_dest_ctqmc(w2dyn_solver *s);

/* the pieces of ctqmc_measure, exposed so that a caller (bench, tests) can time / check them separately:
 * nsteps Monte-Carlo steps for every chain (CTQMC.F90:8428-8533), one pass of the Z-space estimators over every
 * chain (8541-8581), normalisation (8663-8817). */
/* sector and component of the following w2dyn_sweep / w2dyn_measure_once calls (the first lines of ctqmc_measure,
 * CTQMC.F90:8380-8400): every chain goes back to Z space; isector 1 - 16 */
int w2dyn_set_sector(w2dyn_solver *s, int isector, int icomponent);
int w2dyn_measure_begin(w2dyn_solver *s);
int w2dyn_sweep(w2dyn_solver *s, int64_t nsteps, int count_as_sampling);
int w2dyn_measure_once(w2dyn_solver *s);
int w2dyn_measure_finish(w2dyn_solver *s);
/* w2dyn_measure_once for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same kernels, inputs (the chains of the context) and outputs (the
 * context's accumulators) as w2dyn_measure_once in a Z-space run with FourPnt & 4, but the per-chain g2(nu, nu') tables
 * that k_meas_g2iw hands to k_meas_g4iw live in a static __device__ array of 32 MiB (one copy per GPU, held until the process
 * ends) instead of the buffer init_solver allocates; the call does no hipMalloc / hipFree / synchronisation and skips the
 * event timing.  chains x flavours x n2 (n2 + 2 N4iwb + 1) x 16 bytes, n2 = 2 (N4iwf + N4iwb), beyond 32 MiB -> W2DYN_EINVAL.  All contexts of a GPU
 * share the array: calls that overlap on one GPU (two contexts, two streams) race. */
int w2dyn_measure_once_noalloc(w2dyn_solver *s);

/* Cross-rank reduction input, made on the GPU.  Replaces the host side of the reference's exchange step: there every
 * result array is copied out of the module and all-reduced per quantity and per statistic
 * (w2dyn/auxiliaries/statistics.py:130-209 on the arrays of impurity.py:1198-1283).  `names`: comma-separated per-bin results
 * of the finished run ("gleg,gtau,giw,histo,g4iw,g4iw_pp,g2iw,g4iw_worm,g4iwpp_worm,giw_worm,gtau_worm" as present).  For
 * name i with B bins of P = counts[i] doubles (complex: re, im interleaved) the buffer holds at offsets[i]
 *     [ B | S = sum_b x_b (P) | sum_b (x_b - S/B)^2 (P) | S^2/B (P) ],          x_b = the normalised bin (what "<name>_bins" returns)
 * so that ONE sum-all-reduce of the whole buffer over the ranks gives count, mean and variance of every quantity:
 * sum_i |x_i - m|^2 = [sum of the third block] + [sum of the fourth block] - (sum S)^2 / (sum B).
 * *dev_ptr is device memory owned by the context (valid until the next w2dyn_reduce_pack or w2dyn_destroy), complete when
 * the call returns; hand it to RCCL as it is.  Call between ctqmc_measure and dest_ctqmc. */
int w2dyn_reduce_pack(w2dyn_solver *s, const char *names, void **dev_ptr, size_t *ndoubles, size_t *offsets, size_t *counts, int max_names);
int w2dyn_sync(w2dyn_solver *s);                    /* wait for the context's stream */

/* configuration of one chain ---------------------------------------------------------------------------------- */
/* get_mc_config_scalars() -> (n, outer_sst, outer_state)            CTQMC.F90:9358 */
int w2dyn_get_mc_config_scalars(w2dyn_solver *s, int walker, int *n, int *outer_sst, int *outer_state);
/* get_mc_config(n) -> taus, orbs, spins, cas, hashybs (time ordered; hashybs -1 = hybridising, Trace.F90:6861) */
int w2dyn_get_mc_config(w2dyn_solver *s, int walker, int n, double *taus, int *orbs, int *spins, int *cas,
                        int *hashybs, int *outer_sst, int *outer_state);
/* All chains at once: what CtHybConfig.get_from_ctqmc / set_to_ctqmc (impurity.py:699-709, w2dyn/dmft/impurity.py CtHybConfig)
 * do for the one chain of a rank between the component runs of solve_worm / solve_component.  save keeps every chain's
 * configuration (and its position in its random stream) on the host inside the context, across init_solver calls;
 * restore puts them back (same number of chains, flavours and order capacity) and rebuilds M. */
int w2dyn_save_mc_configs(w2dyn_solver *s);
int w2dyn_restore_mc_configs(w2dyn_solver *s);
/* set_mc_config(n, taus, ...) / set_empty_mc_config                 CTQMC.F90:9378, 9391; Trace.F90:6872-7000 */
int w2dyn_set_mc_config(w2dyn_solver *s, int walker, int n, const double *taus, const int *orbs, const int *spins,
                        const int *cas, const int *hashybs, int outer_sst, int outer_state);
/* M block of one chain and flavour (orb, spin 1-based), brought to the reference's time order
 * (row = creator, column = annihilator, CTQMC.F90:2303-2304); returns the order k in *k, copies k*k doubles if
 * out != NULL.  No f2py counterpart (MInv is not exported); it is what recorded-configuration parity tests read. */
int w2dyn_get_minv(w2dyn_solver *s, int walker, int orb, int spin, int *k, double *out, size_t out_doubles);
/* per-chain scalars: [0] sign of det (time ordered), [1] log|det|, [2] window position, [3] NPairs (recounted),
 * [4] number of random draws consumed, [5] cfgsign as the estimators see it */
int w2dyn_get_walker_state(w2dyn_solver *s, int walker, double *out6);

/* results = the module variables the driver reads after ctqmc_measure (impurity.py:1198-1283) ---------------- */
/* name: "gtau" (Nd,2,Ntau) | "gleg" (Nd,2,NLegMax) | "giw" (Nd,2,2Niw) complex | "occ" (Nd,2,Nd,2) |
 *       "histo" (Nd,2,MaxHisto+1) | "densitymatrix" (4^Nd,4^Nd) | "rho1" ((2Nd)^2) | "rho2" ((2Nd)^4) |
 *       "ssquare" (2Nd+1) | "ssquare_operator" (4^Nd,4^Nd) with MeasSsquare (CTQMC.F90:161-162, 2769-2840) |
 *       "gtau_mean_step", "gtau_mid_step" (Nd,2,Nmeas) | "sign_step" (Nmeas): chain-mean series (CTQMC.F90:43-44, 2394-2402) |
 *       "outersuperstatehisto" | "outerstatehisto" | "signhistosuperstates" | "signhistostates" | "tracecontribsuperstates" |
 *       "tracecontribstates" | "statessuperstates" (int32) | "eigenstatesenergies" | "occbasismapping" (4^Nd,Nd,2) int32 |
 *       "accadd" (16) | "accrem" (16) | "accpair" (Nd,2,Ntau) int64 | "cntsampling" (16) int64 | "lhisto" | "rhisto" |
 *       "gtau_full" | "gleg_full" | per-bin: "gleg_bins" (nbins,Nd,2,NLegMax), "gtau_bins", "occ_bins", "giw_bins" |
 *       FourPnt & 1: "g4tau" (Nd,2,Nd,2,N4tau,N4tau,N4tau); FourPnt & 2: "g4leg" (Nd,2,Nd,2,N4tau,N4tau,2N4iwb+1) complex |
 *       FourPnt & 4: "g4iw", "g4iw_pp" (Nd,2,Nd,2,2N4iwf,2N4iwf,2N4iwb+1) complex, "g2iw" (+ "_bins") |
 *       worm sampling (FourPnt & 8): "g4iw_worm", "g4iwpp_worm" (2N4iwf,2N4iwf,2N4iwb+1) complex (+ "_bins"),
 *       worm sector 2 (WormMeasGiw / WormMeasGtau): "giw_worm" (2Niw) complex, "gtau_worm" (Ntau) (+ "_bins": nbins leading),
 *       worm sector 3 (WormMeasGSigmaiw): "gsigmaiw_worm" (2Niw) complex (+ "_bins": nbins leading),
 *       worm sectors 6 / 7 (WormMeasP2iwPH | PP, WormMeasP2tauPH | PP): "p2iw_worm" | "p2iwpp_worm" (2N2iwb+1) complex,
 *       "p2tau_worm" | "p2taupp_worm" (Ntau); sectors 8 / 9 (WormMeasP3iwPH | PP): "p3iw_worm" | "p3iwpp_worm" (2N3iwf,2N3iwb+1)
 *       complex (+ "_bins": scalar "nbe" leading),
 *       worm sectors 5, 10 - 16: "h4iw_worm" | "qqqq_worm" (2N4iwf,2N4iwf,2N4iwb+1), "qq_worm" | "quddag_worm" (2Niw), "nqqdag_worm" (2N3iwf,2N3iwb+1),
 *       "qqdd_worm" (2N3iwf,2N3iwf), "ucaca_worm" | "uccaa_worm" (2N2iwb+1) complex, "qqtau_worm" | "ucacatau_worm" | "uccaatau_worm" (Ntau)
 *       (+ "_bins": scalar "nbie" leading),
 *       "accwormadd" "accwormrem" "accwormrep" (16), "steps_worm_chains" (int64), "worm_counters" per chain
 * Writes min(bytes, size) bytes; *needed (if not NULL) receives the full size. */
int w2dyn_get_result(w2dyn_solver *s, const char *name, void *out, size_t bytes, size_t *needed);
/* scalars: "mean_sign" "aborted" "time_warmup" "time_sim" "time_sim_steps" "accglob" "accshift" "accadd4"
 *          "accrem4" "accflavc" "nwalkers" "nbins" "kmax" "steps_done" "meas_done"
 *          "trace_path" (after init_solver: 0 = closed-form local trace of one-state blocks, 2 | 3 | 4 = tree of block products of that size, 9 = sequential) */
int w2dyn_get_scalar(w2dyn_solver *s, const char *name, double *out);

/* timing of the last w2dyn_sweep / w2dyn_measure_once launches, measured with HIP events on the context's own
 * stream: average kernel duration in ms and number of launches since the last reset (bench.py roofline leg) */
/* timing: off by default; on: every launch is bracketed by a pair of HIP events on the context's stream.  Launches stay
 * asynchronous - the pairs are read out when the stream is waited for anyway (w2dyn_sync, w2dyn_kernel_time) */
int w2dyn_set_timing(w2dyn_solver *s, int on);
int w2dyn_kernel_time(w2dyn_solver *s, const char *kernel /* "sweep" | "sweep_worm" | "sweep_calibrate" | "meas_gleg" | "meas_misc" | "meas_giw" | "meas_g2iw" | "meas_g4iw" | "meas_g4w" | "meas_gw" | "meas_gtau_ac" | "meas_g4leg" */,
                      double *avg_ms, int64_t *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
