// Device side of libw2dyn_hip.so: batched-walker CT-HYB sampler and estimators for gfx950 (MI355X).
//
// Execution model: ONE WAVEFRONT (64 lanes) PER MARKOV CHAIN.  All control flow of a chain is wave-uniform; lanes
// parallelise over matrix columns / operators.  A 256-thread block holds four independent chains (no block-level
// barrier anywhere).  Chain state stays resident in HBM between launches (structure of arrays over walkers).  During
// a sweep launch the chain's operator times, orders and counters live in the wave's LDS slice; the inverse
// hybridisation matrices stay in global memory (L2 resident: k_f^2 doubles per flavour) and are touched with
// coalesced row segments (lane = column).
//
// Data layout per walker w (capacity KMAX per flavour):
//   K [w][f]                      expansion order of flavour f
//   TC[w][f][KMAX], TA[...]       creator / annihilator times in SLOT order (insertion order, swap-with-last removal)
//   M [w][f][KMAX*KMAX]           inverse hybridisation matrix, row = creator slot, column = annihilator slot
// Slot order replaces the reference's time-ordered rows/columns (MatrixUpdate.F90:410-482 shifts and reallocates on
// every accepted move): an insertion appends one row/column in place, a removal is a rank-one update plus moving
// the last row/column into the hole, a tau-shift is a pure sign flip.  Time-ordered ranks, where the reference
// needs them (FPos parity, Trace.F90:2828), are obtained by wave-parallel counting (ballot + popcount).
//
// Local problem: density-density, every block of H_loc holds one Fock state (reference "scalar" trace path,
// Trace.F90:5556-5727).  The local trace ratio of a pair move is evaluated in closed form from occupied-length
// integrals (no time-ordered walk), which is exact for this class of Hamiltonians.
//
// Random numbers: the chain's Philox stream (include/w2dyn_rng.h) is generated 128 draws at a time, lane i
// evaluating block 64*chunk+i, and handed out with v_readlane - the stream itself is unchanged.
#pragma once
#include <hip/hip_runtime.h>
#if defined(W2DYN_STAMPS) && !defined(W2DYN_STAMPS_GLOB) && !defined(W2DYN_STAMPS_SEQ) && !defined(W2DYN_STAMPS_ACC)
#define DBG_STAMP(x, c) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((x).lane == 0) (x).cnt[c] += t_ - (x).tstamp; (x).tstamp = t_; } while (0)
#define DBG_START(x) (x).tstamp = __builtin_amdgcn_s_memtime()
#else
#define DBG_STAMP(x, c)
#define DBG_START(x)
#endif
#ifdef W2DYN_STAMPS_ACC       // diagnostic build: stage stamps inside the accepted-move paths (add: DBG0-3, rem: DBG4-7)
#define ACC_STAMP(x, c) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((x).lane == 0) (x).cnt[c] += t_ - (x).tstamp; (x).tstamp = t_; } while (0)
#define ACC_START(x) (x).tstamp = __builtin_amdgcn_s_memtime()
#else
#define ACC_STAMP(x, c)
#define ACC_START(x)
#endif
#ifdef W2DYN_STAMPS_GLOB      // diagnostic build: the stage stamps time the global move instead of the pair moves
#define GLOB_STAMP(x, c) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((x).lane == 0) (x).cnt[c] += t_ - (x).tstamp; (x).tstamp = t_; } while (0)
#define GLOB_START(x) (x).tstamp = __builtin_amdgcn_s_memtime()
#else
#define GLOB_STAMP(x, c)
#define GLOB_START(x)
#endif
#include <stdint.h>
#include "../../include/w2dyn_rng.h"

// minimum waves per SIMD asked of the compiler for the sweep kernels (second __launch_bounds__ argument).  The worm
// kernel needs 281 registers when left alone, which halves its occupancy; held to 256 it spills 116 bytes that only the
// rare moves touch and runs 2.4x faster (C5 worm, measured).  The matrix-path kernels (264-324 registers when left alone,
// 36-276 bytes spilled at 256) gain as much once the host picks the LDS layout that lets two workgroups share a CU.
#ifndef W2DYN_WPE_MAT
#define W2DYN_WPE_MAT 2
#endif
#ifndef W2DYN_WPE_WORM
#define W2DYN_WPE_WORM 2
#endif

namespace w2dyn {

// the sweep kernel is one flat function per instantiation: a real call would force the chain context into scratch
#define W2DYN_INLINE __attribute__((always_inline)) inline

constexpr int MAXFLAV = 16;
constexpr int KMAX_LIMIT = 256;
constexpr int GLEG_SEG_BLOCK = 4096;     // k_gleg_segments: segments (bin, flavour) per workgroup
constexpr int GLEG_RED_ROWS = 8, GLEG_RED_STRIDE = 65;      // k_meas_gleg: LDS transpose tile of the end-of-segment reduction
constexpr int G2_FC = 32;                                    // k_meas_g2iw: frequency chunk of the two-step transform
// entries of a (chain, flavour)'s g2 table: n2 x n2, then its 2 nwb + 1 central diagonals as rows (k_meas_g2iw)
__host__ __device__ inline size_t g2_stride(int n2, int nwb) { return (size_t)n2 * (size_t)(n2 + 2 * nwb + 1); }

enum Counter { C_TRYADD, C_ACCADD, C_QNADD, C_TRYREM, C_ACCREM, C_QNREM, C_TRYGLOB, C_ACCGLOB, C_QNGLOB,
               C_TRYSHIFT, C_ACCSHIFT, C_SAMPLING,
               C_CYC_ADD, C_CYC_REM, C_CYC_GLOB, C_CYC_SHIFT,   // shader cycles per move type (diagnostic build -DW2DYN_STAMPS only)
               C_DBG0, C_DBG1, C_DBG2, C_DBG3, C_DBG4, C_DBG5, C_DBG6, C_DBG7,   // stage stamps (diagnostic build)
               // worm sampling (sectors 2 and 4): steps and pair moves inside worm space, worm insert / remove / replace
               C_W_SAMPLING, C_W_TRYADD, C_W_ACCADD, C_W_TRYREM, C_W_ACCREM,
               C_TRYWADD, C_ACCWADD, C_QNWADD, C_TRYWREM, C_ACCWREM, C_QNWREM, C_TRYWREP, C_ACCWREP,
               C_TRYADD4, C_ACCADD4, C_TRYREM4, C_ACCREM4,       // two pairs at once (StepAdd4 / StepRem4)
               C_TRYFLAVC, C_ACCFLAVC,                           // StepFlavourchange_general (offdiag with flavourchange_moves)
               C_NCOUNT };
static_assert(C_NCOUNT <= 64, "one lane per event counter");

enum WalkerErr { ERR_NONE = 0, ERR_KMAX = 1, ERR_WCAP = 2, ERR_DETRAT = 4, ERR_SINGULAR = 8, ERR_BLOCKSEQ = 16 };

// number of worm operators a chain's record holds (NOperWorm, Parameters.F90:45: at most 12, sector 11) and the number a kernel keeps
// in registers: 4 in every translation unit but the one of the sectors with 6, 8 and 12 worm operators
constexpr int NW_STORE = 12;

#ifndef W2DYN_NWMAX
#define W2DYN_NWMAX 4
#endif
constexpr int NWMAX = W2DYN_NWMAX;
__host__ __device__ __forceinline__ int worm_nops_of(int sector)
{
    return sector == 2 ? 2 : ((sector == 5 || sector == 10) ? 6 : (sector == 11 ? 12 : ((sector == 12 || sector == 13) ? 8 : 4)));
}

struct WalkerScalars {
    unsigned long long ndraw;   // random numbers consumed
    int outer;                  // outer superstate = Fock state at tau = 0/beta
    int npairs;                 // removable pairs in the window, -1 = stale (Trace.F90:1634, 1670)
    int wpos;                   // window position
    int err;
    double det_sign;            // sign of det F in the reference's time order (Trace%Det%sign)
    double det_log;             // log|det F|
    unsigned long long cnt[C_NCOUNT];
    // worm sampling: current sector (0/1 = Z, 2 = one-particle worm c c+, 4 = two-particle worm c c+ c c+, 6 - 9 = P2 / P2pp / P3 / P3pp, 3 / 5 /
    // 10 - 16 = worms with U-matrix factors, Parameters.F90:28-45) and the worm operators in wormContainer order (Trace.F90:123-125): times
    // and flavours (the first worm_nops_of(sector) entries count)
    int sector; int wflav[NW_STORE]; double wtau[NW_STORE];
    // steps of an interrupted sweep launch still to do: a chain stops (ERR_KMAX set) as soon as one of its flavours has filled
    // its kmax slots; the host enlarges the arrays and lets it finish (w2dyn_hip.hip: finish_paused)
    long long todo;
};

struct DevConst {
    int nd, nflav, nstates, nftau, ntau, nleg, niw2, kmax, nwalkers, num_windows, nslide, has_szt, maxhisto;
    int lane_shift;          // log2 of the lanes that share one flavour in level_integral: 64 / (nflav rounded up to a power of two)
    int walkers_per_bin, nbins, lds_per_wave, wcap, kgj;   // lds_per_wave in bytes; wcap: window-list capacity; kgj: LDS inversion order limit
    double beta, rbeta /* 1/beta, correctly rounded */, taudiff_max, p_glob, p_shift, egs;
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
    int *flags;                        // [0] largest expansion order any chain has stored, [1] chains paused at capacity
    // accumulators (per bin)
    double *acc_gleg;     // [nbins][nflav][nleg]
    double *acc_gtau;     // [nbins][nflav][ntau]
    double *acc_giw;      // [nbins][nflav][niw2][2]
    double *acc_histo;    // [nbins][nflav][maxhisto+1]
    double *acc_dm;       // [nbins][nstates]
    double *acc_outer;    // [nbins][nstates]
    double *acc_signouter;// [nbins][nstates]
    double *acc_sign;     // [nbins]
    double *acc_signstate;// [nbins][nstates] sign per outer eigenstate, block order (SignHistoStates, CTQMC.F90:3313-3319)
    double *acc_tcstate;  // [nbins][nstates] partial-trace weights per outer eigenstate (TraceContribStates, 3332-3337)
    unsigned long long *accpair;  // [nflav][ntau]
    // local problem in block form.  nouter = number of outer blocks (= nstates for one-state blocks); dmsize = doubles of
    // the block-diagonal density matrix; mat != 0 selects the matrix local-trace path (cthyb_matrix.hip.h)
    int nouter, dmsize, mat, nsst, nsm, shared_lds, op_in_lds;
    const int *shift;             // [nsst][nflav][2] block reached by quantum-number arithmetic (local_problem.hpp)
    const int *sst_n, *sst_off, *conn, *sst_first, *sst_of_state, *dmoff;   // [nsst] [nsst] [nsst][nflav][2] [nsst] [nstates] [nsst+1]
    const double *sst_eval;       // [nstates] eigenvalues in block order
    const double *opmat;          // [nsst][nflav][2][nsm][nsm] operators in the eigenbasis, row = target state, zero padded
    double *kcache;               // blocks of more than 4 states: [nwalkers][MAT_CK_ENTRIES][NT*NT+1] checkpoints of the window part (mat_trace_seq_window)
    const double *opmat_pad;      // blocks of more than 4 states: the same with rows and columns padded to mat_seq_pad(nsm) (sequential evaluator)
    // worm sampling (CTQMC.F90:335-378, 434; ctqmc_measure arguments)
    int isector, icomponent, nwf, nwb, wormphconv, g4ph, g4pp;
    // user symmetry moves of the global move (NSymMove, Trace.F90:403-404, 471-478): 2 nsym rows (the listed moves, then their
    // inverses) of nflav entries, 1-based flavour numbers in the order band + Nd * spin
    int nsym; const int *symmoves;
    // moves outside the default mix (CTQMC.F90:332-333, 8441-8442): probability of a two-pair move in place of a pair move, of a pair
    // move that may change the outer block; xmoves: either is set (the sweep runs the k_sweep<..., XM = true> kernels, which keep
    // two free slots per flavour instead of one)
    double p_4op, p_outer; int xmoves;
    // off-diagonal hybridisation (offdiag = 1, cthyb_offdiag.hip.h): F(tau) with both flavour indices [nflav][nflav][nftau] as handed
    // to init_solver, flavour of every creator / annihilator slot [nwalkers][kmax]
    int offdiag; const double *ffull; int *FC, *FA;
    const double *umat;           // U matrix (2Nd)^4 as handed to init_solver (flavour = 2 * orbital + spin): proposal factor of the GSigma worm
    // matrix trace path: the chains of a launch are handed out to the waves from this counter (zeroed before every launch), so
    // that the four waves of a workgroup - which share the staged tables of the local problem - never wait for one another
    int *queue;
    double p_wins, p_wrep, worm_eta;
    double *acc_g4w, *acc_g4wpp;  // [nbins][2nwf][2nwf][2nwb+1] complex
    double *acc_giww, *acc_gtauw; // [nbins][niw2] complex, [nbins][ntau]: giw_worm / gtau_worm of the sampled component (sector 2)
    double *acc_nz, *acc_nw;      // [nbins] number of Z / worm measurements (chains measure in the sector they are in)
    // AccPair histogram (CTQMC.F90:65-70): naccpair bins over [0, accpair_max]; Ntau bins over beta outside the window
    // calibration.  calib: 0 = off, 1 = phase 1 (per-chain list of accepted pair distances), 2 = phase 2 (histogram on
    // the calibrated window; per-chain, per-flavour counts decide when a chain is done), CTQMC.F90:8088-8178
    int naccpair, calib, napt, phase2pairnum;
    int sweep_wpe3;               // scalar path: run the sweep kernel variant with three waves per SIMD (k_sweep<0, false, true>)
    double accpair_max;
    double *apt_tau;              // [nwalkers][napt]
    int *apt_cnt;                 // [nwalkers]
    int *calib_pairs;             // [nwalkers][nflav]
    // [nflav] bit g of entry f: flavours f and g have the same F(tau) table to the last bit, so a relabelling f -> g without the exchange of
    // creators and annihilators carries the inverse hybridisation matrix along (step_glob)
    const int *fsame;
    const int *fclass;            // offdiag: [nflav][nflav] class of the two-flavour table F_ac(tau), equal for tables identical to the last bit
};

// bytes of LDS one wave needs (host and device agree through this one function)
__host__ __device__ inline size_t lds_bytes_per_wave(int nflav, int kmax, int wcap, int kgj, int mat)
{
    // tc, ta, MQ, RM | { Q, R, wt, wm, wc, wp } and the Gauss-Jordan tile gj | counters, K, pivots.  On the scalar trace path
    // gj lies on top of Q..wc (a full rebuild invalidates the window list anyway); the matrix path keeps data in wm/wc
    // across the rebuild and gets a tile of its own
    size_t u1 = ((size_t)2 * kmax + wcap) * sizeof(double) + (size_t)3 * wcap * sizeof(int), u2 = (size_t)kgj * kgj * sizeof(double);
    size_t d = (size_t)2 * nflav * kmax + 2 * (size_t)kmax + ((mat ? u1 + u2 : (u1 > u2 ? u1 : u2)) + 7) / 8;
    size_t i = (size_t)MAXFLAV + kmax;                                         // K, piv
    if (mat) { d += wcap + 2; i += (size_t)3 * wcap + 8; }                     // ot; om, osst, tsst
    if (mat > 4) { const size_t nt = mat <= 8 ? 8 : (mat <= 10 ? 10 : 16); d += (16 + 64) * nt + nt * nt + 2; }   // vector tile + decay table (mat_seq_lds_doubles) + product outside the window (sequential trace, mat = block size)
    size_t b = d * sizeof(double) + C_NCOUNT * sizeof(unsigned long long) + i * sizeof(int);
    return (b + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------------------------
// hand-off through LDS only: orders and drains the wave's LDS operations (s_waitcnt lgkmcnt(0)) and leaves global loads
// in flight - wave_sync() below also waits for every outstanding global access (vmcnt(0)), which undoes a prefetch
__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void wave_sync()
{
    // cross-lane hand-off through LDS / global memory inside ONE wave: order the memory operations for the
    // compiler and drain outstanding stores; no s_barrier is needed (a wave executes in lockstep).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, result wave-uniform.  DPP inside the 16-lane rows (quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror), then four v_readlane.  Must be called with all lanes active.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ double wave_min(double v)
{
    v = fmin(v, dpp_d<0xB1>(v));
    v = fmin(v, dpp_d<0x4E>(v));
    v = fmin(v, dpp_d<0x141>(v));
    v = fmin(v, dpp_d<0x140>(v));
    return fmin(fmin(readlane_d(v, 0), readlane_d(v, 16)), fmin(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_d<0xB1>(v));
    v = fmax(v, dpp_d<0x4E>(v));
    v = fmax(v, dpp_d<0x141>(v));
    v = fmax(v, dpp_d<0x140>(v));
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ int count_pred(bool p) { return __popcll(__ballot(p)); }

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
    double *M;                       // global: this walker's matrices
    // LDS slice of this wave
    double *tc_, *ta_, *Q, *R, *MQ, *RM, *wt, *gj;
    int *wm, *wc, *wp, *K, *piv;
    unsigned long long *cnt;
    unsigned int cntreg;
    // wave-uniform scalars
    unsigned long long chunk;        // index of the 128-draw chunk held in rbuf
    int rpos;                        // next draw inside the chunk (0..128)
    double rbuf0, rbuf1;             // lane i: draws 2i and 2i+1 of the chunk
    int outer, npairs, wpos, err, nwin;
    double det_sign, det_log;
    double wmin, wmax;
    unsigned long long tstamp;
    int kreg;                        // lane g: expansion order of flavour g
    double wr_t; int wr_m, wr_c, wr_p;  // lane e: window-list entry e (tau, f|ca<<8|slot<<16, partner count, pairs before e) when nwin <= 64
    int wvalid;                      // window list (registers or LDS) matches the current configuration and window
    // matrix local-trace path
    double *opbuf;                                   // sequential evaluator (blocks of more than 4 states): vector tile and decay table, mat_seq_lds_doubles()
    double *pmat; int pvalid, p_lo, p_hi;            // sequential trace: product of the operators outside the window (mat_trace_seq_window), its blocks
    double *kc; int kval, k_i1, k_first;             // its checkpoints inside the window (global memory): window operators covered, of the last trial, first changed trial index
    double *ot; int *om, *osst, *tsst; int nops;     // time-ordered operator list, block sequence, trial block sequence (LDS)
    const short *t_conn, *t_shift;                   // 16-bit block numbers: the two tables are the bulk of the shared LDS (5 orbitals: 352 blocks x 20 operators)
    const int *t_n, *t_off; const double *t_eval, *t_op;   // block-shared tables (LDS); t_op null: operators stay in global memory
    double tr_log, tr_sign;                          // current local trace
    int ms_, mi_, spb_;                              // lane = (outer state ms_ of the batch, component mi_); spb_ outer states per batch:
                                                     // all of them (nsm) for blocks of up to 8 states, 64 / nsm beyond (9..16 states)
    // worm sampling (kernels of the worms with more than four operators are built with W2DYN_NWMAX = 12, tu_worm_big.hip)
    int sector; double wtau[NWMAX]; int wfl[NWMAX];

    __device__ Ctx(const DevConst &c) : C(c) {}

    __device__ __forceinline__ void rng_fill()
    {
        unsigned long long blk = chunk * 64ull + (unsigned long long)lane;
        unsigned long long wk = C.first_walker + (unsigned long long)w;
        uint32_t c[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)wk, (uint32_t)(wk >> 32) };
        w2dyn_philox4x32_10(c, (uint32_t)C.seed, (uint32_t)(C.seed >> 32));
        unsigned long long m0 = ((unsigned long long)(c[0] >> 5) << 26) | (unsigned long long)(c[1] >> 6);
        unsigned long long m1 = ((unsigned long long)(c[2] >> 5) << 26) | (unsigned long long)(c[3] >> 6);
        rbuf0 = (double)m0 * (1.0 / 9007199254740992.0);
        rbuf1 = (double)m1 * (1.0 / 9007199254740992.0);
    }
    __device__ __forceinline__ void rng_seek(unsigned long long ndraw)
    {
        chunk = ndraw >> 7; rpos = (int)(ndraw & 127ull);
        rng_fill();
    }
    __device__ __forceinline__ unsigned long long rng_tell() const { return (chunk << 7) + (unsigned long long)rpos; }
    __device__ __forceinline__ double draw()
    {
        if (rpos == 128) { chunk += 1; rpos = 0; rng_fill(); }
        int l = rpos >> 1;
        double u = (rpos & 1) ? readlane_d(rbuf1, l) : readlane_d(rbuf0, l);
        rpos += 1;
        return u;
    }
    // n draws whose values are not used (the reference draws them, CTQMC.F90:8442-8443): the stream moves on
    __device__ __forceinline__ void skip(int n)
    {
        rpos += n;
        if (rpos > 128) { chunk += 1; rpos -= 128; rng_fill(); }
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
    __device__ __forceinline__ double *tc(int f) const { return tc_ + f * C.kmax; }
    __device__ __forceinline__ double *ta(int f) const { return ta_ + f * C.kmax; }
    __device__ __forceinline__ double *mat(int f) const { return M + (size_t)f * C.kmax * C.kmax; }
    __device__ __forceinline__ int order(int f) const { return __builtin_amdgcn_readlane(kreg, f); }
    __device__ __forceinline__ void set_order(int f, int k) { if (lane == f) kreg = k; }
    // event counters of this launch: lane c holds counter c (no LDS round trip per event); added to cnt[] in ctx_store
    __device__ __forceinline__ void count(int c) { cntreg += (lane == c) ? 1u : 0u; }
};

// pair (x,y) of one creator and one annihilator of equal flavour counts as removable (Trace.F90:1794-1814)
__device__ __forceinline__ bool pair_ok(double x, double y, double wmin, double wmax, double tdm)
{
    double a = fmin(x, y), b = fmax(x, y);
    return (a >= wmin) && (b <= wmax) && !(b > a + tdm);
}

// Collect the operators inside the window (wt: tau, wm: f | ca<<8 | slot<<16) and count, for each of them, its partners
// at later times (wc).  Returns the total = NPairs (Trace.F90:1770-1815); x.nwin = list length.  Up to 64 entries
// the list lives in registers (lane e = entry e) and is scanned with v_readlane; longer lists stay in LDS.
__device__ W2DYN_INLINE int build_window(Ctx &x)
{
    const DevConst &C = x.C;
    int nw = 0;
    for (int f = 0; f < C.nflav; ++f) {
        int k = x.order(f);
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
            if (inw && pos < C.wcap) { x.wt[pos] = tau; x.wm[pos] = f | (ca << 8) | (slot << 16); }
            nw += __popcll(mask);
        }
    }
    nw = uni(nw);
    if (nw > C.wcap) { x.err |= ERR_WCAP; nw = C.wcap; }
    x.nwin = nw;
    wave_sync();
    int total = 0;
    if (nw <= 64) {
        bool live = x.lane < nw;
        double te = live ? x.wt[x.lane] : -1.0; int me = live ? x.wm[x.lane] : -1;
        double thi = te + C.taudiff_max;
        int want = (me & 0xffff) ^ 0x300;                // same flavour, other type (ca 1 <-> 2)
        int c = 0;
        for (int q = 0; q < nw; ++q) {
            double tq = readlane_d(te, q); int mq = __builtin_amdgcn_readlane(me, q) & 0xffff;
            c += (live && mq == want && tq > te && !(tq > thi)) ? 1 : 0;
        }
        int P = 0;                                       // pairs whose earlier operator precedes entry e in time
        for (int q = 0; q < nw; ++q) {
            double tq = readlane_d(te, q); int cq = __builtin_amdgcn_readlane(c, q);
            P += (tq < te) ? cq : 0;
        }
        x.wr_t = te; x.wr_m = me; x.wr_c = c; x.wr_p = P;
        total = wave_sum(c);
    } else {
        for (int base = 0; base < nw; base += 64) {
            int e = base + x.lane;
            int c = 0;
            if (e < nw) {
                double te = x.wt[e]; int me = x.wm[e] & 0xffff;
                double thi = te + C.taudiff_max;
                int want = me ^ 0x300;
                for (int q = 0; q < nw; ++q) {
                    double tq = x.wt[q]; int mq = x.wm[q] & 0xffff;
                    c += (mq == want && tq > te && !(tq > thi)) ? 1 : 0;
                }
                x.wc[e] = c;
            }
            total += c;
        }
        total = wave_sum(total);
        wave_sync();
        for (int base = 0; base < nw; base += 64) {      // pairs whose earlier operator precedes entry e in time
            int e = base + x.lane;
            if (e < nw) {
                double te = x.wt[e];
                int P = 0;
                for (int q = 0; q < nw; ++q) P += (x.wt[q] < te) ? x.wc[q] : 0;
                x.wp[e] = P;
            }
        }
        wave_sync();
    }
    x.wvalid = 1;
    return total;
}

// Accepted pair moves keep the register-resident window list current instead of invalidating it (a rebuild costs a
// scan of every flavour plus two passes over the list).  The list order is irrelevant to its users (pairs are enumerated
// by time), so new operators are appended and removed ones are filled from the tail.  Partner counts c and prefix counts
// P (pairs whose earlier operator precedes the entry) are corrected by what the two operators change; the result is
// what build_window computes for the new configuration.  Lists beyond 64 entries (LDS) are still rebuilt.
__device__ W2DYN_INLINE void window_list_add(Ctx &x, int f, int slot, double tnc, double tna)
{
    const int nw = x.nwin;
    if (!x.wvalid || nw + 2 > 64) { x.wvalid = 0; return; }
    const double tm = x.C.taudiff_max;
    const bool live = x.lane < nw;
    const double te = x.wr_t, thi = te + tm;
    const int fc = x.wr_m & 0xffff;
    const bool isa = live && fc == (f | 0x200), isc = live && fc == (f | 0x100);
    // an entry gains a partner for each new operator of the other type at a later time within taudiff_max
    const int inc = ((isa && tnc > te && !(tnc > thi)) ? 1 : 0) + ((isc && tna > te && !(tna > thi)) ? 1 : 0);
    const int cnc = __popcll(__ballot(isa && te > tnc && !(te > tnc + tm))) + ((tna > tnc && !(tna > tnc + tm)) ? 1 : 0);
    const int cna = __popcll(__ballot(isc && te > tna && !(te > tna + tm))) + ((tnc > tna && !(tnc > tna + tm)) ? 1 : 0);
    int c = x.wr_c + inc, P = x.wr_p;
    unsigned long long mk = __ballot(inc != 0);
    while (mk) {
        const int q = __ffsll((long long)mk) - 1; mk &= mk - 1ull;
        const double tq = readlane_d(te, q); const int iq = __builtin_amdgcn_readlane(inc, q);
        P += (tq < te) ? iq : 0;
    }
    P += ((tnc < te) ? cnc : 0) + ((tna < te) ? cna : 0);
    const int Pnc = wave_sum((live && te < tnc) ? c : 0) + ((tna < tnc) ? cna : 0);
    const int Pna = wave_sum((live && te < tna) ? c : 0) + ((tnc < tna) ? cnc : 0);
    if (x.lane == nw) { x.wr_t = tnc; x.wr_m = f | 0x100 | (slot << 16); c = cnc; P = Pnc; }
    if (x.lane == nw + 1) { x.wr_t = tna; x.wr_m = f | 0x200 | (slot << 16); c = cna; P = Pna; }
    x.wr_c = c; x.wr_p = P;
    x.nwin = nw + 2;
}

// e1, e2: list entries of the removed pair; ic / ia: its creator / annihilator slot; last: slot that moves into the hole
__device__ W2DYN_INLINE void window_list_rem(Ctx &x, int f, int e1, int e2, int ic, int ia, int last, double tau_c, double tau_a)
{
    const int nw = x.nwin;
    if (!x.wvalid || nw > 64 || e1 < 0 || e2 < 0) { x.wvalid = 0; return; }
    const double tm = x.C.taudiff_max;
    const bool live = x.lane < nw && x.lane != e1 && x.lane != e2;
    double te = x.wr_t; const double thi = te + tm;
    int me = x.wr_m, c = x.wr_c, P = x.wr_p;
    const int fc = me & 0xffff;
    const bool isa = live && fc == (f | 0x200), isc = live && fc == (f | 0x100);
    const int dec = ((isa && tau_c > te && !(tau_c > thi)) ? 1 : 0) + ((isc && tau_a > te && !(tau_a > thi)) ? 1 : 0);
    const int c1 = __builtin_amdgcn_readlane(c, e1), c2 = __builtin_amdgcn_readlane(c, e2);
    const double t1 = readlane_d(te, e1), t2 = readlane_d(te, e2);
    P -= ((t1 < te) ? c1 : 0) + ((t2 < te) ? c2 : 0);
    unsigned long long mk = __ballot(dec != 0);
    while (mk) {
        const int q = __ffsll((long long)mk) - 1; mk &= mk - 1ull;
        const double tq = readlane_d(te, q); const int dq = __builtin_amdgcn_readlane(dec, q);
        P -= (tq < te) ? dq : 0;
    }
    c -= dec;
    // swap-with-last of the slot arrays: the operators that sat in slot `last` now sit in the freed slots
    if (isc && (me >> 16) == last) me = fc | (ic << 16);
    if (isa && (me >> 16) == last) me = fc | (ia << 16);
    // fill the two holes from the tail
    const int h0 = e1 < e2 ? e1 : e2, h1 = e1 < e2 ? e2 : e1, nn = nw - 2;
    int d0 = -1, s0 = 0, d1 = -1, s1 = 0;
    if (h1 < nn) { d0 = h0; s0 = nn; d1 = h1; s1 = nn + 1; }
    else if (h0 < nn) { d0 = h0; s0 = (h1 == nn) ? nn + 1 : nn; }
    if (d0 >= 0) {
        const double tv = readlane_d(te, s0); const int mv = __builtin_amdgcn_readlane(me, s0), cv = __builtin_amdgcn_readlane(c, s0), pv = __builtin_amdgcn_readlane(P, s0);
        if (x.lane == d0) { te = tv; me = mv; c = cv; P = pv; }
    }
    if (d1 >= 0) {
        const double tv = readlane_d(te, s1); const int mv = __builtin_amdgcn_readlane(me, s1), cv = __builtin_amdgcn_readlane(c, s1), pv = __builtin_amdgcn_readlane(P, s1);
        if (x.lane == d1) { te = tv; me = mv; c = cv; P = pv; }
    }
    if (x.lane >= nn) { te = -1.0; me = -1; c = 0; P = 0; }
    x.wr_t = te; x.wr_m = me; x.wr_c = c; x.wr_p = P;
    x.nwin = nn;
}

// integral over (lo,hi) of  eps_f + sum_g U_fg n_g(tau)   for the current configuration (flavour f itself excluded).
// All other flavours are handled in one pass: the 64 lanes are split into groups of 2^lane_shift lanes, one group per
// flavour, each lane striding over its flavour's pairs; the coupling U_fg is loaded per lane while the loop runs.
template <int WORM = 0>      // WORM: number of worm operators of the sampled sector (0 = pure Z-space kernel)
__device__ W2DYN_INLINE double level_integral(Ctx &x, int f, double lo, double hi)
{
    const DevConst &C = x.C;
    const int g = x.lane >> C.lane_shift, sub = x.lane & ((1 << C.lane_shift) - 1);
    const bool act = g < C.nflav && g != f;
    const int gc = act ? g : 0;
    const double u = act ? C.udd[f * C.nflav + gc] : 0.0;
    const int kall = __builtin_amdgcn_ds_bpermute(gc << 2, x.kreg);     // by all lanes: a masked-off source lane would read as 0
    const int k = act ? kall : 0;
    const double *tcp = x.tc_ + gc * C.kmax, *tap = x.ta_ + gc * C.kmax;
    double acc = (act && sub == 0 && ((x.outer >> flav_bit(gc, C.nd)) & 1)) ? hi - lo : 0.0;
    for (int e = sub; e < k; e += 1 << C.lane_shift) {
        double c = fmin(fmax(tcp[e], lo), hi), a = fmin(fmax(tap[e], lo), hi);
        acc += (hi - c) - (hi - a);
    }
    double part = u * acc;
    if constexpr (WORM) {
        // worm operators of the other flavours change the occupations too (wormContainer order: c, c+, c, c+)
        if (x.sector != 1 && x.lane == 0) {
#pragma unroll
            for (int i = 0; i < WORM; ++i) {
                const int gw = x.wfl[i];
                if (gw == f) continue;
                const double t = fmin(fmax(x.wtau[i], lo), hi);
                part += ((i & 1) ? 1.0 : -1.0) * C.udd[f * C.nflav + gw] * (hi - t);
            }
        }
    }
    return C.eps[f] * (hi - lo) + wave_sum(part);
}

template <bool CALIB>
__device__ __forceinline__ void accpair_add(Ctx &x, int f, double dt)
{
    const DevConst &C = x.C;
    if (x.lane == 0) {
        int bin = (int)floor(dt / C.accpair_max * (double)C.naccpair) + 1;
        if (bin > C.naccpair) bin = C.naccpair;
        atomicAdd(&C.accpair[(size_t)f * C.naccpair + (bin - 1)], 1ull);
        if constexpr (CALIB) {
            if (C.calib == 1) {             // AccPairTau, CTQMC.F90:3583-3588, 4123-4128
                const int i = C.apt_cnt[x.w];
                if (i < C.napt) { C.apt_tau[(size_t)x.w * C.napt + i] = dt; C.apt_cnt[x.w] = i + 1; }
            } else C.calib_pairs[(size_t)x.w * C.nflav + f] += 1;
        }
    }
}

#include "cthyb_matrix.hip.h"

// ---------------------------------------------------------------------------------------------------------------
// pair insertion: CTQMC.F90:3353-3612 (StepAdd), Trace.F90:3657-3698, 2775-2832, MatrixUpdate.F90:392-442
// ---------------------------------------------------------------------------------------------------------------
#ifndef ADD_UNROLL
#define ADD_UNROLL 8
#endif
template <int MAT, int WORM = 0, bool CALIB = false, bool XM = false>
__device__ W2DYN_INLINE void step_add(Ctx &x, bool outer = false)
{
    const DevConst &C = x.C;
    if (WORM && x.sector != 1) x.count(C_W_TRYADD); else x.count(C_TRYADD);
    DBG_START(x);
    double width1 = x.wmax - x.wmin;
    double tau1 = x.wmin + width1 * x.draw();                 // annihilator
    int orb = x.randint(1, C.nd);
    int spin = x.randint(1, 2);
    double tmin = fmax(tau1 - C.taudiff_max, x.wmin);
    double width2 = fmin(tau1 + C.taudiff_max, x.wmax) - tmin;
    double tau2 = tmin + width2 * x.draw();                   // creator
    double tdf = width1 * width2;
    int f = 2 * (orb - 1) + (spin - 1);
    int k = x.order(f);
    const int kmax = C.kmax;
    double *tcp = x.tc(f), *tap = x.ta(f);

    // quantum numbers (Trace.F90:1243-1302 for one-state blocks): no operator of flavour f between the new pair and
    // the right occupation below it; equal times are rejected like QN violations (CTQMC.F90:3405-3416)
    double lo = fmin(tau1, tau2), hi = fmax(tau1, tau2);
    bool creator_first = tau2 < tau1;
    bool valid;
    int p_lo = 0, p_hi = 0;                              // MAT: insertion indices of the two operators in the old list
    int outer_new = x.outer;                             // XM: outer block / outer Fock state after a move that changes it
    if constexpr (MAT) {
        // general blocks: walk the block sequence over the changed span (Trace.F90:1243-1302); equal times are rejected
        bool eq = (tau1 == tau2);
        for (int base = 0; base < x.nops; base += 64) {
            int e = base + x.lane;
            double t = e < x.nops ? x.ot[e] : -1.0;
            eq |= __ballot(t == tau1 || t == tau2) != 0ull;
        }
        p_lo = mat_lower(x, lo); p_hi = mat_lower(x, hi);
        const int nf2 = 2 * C.nflav;
        const int code_lo = mat_code(f, creator_first ? 1 : 2), code_hi = mat_code(f, creator_first ? 2 : 1);
        // trial block sequence of the span, all entries at once: the block in front of old operator i (p_lo <= i <= p_hi;
        // i = p_hi: in front of the new later operator) is the old one shifted by the new earlier operator's quantum
        // numbers; the sequence is allowed iff every link is a connection (tsst[j] = block in front of trial entry j)
        bool bad = false;
        if constexpr (XM) if (outer) {
            // the blocks between the two operators stay; from the later one over beta to the earlier one the sequence is walked
            // anew and names the new outer block (check_sst_sequence over beta, Trace.F90:1278-1295; CTQMC.F90:3422-3449)
            int s = uni(x.osst[p_hi]);
            s = uni(x.t_conn[s * nf2 + code_hi]);
            for (int i = p_hi; i < x.nops && s >= 0; ++i) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
            outer_new = s;
            for (int i = 0; i < p_lo && s >= 0; ++i) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]);
            if (s >= 0) s = uni(x.t_conn[s * nf2 + code_lo]);
            bad = !(s >= 0 && s == uni(x.osst[p_lo]));
        }
        for (int base = p_lo; base <= p_hi && !(XM && outer); base += 64) {
            const int i = base + x.lane;
            if (i <= p_hi) {
                const int ci = x.t_shift[x.osst[i] * nf2 + code_lo];
                x.tsst[i + 1] = ci;
                bool ok = ci >= 0;
                if (ok && i == p_lo) ok = x.t_conn[x.osst[p_lo] * nf2 + code_lo] == ci;
                if (ok) {
                    const int nxt = i < p_hi ? x.t_shift[x.osst[i + 1] * nf2 + code_lo] : x.osst[p_hi];
                    ok = x.t_conn[ci * nf2 + (i < p_hi ? x.om[i] : code_hi)] == nxt;
                }
                bad |= !ok;
            }
        }
        valid = !eq && !__ballot(bad);
    } else {
    int c_lo = 0, a_lo = 0; bool blocked = (tau1 == tau2);
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        c_lo += count_pred(tce < lo); a_lo += count_pred(tae < lo);
        blocked |= __ballot((tce >= lo && tce <= hi) || (tae >= lo && tae <= hi)) != 0ull;
    }
    if constexpr (WORM) {
        if (x.sector != 1) {
#pragma unroll
            for (int i = 0; i < WORM; ++i) {
                if (x.wfl[i] != f) continue;
                blocked |= (x.wtau[i] >= lo && x.wtau[i] <= hi);
                if (x.wtau[i] < lo) { if (i & 1) c_lo += 1; else a_lo += 1; }
            }
        }
    }
    int occ_lo = ((x.outer >> flav_bit(f, C.nd)) & 1) + c_lo - a_lo;
    valid = !blocked && occ_lo == (creator_first ? 0 : 1);
    if constexpr (XM) if (outer) {
        // one-state blocks: the occupation of the flavour changes everywhere BUT between the two operators, which is allowed iff
        // no operator of the flavour lies outside them and the earlier one fits the new occupation at tau = 0
        int outside = 0; bool eq = (tau1 == tau2);
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane;
            bool v = e < k;
            double tce = v ? tcp[e] : lo, tae = v ? tap[e] : lo;
            outside += count_pred(v && (tce < lo || tce > hi)) + count_pred(v && (tae < lo || tae > hi));
            eq |= __ballot(v && (tce == lo || tce == hi || tae == lo || tae == hi)) != 0ull;
        }
        const int n0 = (x.outer >> flav_bit(f, C.nd)) & 1;
        valid = !eq && outside == 0 && creator_first == (n0 == 1);
        outer_new = x.outer ^ (1 << flav_bit(f, C.nd));
    }
    }
    DBG_STAMP(x, C_DBG4);
    if (!valid) { x.count(C_QNADD); return; }
    if (k >= kmax) { x.err |= ERR_KMAX; x.count(C_QNADD); return; }

    // determinant ratio S = F(tau_a-tau_c) - R^T M Q with Q_j = F(tau_a_j - tau_c_new), R_i = F(tau_a_new - tau_c_i)
    // (Trace.F90:2806-2825, MatrixUpdate.F90:392-407).  Flat element loop: all loads of M are independent, one
    // memory round trip for k^2 <= 64 * 4.
    double *Mf = x.mat(f);
    const double *ft = C.ftab + (size_t)f * C.nftau;
    const int kk = k * k;
    const unsigned magic = k > 0 ? (1u << 24) / (unsigned)k + 1u : 0u;      // e / k == (e * magic) >> 24 for e < 65536
    double mv0[4]; int iv0[4], jv0[4];                   // first 256 elements of M: loads in flight during the F lookups
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int e = u * 64 + x.lane;
        int ec = e < kk ? e : (kk > 0 ? kk - 1 : 0);
        int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
        iv0[u] = i; jv0[u] = j;
        mv0[u] = (kk > 0) ? Mf[(size_t)i * kmax + j] : 0.0;
    }
    for (int e = x.lane; e < k; e += 64) {
        x.Q[e] = hybr(ft, C.nftau, C.beta, tap[e] - tau2);
        x.R[e] = hybr(ft, C.nftau, C.beta, tau1 - tcp[e]);
    }
    double S0 = hybr(ft, C.nftau, C.beta, tau1 - tau2);
    wave_sync();
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int e = u * 64 + x.lane;
        if (e < kk) acc += x.R[iv0[u]] * mv0[u] * x.Q[jv0[u]];
    }
    for (int e0 = 256; e0 < kk; e0 += 256) {
        double mv[4]; int iv[4], jv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            int ec = e < kk ? e : kk - 1;
            int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
            iv[u] = i; jv[u] = j;
            mv[u] = Mf[(size_t)i * kmax + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            double t = x.R[iv[u]] * mv[u] * x.Q[jv[u]];
            acc += (e < kk) ? t : 0.0;
        }
    }
    double S = S0 - wave_sum(acc);
    DBG_STAMP(x, C_DBG5);

    double dlogtr, tl_new = 0.0, ts_new = 1.0;
    if constexpr (MAT) {
        if (x.nops + 2 > C.wcap) { x.err |= ERR_WCAP; return; }
        const int code_lo = mat_code(f, creator_first ? 1 : 2), code_hi = mat_code(f, creator_first ? 2 : 1);
        for (int j = x.lane; j <= x.nops + 2; j += 64) {           // blocks outside the span are unchanged
            if (j <= p_lo) x.tsst[j] = x.osst[j];
            else if (j >= p_hi + 2) x.tsst[j] = x.osst[j - 2];
        }
        mat_insert(x, p_lo, lo, code_lo); mat_insert(x, p_hi + 1, hi, code_hi);
        x.k_first = p_lo;
        bool plain = true;
        if constexpr (XM) if (outer) {
            plain = false;
            mat_walk_into(x, outer_new, x.tsst);
            mat_trace_trial<MAT>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new);
        }
        if (plain) mat_trace_trial<MAT, (MAT > 4 && WORM == 0)>(x, x.tsst, x.nops, -1, -1, tl_new, ts_new);
        dlogtr = tl_new - x.tr_log;
    } else {
        double I = level_integral<WORM>(x, f, lo, hi);
        dlogtr = creator_first ? -I : I;
        if constexpr (XM) if (outer) {
            // occupied (n0 = 0) or emptied (n0 = 1) on [0, lo) and (hi, beta]
            const double Ic = level_integral<WORM>(x, f, 0.0, C.beta) - I;
            dlogtr = ((x.outer >> flav_bit(f, C.nd)) & 1) ? Ic : -Ic;
        }
    }
    double rnd = x.draw();

    if (x.npairs < 0) x.npairs = build_window(x);
    int dnp = 0;
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 0.0, tae = v ? tap[e] : 0.0;
        dnp += count_pred(v && pair_ok(tau2, tae, x.wmin, x.wmax, C.taudiff_max));
        dnp += count_pred(v && pair_ok(tce, tau1, x.wmin, x.wmax, C.taudiff_max));
    }
    dnp += pair_ok(tau2, tau1, x.wmin, x.wmax, C.taudiff_max) ? 1 : 0;
    int np_new = x.npairs + dnp;

    double prob = fabs(tdf * (double)(2 * C.nd) / (double)np_new * fabs(S) * exp(dlogtr));
    if constexpr (MAT) { if (ts_new == 0.0) prob = 0.0; }
    DBG_STAMP(x, C_DBG6);
    if (!(rnd < prob)) {
        if constexpr (MAT) { mat_remove(x, p_hi + 1); mat_remove(x, p_lo); }
        return;
    }

    ACC_START(x);
    // time-ordered ranks of the new operators for the cofactor sign (-1)^|FPos1-FPos2| (Trace.F90:2828)
    int pc = 0, pa = 0;
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        pc += count_pred(tce < tau2); pa += count_pred(tae < tau1);
    }
    const double sig = ((pc - pa) & 1) ? -1.0 : 1.0;
    // accepted: M' = [[M + MQ RM^T / S, -MQ/S], [-RM^T/S, 1/S]] with the new row/column appended at slot k.
    // MQ_i = sum_j M_ij Q_j and RM_j = sum_i R_i M_ij come from one coalesced pass over the flattened block (element e in
    // lane e mod 64): every element adds its two products to the LDS vectors with ds_add_f64, instead of every lane
    // walking a row and a column of M (strided loads, three dependent round trips at k = 17)
    for (int e = x.lane; e < k; e += 64) { x.MQ[e] = 0.0; x.RM[e] = 0.0; }
    lds_sync();
    for (int e0 = 0; e0 < kk; e0 += 256) {
        double mv[4]; int iv[4], jv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            int ec = e < kk ? e : kk - 1;
            int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
            iv[u] = i; jv[u] = j;
            mv[u] = Mf[(size_t)i * kmax + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            if (e < kk) { atomicAdd(&x.MQ[iv[u]], mv[u] * x.Q[jv[u]]); atomicAdd(&x.RM[jv[u]], x.R[iv[u]] * mv[u]); }
        }
    }
    lds_sync();
    ACC_STAMP(x, C_DBG0);
    double invS = 1.0 / S;
    for (int e0 = 0; e0 < kk; e0 += 256) {
        double mv[4]; int iv[4], jv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            int ec = e < kk ? e : kk - 1;
            int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
            iv[u] = i; jv[u] = j;
            mv[u] = Mf[(size_t)i * kmax + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            if (e < kk) Mf[(size_t)iv[u] * kmax + jv[u]] = mv[u] + S * (-x.MQ[iv[u]] * invS) * (-x.RM[jv[u]] * invS);
        }
    }
    for (int e = x.lane; e < k; e += 64) {
        Mf[(size_t)e * kmax + k] = -x.MQ[e] * invS;
        Mf[(size_t)k * kmax + e] = -x.RM[e] * invS;
    }
    if (x.lane == 0) { Mf[(size_t)k * kmax + k] = invS; tcp[k] = tau2; tap[k] = tau1; }
    x.set_order(f, k + 1);
    if (k + 1 >= kmax - (XM ? 1 : 0)) x.err |= ERR_KMAX;   // flavour full (XM: one slot left): the sweep loop stops before the next step (capacity is enlarged by the host)
    wave_sync();
    ACC_STAMP(x, C_DBG1);
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = np_new;
    ACC_STAMP(x, C_DBG2);
    window_list_add(x, f, k, tau2, tau1);
    ACC_STAMP(x, C_DBG3);
    if constexpr (MAT) { x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); }
    if constexpr (XM) if (outer) { x.outer = outer_new; x.pvalid = 0; x.kval = 0; }
    DBG_STAMP(x, C_DBG7);
    if (WORM && x.sector != 1) x.count(C_W_ACCADD); else x.count(C_ACCADD);
    accpair_add<CALIB>(x, f, hi - lo);
}

// ---------------------------------------------------------------------------------------------------------------
// pair removal: CTQMC.F90:3892-4150 (StepRem), Trace.F90:4851-4950, 2937-2963, MatrixUpdate.F90:445-482
// ---------------------------------------------------------------------------------------------------------------
template <int MAT, int WORM = 0, bool CALIB = false, bool XM = false>
__device__ W2DYN_INLINE void step_rem(Ctx &x, bool outer = false)
{
    const DevConst &C = x.C;
    if (WORM && x.sector != 1) x.count(C_W_TRYREM); else x.count(C_TRYREM);
    DBG_START(x);
    // window list, partner counts and prefix counts are kept until the configuration or the window changes
    if (!x.wvalid || x.npairs < 0) x.npairs = build_window(x);
    const int np = x.npairs;
    DBG_STAMP(x, C_DBG0);
    if (np <= 0) { x.count(C_QNREM); return; }
    int pick = x.randint(1, np);
    const int nw = x.nwin;
#ifdef W2DYN_STAMPS_PICK
    if (x.lane == 0) x.cnt[C_DBG4] += 1ull;
    if (x.lane == 0 && nw > 64) x.cnt[C_DBG5] += 1ull;
    if (x.lane == 0) x.cnt[C_DBG6] += (unsigned long long)nw;
    unsigned long long tp0_ = __builtin_amdgcn_s_memtime();
#endif

    // first operator of the pick-th pair in the reference's enumeration (time order of the earlier operator),
    // then its rnk-th partner in time order
    double t1, t2; int m1, m2;
    int e1 = -1, e2 = -1;                                // list entries of the pair (register-resident list only)
    if (nw <= 64) {
        const double te = x.wr_t; const int me = x.wr_m, ce = x.wr_c, P = x.wr_p;
        const bool live = x.lane < nw;
        unsigned long long m = __ballot(live && (P < pick) && (pick <= P + ce));
        int estar = __ffsll((long long)m) - 1;
        int rnk = pick - __builtin_amdgcn_readlane(P, estar);
        t1 = readlane_d(te, estar); m1 = __builtin_amdgcn_readlane(me, estar);
        const int want = (m1 & 0xffff) ^ 0x300;
        const double thi = t1 + C.taudiff_max;
        bool partner = live && ((me & 0xffff) == want) && (te > t1) && !(te > thi);
        unsigned long long pm = __ballot(partner);
        int r = 0;
        while (pm) {                                     // rank of each partner among the partners, by time
            int p = __ffsll((long long)pm) - 1; pm &= pm - 1ull;
            double tp = readlane_d(te, p);
            r += (tp < te) ? 1 : 0;
        }
        unsigned long long hm = __ballot(partner && r == rnk - 1);
        int qstar = __ffsll((long long)hm) - 1;
        t2 = readlane_d(te, qstar); m2 = __builtin_amdgcn_readlane(me, qstar);
        e1 = estar; e2 = qstar;
#ifdef W2DYN_STAMPS_PICK
        if (x.lane == 0) x.cnt[C_DBG7] += __builtin_amdgcn_s_memtime() - tp0_;
#endif
    } else {
        int estar = -1, rnk = 0;
        for (int base = 0; base < nw; base += 64) {
            int e = base + x.lane;
            int P = e < nw ? x.wp[e] : 0;
            bool hit = e < nw && (P < pick) && (pick <= P + x.wc[e]);
            unsigned long long m = __ballot(hit);
            if (m) { int src = __ffsll((long long)m) - 1; estar = base + src; rnk = pick - __shfl(P, src, 64); }
        }
        estar = uni(estar); rnk = uni(rnk);
        t1 = x.wt[estar]; m1 = x.wm[estar];
        const int want = (m1 & 0xffff) ^ 0x300;
        const double thi = t1 + C.taudiff_max;
        // the partner of rank rnk-1 in time: for each partner (a few), count the partners at earlier times
        int qstar = -1;
        for (int b1 = 0; b1 < nw && qstar < 0; b1 += 64) {
            int q = b1 + x.lane;
            double tq = q < nw ? x.wt[q] : -1.0;
            unsigned long long pm = __ballot(q < nw && ((x.wm[q < nw ? q : 0] & 0xffff) == want) && (tq > t1) && !(tq > thi));
            while (pm && qstar < 0) {
                const int src = __ffsll((long long)pm) - 1; pm &= pm - 1ull;
                const double tp = x.wt[b1 + src];
                int less = 0;
                for (int b2 = 0; b2 < nw; b2 += 64) {
                    int q2 = b2 + x.lane;
                    double t2_ = q2 < nw ? x.wt[q2] : -1.0;
                    less += __popcll(__ballot(q2 < nw && ((x.wm[q2 < nw ? q2 : 0] & 0xffff) == want) && (t2_ > t1) && !(t2_ > thi) && t2_ < tp));
                }
                if (less == rnk - 1) qstar = b1 + src;
            }
        }
        qstar = uni(qstar);
        t2 = x.wt[qstar]; m2 = x.wm[qstar];
    }
    DBG_STAMP(x, C_DBG1);
    int f = m1 & 0xff;
    int ca1 = (m1 >> 8) & 0xff;
    int ic = (ca1 == 1) ? (m1 >> 16) : (m2 >> 16);
    int ia = (ca1 == 2) ? (m1 >> 16) : (m2 >> 16);
    double tau_c = (ca1 == 1) ? t1 : t2, tau_a = (ca1 == 2) ? t1 : t2;
    double hybf = (double)np / ((x.wmax - x.wmin) *
                                (fmin(tau_a + C.taudiff_max, x.wmax) - fmax(tau_a - C.taudiff_max, x.wmin)));

    int k = x.order(f);
    const int kmax = C.kmax;
    double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
    double Sv = Mf[(size_t)ic * kmax + ia];            // issued early, consumed after the trace integral
    // quantum numbers: the pair may go iff no operator of its flavour lies strictly between (both are in the window,
    // so anything between them is in the window list too)
    bool blocked;
    int i1 = 0, i2 = 0;                                // MAT: list indices of the two operators
    int outer_new = x.outer;                           // XM: outer block / outer Fock state after a move that changes it
    if constexpr (MAT) {
        i1 = mat_lower(x, t1); i2 = mat_lower(x, t2);
        const int nf2 = 2 * C.nflav;
        // trial entry j is list entry j (j < i1), j+1 (i1 <= j < i2-1) or j+2; tsst[j] = block in front of it.  Between
        // the two operators every block is the old one shifted back by the earlier operator's quantum numbers; the pair
        // may go iff every link of that sequence is a connection (all entries checked at once)
        const int inv1 = uni(x.om[i1]) ^ 1;
        bool bad = false;
        if constexpr (XM) if (outer) {
            // the blocks between the two operators stay; the rest of the sequence, from behind the later one over beta to the
            // earlier one, is walked anew (CTQMC.F90:3935-3962) and written to the trial sequence as it goes
            int s = uni(x.osst[i2]);
            for (int i = i2 + 1; i < x.nops; ++i) { if (x.lane == 0) x.tsst[i - 2] = s; if (s >= 0) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]); }
            outer_new = s;
            if (x.lane == 0) x.tsst[x.nops - 2] = s;
            for (int i = 0; i < i1; ++i) { if (x.lane == 0) x.tsst[i] = s; if (s >= 0) s = uni(x.t_conn[s * nf2 + uni(x.om[i])]); }
            bad = !(s >= 0 && s == uni(x.osst[i1 + 1]));
            for (int j = i1 + x.lane; j < i2 - 1; j += 64) x.tsst[j] = x.osst[j + 1];
        }
        for (int base = i1 + 1; base <= i2 && !(XM && outer); base += 64) {
            const int i = base + x.lane;
            if (i <= i2) {
                const int ci = x.t_shift[x.osst[i] * nf2 + inv1];        // block in front of old operator i in the trial sequence
                if (i < i2) x.tsst[i - 1] = ci; 
                bool ok = ci >= 0;
                if (ok && i < i2) ok = x.t_conn[ci * nf2 + x.om[i]] == x.t_shift[x.osst[i + 1] * nf2 + inv1];
                if (ok && i == i1 + 1) ok = ci == x.osst[i1];
                if (ok && i == i2) ok = ci == x.osst[i2 + 1];
                bad |= !ok;
            }
        }
        blocked = __ballot(bad) != 0ull;
    } else if (nw <= 64) {
        blocked = __ballot(x.lane < nw && (x.wr_m & 0xff) == f && x.wr_t > t1 && x.wr_t < t2) != 0ull;
    } else {
        int between = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane;
            bool v = e < k;
            double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
            between += count_pred(tce > t1 && tce < t2) + count_pred(tae > t1 && tae < t2);
        }
        blocked = between != 0;
    }
    if constexpr (XM && MAT == 0) if (outer) {
        // one-state blocks: the occupation changes everywhere but between the two operators: none of the flavour may lie outside
        int outside = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane;
            bool v = e < k;
            double tce = v ? tcp[e] : t1, tae = v ? tap[e] : t1;
            outside += count_pred(v && (tce < t1 || tce > t2)) + count_pred(v && (tae < t1 || tae > t2));
        }
        blocked = outside != 0;
        outer_new = x.outer ^ (1 << flav_bit(f, C.nd));
    }
    if constexpr (WORM != 0 && MAT == 0) {        // one-state blocks: a worm operator of the pair's flavour between its operators blocks the removal
        if (x.sector != 1) {                      // (with larger blocks the list walk above has seen the worm operators already)
#pragma unroll
            for (int i = 0; i < WORM; ++i) blocked |= (x.wfl[i] == f && x.wtau[i] > t1 && x.wtau[i] < t2);
        }
    }
    DBG_STAMP(x, C_DBG2);
    if (blocked) { x.count(C_QNREM); return; }

    bool creator_first = tau_c < tau_a;
    double dlogtr, tl_new = 0.0, ts_new = 1.0;
    if constexpr (MAT) {
        bool plain = true;
        if constexpr (XM) if (outer) {
            plain = false;
            wave_sync();
            mat_trace_trial<MAT>(x, x.tsst, x.nops - 2, i1, i2, tl_new, ts_new);
        }
        if (plain) {
        for (int j = x.lane; j <= x.nops - 2; j += 64) {
            if (j < i1) x.tsst[j] = x.osst[j];
            else if (j >= i2 - 1) x.tsst[j] = x.osst[j + 2];
        }
        wave_sync();
        x.k_first = i1;
        mat_trace_trial<MAT, (MAT > 4 && WORM == 0)>(x, x.tsst, x.nops - 2, i1, i2, tl_new, ts_new);
        }
        dlogtr = tl_new - x.tr_log;
    } else {
        // the pair's own operators sit on the boundary of (t1,t2) and belong to flavour f, which level_integral skips
        double I = level_integral<WORM>(x, f, t1, t2);
        dlogtr = creator_first ? I : -I;
        if constexpr (XM) if (outer) {
            const double Ic = level_integral<WORM>(x, f, 0.0, C.beta) - I;
            dlogtr = ((x.outer >> flav_bit(f, C.nd)) & 1) ? Ic : -Ic;
        }
    }
    double S = uni(Sv);
    double prob = fabs(hybf / (double)(2 * C.nd) * fabs(S) * exp(dlogtr));
    if constexpr (MAT) { if (ts_new == 0.0) prob = 0.0; }
    double rnd = x.draw();
    DBG_STAMP(x, C_DBG3);
    if (!(rnd < prob)) return;
    if (fabs(S) < 1e-12) { x.err |= ERR_DETRAT; return; }   // CTQMC.F90:4062 stops here

    ACC_START(x);
    // NPairs after the removal: every pair that contains one of the two operators disappears; time-ordered ranks
    // of the two operators for the cofactor sign (Trace.F90:2958-2959)
    int dnp = 0, pc = 0, pa = 0;
    for (int base = 0; base < k; base += 64) {
        int e = base + x.lane;
        bool v = e < k;
        double tce = v ? tcp[e] : 2.0 * C.beta, tae = v ? tap[e] : 2.0 * C.beta;
        pc += count_pred(tce < tau_c); pa += count_pred(tae < tau_a);
        dnp += count_pred(v && e != ia && pair_ok(tau_c, tae, x.wmin, x.wmax, C.taudiff_max));
        dnp += count_pred(v && e != ic && pair_ok(tce, tau_a, x.wmin, x.wmax, C.taudiff_max));
    }
    dnp += 1;   // the removed pair itself (it was picked from the removable ones)
    const double sig = ((pc - pa) & 1) ? -1.0 : 1.0;

    ACC_STAMP(x, C_DBG4);
    // M'_ij = M_ij - M_i,ia M_ic,j / M_ic,ia for i != ic, j != ia; the last row/column move into the freed slots.  A lane
    // writes either to the place it read from or into row ic / column ia, which are sources of nothing but the two
    // vectors read first, so the update is done in place and no wait separates the loads of a chunk from its stores.
    const int last = k - 1;
    const int kk = k * k;
    const unsigned magic = (1u << 24) / (unsigned)k + 1u;
    const double invS = 1.0 / S;
    for (int e = x.lane; e < k; e += 64) { x.RM[e] = Mf[(size_t)ic * kmax + e]; x.MQ[e] = Mf[(size_t)e * kmax + ia] * invS; }
    wave_sync();
    ACC_STAMP(x, C_DBG5);
    for (int e0 = 0; e0 < kk; e0 += 256) {
        double mv[4]; int iv[4], jv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            int ec = e < kk ? e : kk - 1;
            int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
            iv[u] = i; jv[u] = j;
            mv[u] = Mf[(size_t)i * kmax + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + u * 64 + x.lane;
            int i = iv[u], j = jv[u];
            if (e < kk && i != ic && j != ia) {
                int id = (i == last) ? ic : i, jd = (j == last) ? ia : j;
                Mf[(size_t)id * kmax + jd] = mv[u] - x.MQ[i] * x.RM[j];
            }
        }
    }
    if (x.lane == 0) { tcp[ic] = tcp[last]; tap[ia] = tap[last]; }
    x.set_order(f, last);
    wave_sync();
    ACC_STAMP(x, C_DBG6);
    x.det_sign *= (S < 0.0 ? -1.0 : 1.0) * sig;
    x.det_log += log(fabs(S));
    x.npairs = np - dnp;
    window_list_rem(x, f, e1, e2, ic, ia, last, tau_c, tau_a);
    ACC_STAMP(x, C_DBG7);
    if constexpr (MAT) { mat_remove(x, i2); mat_remove(x, i1); x.tr_log = tl_new; x.tr_sign = ts_new; mat_commit_sst(x); }
    if constexpr (XM) if (outer) { x.outer = outer_new; x.pvalid = 0; x.kval = 0; }
    if (WORM && x.sector != 1) x.count(C_W_ACCREM); else x.count(C_ACCREM);
    accpair_add<CALIB>(x, f, t2 - t1);
}

// ---------------------------------------------------------------------------------------------------------------
// tau-shift: CTQMC.F90:4557-4835, Trace.F90:2355-2457; empty configuration: StepChangeOuter 4570-4638
// ---------------------------------------------------------------------------------------------------------------
__device__ W2DYN_INLINE void step_change_outer(Ctx &x)
{
    const DevConst &C = x.C;
    double r = x.draw();
    int sel = C.nouter - 1;
    bool found = false;
    for (int base = 0; base < C.nouter && !found; base += 64) {
        int s = base + x.lane;
        bool hit = s < C.nouter && C.outer_cum[s] >= r && C.outer_w[s] != 0.0;
        unsigned long long m = __ballot(hit);
        if (m) { sel = base + __ffsll((long long)m) - 1; found = true; }
    }
    x.outer = uni(sel);
}

template <int MAT>
__device__ W2DYN_INLINE void step_shift(Ctx &x)
{
    const DevConst &C = x.C;
    int nops = 0;
    for (int f = 0; f < C.nflav; ++f) nops += 2 * x.order(f);
    if (nops == 0) {
        step_change_outer(x);
        x.pvalid = 0;
        if constexpr (MAT) { mat_walk_sst(x); mat_trace_current<MAT>(x); }
        return;
    }
    double dt = x.draw() * C.beta;
    if (dt == 0.0 || dt == C.beta) return;
    x.count(C_TRYSHIFT);
    const int kmax = C.kmax;
    int outer = x.outer;
    if constexpr (MAT) {       // block in front of the first operator that does not wrap (CTQMC.F90:4717)
        int first = mat_lower(x, dt);
        if (first < x.nops) outer = uni(x.osst[first]);
    }
    double detrat = 1.0;
    for (int f = 0; f < C.nflav; ++f) {
        int k = x.order(f);
        if (k == 0) continue;
        double *tcp = x.tc(f), *tap = x.ta(f), *Mf = x.mat(f);
        int wcnt = 0, wacnt = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane;
            bool v = e < k;
            wcnt += count_pred(v && tcp[e] < dt); wacnt += count_pred(v && tap[e] < dt);
        }
        if constexpr (!MAT) { if ((wcnt + wacnt) & 1) outer ^= 1 << flav_bit(f, C.nd); }
        if (!((wcnt == 0 && wacnt == 0) || (wcnt == k && wacnt == k))) {
            // rows of creators and columns of annihilators that wrap change sign: flat element loop, four loads of M in
            // flight per lane (M lives in HBM: a row-by-row loop pays one memory round trip per row)
            const int kk = k * k;
            const unsigned magic = (1u << 24) / (unsigned)k + 1u;      // e / k == (e * magic) >> 24 for e < 65536
            for (int e0 = 0; e0 < kk; e0 += 256) {
                double mv[4]; size_t at[4]; bool flip[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 64 + x.lane, ec = e < kk ? e : kk - 1;
                    const int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
                    at[u] = (size_t)i * kmax + j;
                    flip[u] = e < kk && ((tcp[i] < dt) != (tap[j] < dt));
                    mv[u] = Mf[at[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (flip[u]) Mf[at[u]] = -mv[u];
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
    x.npairs = -1; x.wvalid = 0;
    if constexpr (MAT) { mat_build_list(x); mat_walk_sst(x); mat_trace_current<MAT>(x); x.pvalid = 0; }
    x.count(C_ACCSHIFT);
}

// ---------------------------------------------------------------------------------------------------------------
// global moves: CTQMC.F90:4433-4554, Trace.F90:1083-1198, 1306-1433, 2126-2188, 2279-2333
// ---------------------------------------------------------------------------------------------------------------

// Build F(tau_a_r - tau_c_c) of one flavour (row = annihilator slot, column = creator slot) and invert it by
// Gauss-Jordan with partial pivoting; the result read as [creator][annihilator] is M and is stored to `A` (global,
// row stride kmax).  Returns log|det F| and its sign in slot order.  What the reference obtains from dgetrf/dgetri
// (Trace.F90:2279-2333, MatrixUpdate.F90:41-99).
// k <= kgj (<= 32): the elimination runs in the wave's LDS tile (compact stride k) with IMPLICIT row pivoting: rows
// never move, lane r owns row r in the pivot search and remembers the step at which it was eliminated, the row
// permutation sigma (sigma(p) = row eliminated at step p) is applied once when the result is written out,
// A^-1[r][c] = W[sigma(r)][sigma^-1(c)].  Pivots, multipliers and updates are those of the row-swapping elimination,
// so the result is the same to the last bit; what is saved is a row swap per step, the k column swaps at the end and
// one log per pivot (taken by all lanes at once afterwards).  Larger matrices are eliminated in place in global
// memory with explicit swaps.
// OD: off-diagonal hybridisation - the slots carry their flavours (fcp, fap) and F comes from the two-flavour table
template <bool OD>
__device__ W2DYN_INLINE void build_and_invert(Ctx &x, int f, int k, const double *tcp, const double *tap, double *A,
                                 double &logdet, double &sgn, const int *fcp, const int *fap)
{
    const DevConst &C = x.C;
    const int kmax = C.kmax;
    const double *ft = C.ftab + (size_t)f * C.nftau;
    logdet = 0.0; sgn = 1.0;
    if (k == 0) return;
    const bool in_lds = k <= C.kgj;
    const int kk = k * k;
    const unsigned magic = (1u << 24) / (unsigned)k + 1u;
    for (int e = x.lane; e < kk; e += 64) {
        int r = (int)(((unsigned)e * magic) >> 24), c = e - r * k;
        double v;
        if constexpr (OD) v = hybr(C.ffull + ((size_t)fap[r] * C.nflav + fcp[c]) * C.nftau, C.nftau, C.beta, tap[r] - tcp[c]);
        else v = hybr(ft, C.nftau, C.beta, tap[r] - tcp[c]);
        if (in_lds) x.gj[e] = v; else A[(size_t)r * kmax + c] = v;
    }
    wave_sync();
    int *piv = x.piv;                // kmax entries of LDS
    if (in_lds) {
        const bool lv = x.lane < k;
        const int rpt = 64 / k, gr = (int)(((unsigned)x.lane * magic) >> 24), gc = x.lane - gr * k;   // rows per group; the lane's row in it, its column
        const bool gact = gr < rpt;
        int pstep = -1, sig = 0;     // lane r: step at which row r was eliminated; lane p: sigma(p)
        double dreg = 1.0;           // lane p: pivot of step p
        for (int p = 0; p < k; ++p) {
            const double a = lv ? x.gj[x.lane * k + p] : 0.0;          // column p, lane = row
            const double v = (lv && pstep < 0) ? fabs(a) : -1.0;
            const double mx = wave_max(v);
            if (!(mx > 0.0)) { x.err |= ERR_SINGULAR; return; }
            const int bi = __ffsll((long long)__ballot(v == mx)) - 1;
            const double rowv = lv ? x.gj[bi * k + x.lane] : 0.0;      // pivot row, lane = column
            const double d = readlane_d(a, bi);
            const double id = 1.0 / d;
            if (x.lane == bi) pstep = p;
            if (x.lane == p) { dreg = d; sig = bi; }
            if (lv) { x.RM[x.lane] = (x.lane == p) ? id : rowv * id; x.MQ[x.lane] = a; }   // scaled pivot row, pivot column
            wave_sync();
            // lane = (row inside a group of 64 / k rows, column): the column, its pivot-row element and the test against p are
            // the same for every group of rows, only the row moves on
            {
                const double rp = gact ? x.RM[gc] : 0.0;
                const bool cp = gc == p;
                for (int r = gr; r < k; r += rpt) {
                    if (gact) {
                        const int e = r * k + gc;
                        const double g = x.gj[e];
                        x.gj[e] = (r == bi) ? rp : ((cp ? 0.0 : g) - x.MQ[r] * rp);
                    }
                }
            }
            wave_sync();
        }
        logdet = wave_sum(lv ? log(fabs(dreg)) : 0.0);
        int odd = __popcll(__ballot(lv && dreg < 0.0));
        int inv = 0;                 // inversions of sigma^-1 (same parity as sigma = parity of the row interchanges)
        for (int i = 0; i < k; ++i) { int pi = __builtin_amdgcn_readlane(pstep, i); inv += (lv && i < x.lane && pi > pstep) ? 1 : 0; }
        odd += wave_sum(inv);
        if (odd & 1) sgn = -1.0;
        if (lv) piv[x.lane] = sig | (pstep << 16);
        wave_sync();
        for (int e = x.lane; e < kk; e += 64) {
            int r = (int)(((unsigned)e * magic) >> 24), c = e - r * k;
            A[(size_t)r * kmax + c] = x.gj[(piv[r] & 0xffff) * k + (piv[c] >> 16)];
        }
        wave_sync();
        return;
    }
    const int gs = kmax;
    for (int p = 0; p < k; ++p) {
        double best = -1.0; int brow = p;
        for (int r = p + x.lane; r < k; r += 64) {
            double v = fabs(A[(size_t)r * gs + p]);
            if (v > best) { best = v; brow = r; }
        }
        double mx = wave_max(best);
        if (!(mx > 0.0)) { x.err |= ERR_SINGULAR; return; }
        unsigned long long m = __ballot(best == mx);
        int bi = __builtin_amdgcn_readlane(brow, __ffsll((long long)m) - 1);
        if (x.lane == 0) piv[p] = bi;
        if (bi != p) {
            for (int j = x.lane; j < k; j += 64) { double t = A[(size_t)p * gs + j]; A[(size_t)p * gs + j] = A[(size_t)bi * gs + j]; A[(size_t)bi * gs + j] = t; }
            sgn = -sgn;
            wave_sync();
        }
        double d = uni(A[(size_t)p * gs + p]);
        logdet += log(fabs(d)); if (d < 0.0) sgn = -sgn;
        double id = 1.0 / d;
        for (int j = x.lane; j < k; j += 64) {
            double rv = A[(size_t)p * gs + j];
            double cv = A[(size_t)j * gs + p];
            x.RM[j] = (j == p) ? id : rv * id;            // scaled pivot row
            x.MQ[j] = cv;                                  // pivot column
        }
        wave_sync();
        for (int e0 = 0; e0 < kk; e0 += 256) {
            double mv[4]; int rv[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = e0 + u * 64 + x.lane; int ec = e < kk ? e : kk - 1;
                int r = (int)(((unsigned)ec * magic) >> 24), c = ec - r * k;
                rv[u] = r; cv[u] = c; mv[u] = A[(size_t)r * kmax + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = e0 + u * 64 + x.lane;
                if (e < kk) {
                    double rp = x.RM[cv[u]];
                    A[(size_t)rv[u] * kmax + cv[u]] = (rv[u] == p) ? rp : (((cv[u] == p) ? 0.0 : mv[u]) - x.MQ[rv[u]] * rp);
                }
            }
        }
        wave_sync();
    }
    for (int p = k - 1; p >= 0; --p) {    // undo the row interchanges as column interchanges
        int bi = uni(piv[p]);
        if (bi != p) {
            for (int r = x.lane; r < k; r += 64) { double t = A[(size_t)r * gs + p]; A[(size_t)r * gs + p] = A[(size_t)r * gs + bi]; A[(size_t)r * gs + bi] = t; }
            wave_sync();
        }
    }
}

// parity (+1/-1) of the permutation that sorts `t[0..k)` (number of inversions)
__device__ W2DYN_INLINE double sort_parity(Ctx &x, const double *t, int k)
{
    int inv = 0;
    for (int j = x.lane; j < k; j += 64) { double tj = t[j]; for (int i = 0; i < j; ++i) inv += (t[i] > tj) ? 1 : 0; }
    return (wave_sum(inv) & 1) ? -1.0 : 1.0;
}

// Scalar local trace of the current configuration and of its relabelled image under a global move, both from one pass.
// log tr = -int_0^beta E(n(tau)) dtau, E(n) = sum_f eps_f n_f + sum_{f<g} U_fg n_f n_g, up to the constant Egs*beta, with
// n_f(tau) = n_f(0) + #creators before tau - #annihilators before tau:
//   log tr = -[ sum_f eps_f (n0_f beta + S_f) + sum_{f<g} U_fg (n0_f n0_g beta + n0_f S_g + n0_g S_f + D_fg) ]
//   S_f  = sum over pairs of f of (ta - tc)                        (occupied length beyond the tau=0 occupation)
//   D_fg = sum_{x in f, y in g} s_x s_y (beta - max(t_x, t_y))     (s = +1 creator, -1 annihilator)
// A flavour relabelling f -> gmap[f] only relabels S and D; exchanging creators and annihilators flips the sign of
// S and leaves D as it is, so D_fg is evaluated once (lane = operator y of g, loop over the operators x of f in LDS)
// and enters both traces with its own coupling.
__device__ W2DYN_INLINE void glob_logtraces(Ctx &x, const int *gmap, bool flip, int outer_new, double &ltr_old, double &ltr_new)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd;
    const double beta = C.beta;
    double sreg = 0.0;                               // lane f: S_f
    for (int f = 0; f < nf; ++f) {
        const int k = x.order(f);
        const double *cf = x.tc(f), *af = x.ta(f);
        double p = 0.0;
        for (int e = x.lane; e < k; e += 64) p += (beta - cf[e]) - (beta - af[e]);
        p = wave_sum(p);
        if (x.lane == f) sreg = p;
    }
    double uo = 0.0, un = 0.0, po = 0.0, pn = 0.0;   // wave-uniform parts, per-lane parts (old / new)
    for (int f = 0; f < nf; ++f) {
        const int gf = gmap[f];
        const double sf = readlane_d(sreg, f), sfn = flip ? -sf : sf;
        const double nf0 = (double)((x.outer >> flav_bit(f, nd)) & 1), nfn = (double)((outer_new >> flav_bit(gf, nd)) & 1);
        uo += C.eps[f] * (nf0 * beta + sf);
        un += C.eps[gf] * (nfn * beta + sfn);
        const int kf = x.order(f);
        const double *cf = x.tc(f), *af = x.ta(f);
        for (int g = f + 1; g < nf; ++g) {
            const int gg = gmap[g];
            const double u_old = C.udd[f * nf + g], u_new = C.udd[gf * nf + gg];
            if (u_old == 0.0 && u_new == 0.0) continue;
            const double sg = readlane_d(sreg, g), sgn_ = flip ? -sg : sg;
            const double ng0 = (double)((x.outer >> flav_bit(g, nd)) & 1), ngn = (double)((outer_new >> flav_bit(gg, nd)) & 1);
            uo += u_old * (nf0 * ng0 * beta + nf0 * sg + ng0 * sf);
            un += u_new * (nfn * ngn * beta + nfn * sgn_ + ngn * sfn);
            const int kg = x.order(g);
            if (kf == 0 || kg == 0) continue;
            const double *cg = x.tc(g), *ag = x.ta(g);
            for (int base = 0; base < 2 * kg; base += 64) {
                const int iy = base + x.lane;
                if (iy < 2 * kg) {
                    const double ty = iy < kg ? cg[iy] : ag[iy - kg];
                    double dc = 0.0, da = 0.0;
                    for (int ix = 0; ix < kf; ++ix) { dc += beta - fmax(cf[ix], ty); da += beta - fmax(af[ix], ty); }
                    const double t = iy < kg ? dc - da : da - dc;
                    po += u_old * t; pn += u_new * t;
                }
            }
        }
    }
    ltr_old = -(uo + wave_sum(po));
    ltr_new = -(un + wave_sum(pn));
}

// log of the scalar local trace of the current configuration as the reference carries it (Trace%Trace%log,
// Trace.F90:5595-5724: energies relative to the ground level): the `old` half of glob_logtraces plus Egs * beta
// pairs != nullptr: sum_{f<g} U_fg D_fg comes from the caller (k_meas_misc gets it out of its sign scan, which visits the
// same operator pairs: D_fg = sum_z s_z (beta - t_z) P(t_z), z over the operators of f and g, P = creators minus annihilators
// of the other flavour at earlier times)
__device__ W2DYN_INLINE double scalar_logtrace(Ctx &x, const double *pairs = nullptr)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd;
    const double beta = C.beta;
    double sreg = 0.0;                               // lane f: S_f
    for (int f = 0; f < nf; ++f) {
        const int k = x.order(f);
        const double *cf = x.tc(f), *af = x.ta(f);
        double p = 0.0;
        for (int e = x.lane; e < k; e += 64) p += (beta - cf[e]) - (beta - af[e]);
        p = wave_sum(p);
        if (x.lane == f) sreg = p;
    }
    double uo = 0.0, po = 0.0;
    for (int f = 0; f < nf; ++f) {
        const double sf = readlane_d(sreg, f);
        const double nf0 = (double)((x.outer >> flav_bit(f, nd)) & 1);
        uo += C.eps[f] * (nf0 * beta + sf);
        const int kf = x.order(f);
        const double *cf = x.tc(f), *af = x.ta(f);
        for (int g = f + 1; g < nf; ++g) {
            const double u = C.udd[f * nf + g];
            if (u == 0.0) continue;
            const double sg = readlane_d(sreg, g);
            const double ng0 = (double)((x.outer >> flav_bit(g, nd)) & 1);
            uo += u * (nf0 * ng0 * beta + nf0 * sg + ng0 * sf);
            const int kg = x.order(g);
            if (pairs || kf == 0 || kg == 0) continue;
            const double *cg = x.tc(g), *ag = x.ta(g);
            for (int base = 0; base < 2 * kg; base += 64) {
                const int iy = base + x.lane;
                if (iy < 2 * kg) {
                    const double ty = iy < kg ? cg[iy] : ag[iy - kg];
                    double dc = 0.0, da = 0.0;
                    for (int ix = 0; ix < kf; ++ix) { dc += beta - fmax(cf[ix], ty); da += beta - fmax(af[ix], ty); }
                    po += u * (iy < kg ? dc - da : da - dc);
                }
            }
        }
    }
    return C.egs * beta - (uo + (pairs ? *pairs : wave_sum(po)));
}

__device__ W2DYN_INLINE void step_glob(Ctx &x)
{
    const DevConst &C = x.C;
    const int nf = C.nflav, nd = C.nd, kmax = C.kmax;
    x.count(C_TRYGLOB);
    GLOB_START(x);
    int nops = 0;
    for (int f = 0; f < nf; ++f) nops += 2 * x.order(f);
    if (nops == 0) return;
    x.wvalid = 0;                    // the inversion below uses the window list's LDS as pivot storage
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
    const bool flip = gu[0] != 0;
    // relabelling: old flavour f becomes gmap[f]; new flavour g comes from fsrc[g].  The new configuration is read from
    // the old tau arrays in LDS (creators and annihilators exchanged when flip); a copy goes to scratch
    // (TS[g][0] = creators, TS[g][1] = annihilators of new flavour g) for the installation after acceptance.
    double *TS = C.TS + (size_t)x.w * nf * 2 * kmax;
    int gmap[MAXFLAV], fsrc[MAXFLAV];
    for (int f = 0; f < nf; ++f) {
        int num = (f >> 1) + nd * (f & 1) + 1;
        int t = gu[num];
        int g = 2 * ((t - 1) % nd) + (t - 1) / nd;
        int k = x.order(f);
        gmap[f] = g; fsrc[g] = f;
        const double *srcc = flip ? x.ta(f) : x.tc(f), *srca = flip ? x.tc(f) : x.ta(f);
        for (int e = x.lane; e < k; e += 64) { TS[(size_t)(2 * g) * kmax + e] = srcc[e]; TS[(size_t)(2 * g + 1) * kmax + e] = srca[e]; }
    }
    // quantum numbers: occupancy of flavour g at tau=0 is fixed by its first operator, free if it has none
    int fixed = 0, freemask = 0;
    for (int g = 0; g < nf; ++g) {
        const int f = fsrc[g];
        int k = x.order(f);
        if (k == 0) { freemask |= 1 << flav_bit(g, nd); continue; }
        const double *srcc = flip ? x.ta(f) : x.tc(f), *srca = flip ? x.tc(f) : x.ta(f);
        double mc = 1e300, ma = 1e300;
        for (int e = x.lane; e < k; e += 64) { mc = fmin(mc, srcc[e]); ma = fmin(ma, srca[e]); }
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
    GLOB_STAMP(x, C_DBG0);
    if (sst_factor == 0.0) { x.count(C_QNGLOB); return; }

    double ltr_old, ltr_new;
    glob_logtraces(x, gmap, flip, outer_new, ltr_old, ltr_new);
    GLOB_STAMP(x, C_DBG3);
    // new inverse matrices into the second buffer, determinant in the reference's time order.  A relabelling between flavours of identical
    // F(tau) that leaves creators creators (spin flip of a paramagnetic bath, symmetry moves) maps every hybridisation matrix onto one that
    // is already inverted - same slots, same determinant, same sign: the matrices move along with their flavours and nothing is inverted
    // (the reference inverts from scratch and gets the same numbers up to rounding).
    double *MBw = C.MB + (size_t)x.w * nf * kmax * kmax;
    double logdet_new = 0.0, sign_new = 1.0;
    bool carried = !flip;
    for (int g = 0; g < nf; ++g) { const int f = fsrc[g]; if (x.order(f) > 0 && !((C.fsame[f] >> g) & 1)) carried = false; }
    if (carried) {
        logdet_new = x.det_log; sign_new = x.det_sign;
        for (int g = 0; g < nf; ++g) {
            const int f = fsrc[g], k = x.order(f), kk = k * k;
            const double *Mo = x.mat(f); double *Mn = MBw + (size_t)g * kmax * kmax;
            const unsigned magic = k > 0 ? (1u << 24) / (unsigned)k + 1u : 0u;
            for (int e = x.lane; e < kk; e += 64) { const int i = (int)(((unsigned)e * magic) >> 24), o = i * kmax + (e - i * k); Mn[o] = Mo[o]; }
        }
    } else
    for (int g = 0; g < nf; ++g) {
        double ld, sg;
        const int f = fsrc[g], k = x.order(f);
        const double *tcg = flip ? x.ta(f) : x.tc(f), *tag = flip ? x.tc(f) : x.ta(f);
        build_and_invert(x, g, k, tcg, tag, MBw + (size_t)g * kmax * kmax, ld, sg);
        GLOB_STAMP(x, C_DBG1);
        if (k > 0) sg *= sort_parity(x, tcg, k) * sort_parity(x, tag, k);
        GLOB_STAMP(x, C_DBG2);
        logdet_new += ld; sign_new *= sg;
    }
    double rnd = x.draw();
    if (x.err & ERR_SINGULAR) return;
    double ratio = fabs(exp((ltr_new - ltr_old) + (logdet_new - x.det_log)) * sst_factor);
    if (!(rnd < ratio)) return;
    // accept: install relabelled taus and new matrices
    const int kold = x.kreg;         // lane f: order of old flavour f
    wave_sync();
    for (int g = 0; g < nf; ++g) {
        const int k = __builtin_amdgcn_readlane(kold, fsrc[g]);
        double *tcp = x.tc(g), *tap = x.ta(g), *Mg = x.mat(g);
        const double *Mn = MBw + (size_t)g * kmax * kmax;
        for (int e = x.lane; e < k; e += 64) { tcp[e] = TS[(size_t)(2 * g) * kmax + e]; tap[e] = TS[(size_t)(2 * g + 1) * kmax + e]; }
        const int kk = k * k;
        const unsigned magic = k > 0 ? (1u << 24) / (unsigned)k + 1u : 0u;
        for (int e0 = 0; e0 < kk; e0 += 256) {
            double mv[4]; int ov[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int e = e0 + u * 64 + x.lane; int ec = e < kk ? e : kk - 1;
                int i = (int)(((unsigned)ec * magic) >> 24), j = ec - i * k;
                ov[u] = i * kmax + j; mv[u] = Mn[ov[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { int e = e0 + u * 64 + x.lane; if (e < kk) Mg[ov[u]] = mv[u]; }
        }
        x.set_order(g, k);
    }
    wave_sync();
    x.outer = outer_new;
    x.det_log = logdet_new; x.det_sign = sign_new;
    x.npairs = -1;
    GLOB_STAMP(x, C_DBG4);
    x.count(C_ACCGLOB);
}

// ---------------------------------------------------------------------------------------------------------------
// chain state: HBM <-> LDS/registers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ctx_setup(Ctx &x, unsigned char *smem, int chain = -1)
{
    const DevConst &C = x.C;
    int wib = threadIdx.x >> 6;
    x.lane = threadIdx.x & 63;
    x.w = chain >= 0 ? chain : uni((int)(blockIdx.x * (blockDim.x >> 6) + wib));
    unsigned char *base = smem + C.shared_lds + (size_t)wib * C.lds_per_wave;
    x.tc_ = (double *)base; x.ta_ = x.tc_ + C.nflav * C.kmax;
    x.MQ = x.ta_ + C.nflav * C.kmax; x.RM = x.MQ + C.kmax;
    x.Q = x.RM + C.kmax; x.R = x.Q + C.kmax; x.wt = x.R + C.kmax;
    x.wm = (int *)(x.wt + C.wcap); x.wc = x.wm + C.wcap; x.wp = x.wc + C.wcap;
    {
        const size_t u1 = ((size_t)2 * C.kmax + C.wcap) * sizeof(double) + (size_t)3 * C.wcap * sizeof(int), u2 = (size_t)C.kgj * C.kgj * sizeof(double);
        x.gj = C.mat ? x.Q + u1 / 8 : x.Q;                  // scalar path: the tile shares its space with Q, R and the window list
        x.cnt = (unsigned long long *)(x.Q + ((C.mat ? u1 + u2 : (u1 > u2 ? u1 : u2)) + 7) / 8);
    }
    x.K = (int *)(x.cnt + C_NCOUNT); x.piv = x.K + MAXFLAV;
    x.nops = 0; x.tr_log = 0.0; x.tr_sign = 1.0; x.ms_ = 0; x.mi_ = 0; x.spb_ = 1;
    x.pmat = nullptr; x.pvalid = 0; x.p_lo = x.p_hi = -1;
    x.kc = nullptr; x.kval = 0; x.k_i1 = 0; x.k_first = 0;
    if (C.mat) {
        // per-wave list arrays behind the common slice; block-shared tables at the start of the dynamic LDS
        x.om = x.piv + C.kmax; x.osst = x.om + C.wcap; x.tsst = x.osst + C.wcap + 2;
        x.ot = (double *)(((size_t)(x.tsst + C.wcap + 2) + 7) & ~(size_t)7);
        x.opbuf = x.ot + C.wcap + 2;
        { const int nt = C.nsm <= 8 ? 8 : (C.nsm <= 10 ? 10 : 16); x.pmat = x.opbuf + (16 + 64) * nt;
          if (C.kcache) x.kc = C.kcache + (size_t)x.w * 131 * (nt * nt + 1); }
        const int nf2 = 2 * C.nflav;
        short *ti = (short *)smem;
        x.t_conn = ti; x.t_shift = ti + C.nsst * nf2; x.t_n = (const int *)(ti + 2 * C.nsst * nf2); x.t_off = x.t_n + C.nsst;
        double *td = (double *)(((size_t)(x.t_off + C.nsst) + 7) & ~(size_t)7);
        x.t_eval = td;
        x.t_op = C.op_in_lds ? td + C.nstates : nullptr;
        x.ms_ = x.lane / C.nsm; x.mi_ = x.lane - x.ms_ * C.nsm; x.spb_ = C.nsm <= 8 ? C.nsm : 64 / C.nsm;
    }
}

// stage the block-shared tables of the local problem (all threads of the workgroup; call before any early return)
__device__ __forceinline__ void stage_shared_tables(const DevConst &C, unsigned char *smem)
{
    if (!C.mat) return;
    const int nf2 = 2 * C.nflav;
    short *ti = (short *)smem;
    short *t_shift = ti + C.nsst * nf2; int *t_n = (int *)(ti + 2 * C.nsst * nf2), *t_off = t_n + C.nsst;
    double *td = (double *)(((size_t)(t_off + C.nsst) + 7) & ~(size_t)7);
    for (int i = threadIdx.x; i < C.nsst * nf2; i += blockDim.x) { ti[i] = (short)C.conn[i]; t_shift[i] = (short)C.shift[i]; }
    for (int i = threadIdx.x; i < C.nsst; i += blockDim.x) { t_n[i] = C.sst_n[i]; t_off[i] = C.sst_off[i]; }
    for (int i = threadIdx.x; i < C.nstates; i += blockDim.x) td[i] = C.sst_eval[i];
    if (C.op_in_lds) {
        const int nop = C.nsst * nf2 * C.nsm * C.nsm;
        for (int i = threadIdx.x; i < nop; i += blockDim.x) td[C.nstates + i] = C.opmat[i];
    }
    __syncthreads();
}

// MATPATH = false: a kernel of the closed-form trace path, which never has the time-ordered list.  WORMOPS = false: a Z-space kernel, which
// never looks at the worm operators of the record (and ctx_store<false> leaves them as they are).  Neither is used by the sweep kernels: measured
// on the headline workload, both made k_sweep<0> SLOWER (25.2 against 24.2 ms per launch; the register allocation of the kernel's hot loop
// follows what its prologue keeps alive), and so did a run-time loop over the record in the list branch below (25.6 ms), which is why the
// operators beyond NWMAX join the list only where WORMREC asks for it.
// WORMREC = true: worm operators beyond those the kernel keeps in registers join the list from the record (the estimator kernels, which see the
// chains of every sector)
template <bool MATPATH = true, bool WORMOPS = true, bool WORMREC = false>
__device__ __forceinline__ void ctx_load(Ctx &x)
{
    const DevConst &C = x.C;
    const int w = x.w, nf = C.nflav, kmax = C.kmax;
    x.M = C.M + (size_t)w * nf * kmax * kmax;
    const WalkerScalars &s = C.ws[w];
    x.outer = uni(s.outer); x.npairs = uni(s.npairs); x.wpos = uni(s.wpos); x.err = uni(s.err);
    x.det_sign = uni(s.det_sign); x.det_log = uni(s.det_log);
    { const int sec = uni(s.sector); x.sector = (sec >= 2 && sec <= 16) ? sec : 1; }
    if constexpr (WORMOPS) {
#pragma unroll
        for (int i = 0; i < NWMAX; ++i) { x.wtau[i] = uni(s.wtau[i]); x.wfl[i] = uni(s.wflav[i]); }
    }
    unsigned long long nd_ = s.ndraw;
    nd_ = ((unsigned long long)(unsigned)uni((int)(nd_ >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)nd_);
    if (x.lane < C_NCOUNT) x.cnt[x.lane] = s.cnt[x.lane];
    x.cntreg = 0u;
    x.kreg = (x.lane < nf) ? C.K[(size_t)w * nf + x.lane] : 0;
    wave_sync();
    for (int f = 0; f < nf; ++f) {
        int k = x.order(f);
        const double *gc = C.TC + ((size_t)w * nf + f) * kmax, *ga = C.TA + ((size_t)w * nf + f) * kmax;
        for (int e = x.lane; e < k; e += 64) { x.tc(f)[e] = gc[e]; x.ta(f)[e] = ga[e]; }
    }
    x.nwin = 0; x.wvalid = 0;
    x.window_bounds();
    x.rng_seek(nd_);
    wave_sync();
    if constexpr (MATPATH) if (C.mat) {      // time-ordered list, block sequence and trace of the stored configuration
        mat_build_list(x);
        if (x.sector != 1) {                         // worm operators are list entries like any other (no hybridisation lines: not in M)
            if constexpr (WORMOPS) {
#pragma unroll
                for (int i = 0; i < NWMAX; ++i)
                    if (i < worm_nops_of(x.sector)) mat_insert(x, mat_lower(x, x.wtau[i]), x.wtau[i], mat_code(x.wfl[i], (i & 1) ? 1 : 2));
            }
            if constexpr (WORMREC) {
                for (int i = WORMOPS ? NWMAX : 0, nwo = worm_nops_of(x.sector); i < nwo; ++i) {
                    const double tw = uni(s.wtau[i]);
                    mat_insert(x, mat_lower(x, tw), tw, mat_code(uni(s.wflav[i]), (i & 1) ? 1 : 2));
                }
            }
        }
        mat_walk_sst(x);                 // the caller evaluates the trace (mat_trace_current<MAT>)
    }
}

template <bool WORMOPS = true>
__device__ __forceinline__ void ctx_store(Ctx &x, long long todo = 0)
{
    const DevConst &C = x.C;
    const int w = x.w, nf = C.nflav, kmax = C.kmax;
    wave_sync();
    {   // what the host looks at between launches: the largest order stored so far (read first: the atomic is rarely needed)
        int mk = 0;
        for (int f = 0; f < nf; ++f) mk = max(mk, x.order(f));
        if (x.lane == 0) {
            if (mk > C.flags[0]) atomicMax(&C.flags[0], mk);
            if (todo > 0) atomicAdd(&C.flags[1], 1);
        }
    }
    for (int f = 0; f < nf; ++f) {
        int k = x.order(f);
        double *gc = C.TC + ((size_t)w * nf + f) * kmax, *ga = C.TA + ((size_t)w * nf + f) * kmax;
        for (int e = x.lane; e < k; e += 64) { gc[e] = x.tc(f)[e]; ga[e] = x.ta(f)[e]; }
    }
    WalkerScalars &s = C.ws[w];
    if (x.lane < C_NCOUNT) s.cnt[x.lane] = x.cnt[x.lane] + (unsigned long long)x.cntreg;
    if (x.lane < nf) C.K[(size_t)w * nf + x.lane] = x.kreg;
    if (x.lane == 0) {
        s.ndraw = x.rng_tell(); s.outer = x.outer; s.npairs = x.npairs; s.wpos = x.wpos; s.err = x.err; s.todo = todo;
        s.det_sign = x.det_sign; s.det_log = x.det_log;
        s.sector = x.sector;
        if constexpr (WORMOPS) {
#pragma unroll
            for (int i = 0; i < NWMAX; ++i) { s.wtau[i] = x.wtau[i]; s.wflav[i] = x.wfl[i]; }
        }
    }
}

// Steps the chain has to do in this launch.  A normal launch (resume = 0) gives every chain nsteps; a chain still paused at
// capacity from an earlier launch only notes them down.  A resume launch lets the paused chains finish what they owe.
__device__ __forceinline__ long long launch_steps(const DevConst &C, int w, int lane, long long nsteps, int resume)
{
    WalkerScalars &s = C.ws[w];
    const long long owed = ((long long)uni((int)(s.todo >> 32)) << 32) | (long long)(unsigned)uni((int)s.todo);
    if (resume) return owed;
    if (owed > 0) { if (lane == 0) { s.todo = owed + nsteps; atomicAdd(&C.flags[1], 1); } return 0; }
    return nsteps;
}
// next chain of the launch for this wave (matrix trace path); beyond the last one every wave gets nwalkers or more and leaves
__device__ __forceinline__ int next_chain(const DevConst &C)
{
    int w = 0;
    if ((threadIdx.x & 63) == 0) w = atomicAdd(C.queue, 1);
    return __builtin_amdgcn_readfirstlane(w);
}

// a chain loaded with a full flavour (set_mc_config with k = kmax) pauses before its first step
__device__ __forceinline__ void flag_full(Ctx &x, int spare = 0)
{
    if (__ballot(x.lane < x.C.nflav && x.kreg >= x.C.kmax - spare) != 0ull) x.err |= ERR_KMAX;
}

#include "cthyb_worm.hip.h"
#include "cthyb_xmoves.hip.h"
#include "cthyb_offdiag.hip.h"

#ifdef W2DYN_STAMPS
#define STAMP_KIND(c) which_ = (c)
#else
#define STAMP_KIND(c)
#endif

// sweep kernel: nsteps iterations of the Z-space loop (CTQMC.F90:8213-8257 / 8428-8533) for every chain
// CALIB: the loop of ctqmc_calibrate (CTQMC.F90:8121-8174): same moves, no window shift, no step counting; a chain
// whose own calibration criterion is met does nothing
#ifndef W2DYN_WPE
#define W2DYN_WPE 2      // the scalar-path kernel sits right at 256 registers: never let it tip over into one wave per SIMD
#endif

// WPE3: the same kernel held to 168 registers (three waves per SIMD).  With the branch weights below the ~90 spilled
// registers land mostly in the rare moves, and the step rate rises by 6-10 % for up to three orbitals (C1, C2, C3); with five
// (C5Z: longer flavour loops everywhere) the spills cost more than the third wave brings.  The host chooses (walker_launch).
// XM: the moves outside the default mix (two pairs at once, pair moves that change the outer block; cthyb_xmoves.hip.h)
template <int MAT, bool CALIB = false, bool WPE3 = false, bool XM = false>
__global__ __launch_bounds__(256, MAT == 0 ? (WPE3 ? 3 : W2DYN_WPE) : W2DYN_WPE_MAT) void k_sweep(DevConst C, long long nsteps_launch, int count_sampling, int resume)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    // scalar path: wave = chain by position in the grid (one pass of the loop); matrix path: chains from the launch's queue
    for (bool first = true;; first = false) {
    if constexpr (MAT == 0) { if (!first) break; ctx_setup(x, smem); } else ctx_setup(x, smem, next_chain(C));
    if (x.w >= C.nwalkers) break;
    if constexpr (CALIB) if (!resume) {
        bool done;
        if (C.calib == 1) done = C.apt_cnt[x.w] >= C.napt;
        else {
            done = true;
            for (int f = 0; f < C.nflav; ++f) done = done && C.calib_pairs[(size_t)x.w * C.nflav + f] >= C.phase2pairnum;
        }
        if (uni((int)done)) continue;
    }
    const long long nsteps = launch_steps(C, x.w, x.lane, nsteps_launch, resume);
    if (nsteps == 0) continue;
    ctx_load(x);
    flag_full(x, XM ? 1 : 0);
    if constexpr (MAT != 0) mat_trace_current<MAT>(x);
    const double pg = C.p_glob, ps = C.p_shift;
    // position inside the window-shift period: (TryAdd+TryRem) mod NSlide (CTQMC.F90:8255, 8531)
    unsigned int phase = (unsigned int)((x.cnt[C_TRYADD] + x.cnt[C_TRYREM]) % (unsigned long long)C.nslide);
    phase = (unsigned int)uni((int)phase);
    long long it = 0;
    for (; it < nsteps && !(x.err & ERR_KMAX); ++it) {
        double r = x.draw();
#ifdef W2DYN_STAMPS
        unsigned long long t0_ = __builtin_amdgcn_s_memtime();
        int which_;
#endif
        // The chain of moves twice: with branch weights in the three-waves variant - the rare moves contain deep loop nests,
        // which the register allocator would otherwise take for the hot part of the kernel and spill the pair moves instead -,
        // without in the others, where nothing spills and the weights only disturb the block layout (measured: 4 % slower).
        if constexpr (WPE3) {
            if (r < (1.0 - ps - pg) / 2.0) { x.skip(2);
#ifndef W2DYN_NO_ADD
                step_add<MAT, false, CALIB>(x);
#endif
                STAMP_KIND(C_CYC_ADD); phase += 1; }
            else if (__builtin_expect(r <= 1.0 - ps - pg, 1)) { x.skip(2);
#ifndef W2DYN_NO_REM
                step_rem<MAT, false, CALIB>(x);
#endif
                STAMP_KIND(C_CYC_REM); phase += 1; }
            else if (__builtin_expect(r <= 1.0 - ps, 0)) {
#ifndef W2DYN_NO_GLOB
                if constexpr (MAT != 0) step_glob_mat<MAT>(x); else step_glob(x);
#endif
                STAMP_KIND(C_CYC_GLOB); }
            else {
#ifndef W2DYN_NO_SHIFT
                step_shift<MAT>(x);
#endif
                STAMP_KIND(C_CYC_SHIFT); }
        } else if constexpr (XM) {
            // CTQMC.F90:8441-8449, 8456-8464: a two-pair move with probability Percentage4OperatorMove, else the pair move,
            // which may change the outer block with probability PercentageOuterMove
            const bool add = r < (1.0 - ps - pg) / 2.0;
            if (add || r <= 1.0 - ps - pg) {
                if (x.draw() >= C.p_4op) {
                    const bool outer = x.draw() < C.p_outer;
                    if (add) step_add<MAT, 0, false, true>(x, outer); else step_rem<MAT, 0, false, true>(x, outer);
                    phase += 1;
                } else if (add) step_add4<MAT>(x); else step_rem4<MAT>(x);
                STAMP_KIND(add ? C_CYC_ADD : C_CYC_REM);
            } else if (r <= 1.0 - ps) {
                if constexpr (MAT != 0) step_glob_mat<MAT>(x); else step_glob(x);
                STAMP_KIND(C_CYC_GLOB);
            } else {
                step_shift<MAT>(x);
                STAMP_KIND(C_CYC_SHIFT);
            }
        } else {
            if (r < (1.0 - ps - pg) / 2.0) { x.skip(2);
#ifndef W2DYN_NO_ADD
                step_add<MAT, false, CALIB>(x);
#endif
                STAMP_KIND(C_CYC_ADD); phase += 1; }
            else if (r <= 1.0 - ps - pg) { x.skip(2);
#ifndef W2DYN_NO_REM
                step_rem<MAT, false, CALIB>(x);
#endif
                STAMP_KIND(C_CYC_REM); phase += 1; }
            else if (r <= 1.0 - ps) {
#ifndef W2DYN_NO_GLOB
                if constexpr (MAT != 0) step_glob_mat<MAT>(x); else step_glob(x);
#endif
                STAMP_KIND(C_CYC_GLOB); }
            else {
#ifndef W2DYN_NO_SHIFT
                step_shift<MAT>(x);
#endif
                STAMP_KIND(C_CYC_SHIFT); }
        }
#ifdef W2DYN_STAMPS
        if (x.lane == 0) x.cnt[which_] += __builtin_amdgcn_s_memtime() - t0_;
#endif
        if constexpr (CALIB) continue;
        if (phase == (unsigned int)C.nslide) phase = 0;
        if (phase == 0) {                                        // shift_window, Trace.F90:1652-1678
            if (C.num_windows > 1) {
                x.wpos += 1; if (x.wpos == C.num_windows) x.wpos = 0;
                x.window_bounds(); x.npairs = -1; x.wvalid = 0; x.pvalid = 0;
            } else x.wpos = 0;
        }
    }
    if (count_sampling && x.lane == 0) x.cnt[C_SAMPLING] += (unsigned long long)it;
    ctx_store(x, nsteps - it);
    }
}

// The library is built from several translation units (csrc/tu_*.hip, launch.hpp) that all include this header; kernel
// templates are instantiated where they are launched, the plain kernels below are compiled by the one unit that
// defines their W2DYN_TU_* macro.
#ifdef W2DYN_TU_SCALAR
// rebuild M, det and NPairs of every chain from its tau arrays (set_mc_config, Trace.F90:6972-6999)
__global__ __launch_bounds__(256) void k_rebuild(DevConst C)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load(x);
    double ld = 0.0, sg = 1.0;
    for (int f = 0; f < C.nflav; ++f) {
        double l, s;
        int k = x.order(f);
        build_and_invert(x, f, k, x.tc(f), x.ta(f), x.mat(f), l, s);
        if (k > 0) s *= sort_parity(x, x.tc(f), k) * sort_parity(x, x.ta(f), k);
        ld += l; sg *= s;
    }
    x.det_log = ld; x.det_sign = sg;
    x.npairs = build_window(x);
    ctx_store(x);
}
#endif  // W2DYN_TU_SCALAR

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
template <int MAT>
__global__ __launch_bounds__(256) void k_meas_misc(DevConst C)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load<true, true, true>(x);
    if constexpr (MAT != 0) if (C.offdiag) od_load(x);  // off-diagonal hybridisation: the list with the true flavours
    double pt_ = 0.0, lsc_ = 0.0; int sk_ = 0;          // blocks of more than 4 states: partial traces and the block at beta/2 (mat_halves)
    constexpr int NT2 = (MAT >= 2 && MAT <= 4) ? MAT * MAT : 1;
    double Kt_[NT2], At_[NT2];                          // blocks of up to 4 states: the two halves as matrices (lane 0; mat_halves_tree), sk_ = operators up to beta/2
    if constexpr (MAT > 4) mat_halves(x, pt_, lsc_, sk_);
    else if constexpr (MAT != 0) mat_halves_tree<MAT>(x, Kt_, At_, pt_, lsc_, sk_);
    const int nf = C.nflav, nd = C.nd;
    int par = 0;
    double pairsum = 0.0;            // one-state blocks: this lane's part of sum_{f<g} U_fg D_fg (scalar_logtrace)
    // every lane takes one operator (f, type, slot) of the flat list of all flavours' operators (so the lanes stay busy
    // whatever the orders of the single flavours are) and scans the operators that can contribute a transposition
    int N = 0;
    for (int f = 0; f < nf; ++f) N += 2 * x.order(f);
    bool odiag = false;
    if constexpr (MAT != 0) odiag = C.offdiag != 0;
    for (int base = 0; base < N && !odiag; base += 64) {
        const int li = base + x.lane;
        const bool act = li < N;
        int fi = 0, oi = 0, ki = 0;
        for (int f = 0, of = 0; f < nf; ++f) {
            const int k2 = 2 * x.order(f);
            if (li >= of && li < of + k2) { fi = f; oi = li - of; ki = k2 >> 1; }
            of += k2;
        }
        const int biti = flav_bit(fi, nd);
        const bool ci = oi < ki;
        const double ti = !act ? 0.0 : (ci ? x.tc_[fi * C.kmax + oi] : x.ta_[fi * C.kmax + oi - ki]);
        if (x.sector != 1 && act) {
            // worm operators enter the trace sign like any other operator; get_WormSign (Trace.F90:3623-3651) adds the
            // hybridising operators at later times than each worm operator
            const WalkerScalars &wsr = C.ws[x.w];             // all worm operators of the record (up to twelve)
            for (int i = 0, nwo = worm_nops_of(x.sector); i < nwo; ++i) {
                const int bw = flav_bit(wsr.wflav[i], nd); const double tw = wsr.wtau[i];
                if (!MAT) par ^= ((tw < ti && bw < biti) ? 1 : 0) ^ ((ti < tw && biti < bw) ? 1 : 0);   // trace sign (scalar path only)
                par ^= (ti > tw) ? 1 : 0;
            }
        }
        double uacc = 0.0;           // one-state blocks: sum_g U(fi, g) (creators - annihilators of g before ti), for the log trace
        for (int fj = 0; fj < nf; ++fj) {
            const int kj = x.order(fj);
            const int bitj = flav_bit(fj, nd);
            const bool wtr = act && !MAT && bitj < biti, wbl = act && fj < fi, wsame = act && (fj == fi) && ci;
            if (MAT && !__ballot(wtr || wbl || wsame)) continue;
            const double *cj = x.tc(fj), *aj = x.ta(fj);
            // operators of flavour fj at earlier times; the later ones are the rest (times are distinct; the operator itself
            // is neither)
            int cb = 0, ab = 0;
            for (int lj = 0; lj < kj; ++lj) { cb += cj[lj] < ti; ab += aj[lj] < ti; }
            const int below = cb + ab;
            const int above = 2 * kj - below - ((act && fj == fi) ? 1 : 0);
            if (wtr) par ^= below & 1;
            if (wbl) par ^= above & 1;
            if (wsame) par ^= above & 1;
            if constexpr (!MAT) { if (act && fj != fi && cb != ab) uacc = fma(C.udd[fi * nf + fj], (double)(cb - ab), uacc); }
        }
        if constexpr (!MAT) { if (act) pairsum += (ci ? 1.0 : -1.0) * (C.beta - ti) * uacc; }
    }
    const bool worm = x.sector != 1;
    if (worm && x.lane == 0) {
        // worm operators among themselves: trace sign, and the pairs i < j (container order) with tau_i < tau_j
        const WalkerScalars &wsr = C.ws[x.w];
        const int nwo = worm_nops_of(x.sector);
        for (int i = 0; i < nwo; ++i)
            for (int j = 0; j < nwo; ++j) {
                if (!MAT && wsr.wtau[j] < wsr.wtau[i] && flav_bit(wsr.wflav[j], nd) < flav_bit(wsr.wflav[i], nd)) par ^= 1;
                if (i < j && wsr.wtau[i] < wsr.wtau[j]) par ^= 1;
            }
    }
    par = wave_sum(par) & 1;
    if ((N / 2) & 1) par ^= 1;
    if constexpr (MAT != 0) if (odiag) par = od_sign_parity(x);      // Trace.F90:3483-3486: trace sign, determinant sign, recalc_sign
    double sign = x.det_sign * (par ? -1.0 : 1.0);
    if constexpr (MAT) sign *= x.tr_sign;            // the trace sign comes with the matrix elements (Trace.F90:3488)
    double cfgsign = -sign;
    const int bin = x.w / C.walkers_per_bin;
    if (worm) {
        // a chain in worm space contributes to the worm estimator only (CTQMC.F90:8586-8593); k_meas_g4w picks up its sign
        if (x.lane == 0) { C.cfgsign[x.w] = cfgsign; atomicAdd(&C.acc_nw[bin], 1.0); }
        return;
    }
    if constexpr (MAT) {
        // MeasDensityMatrix_beta_half (CTQMC.F90:3088-3151): kets from 0 to beta/2 through the operators with
        // tau <= beta/2, bras from beta down to beta/2, for all states of the outer block at once
        if constexpr (MAT >= 2 && MAT <= 4) { if (x.tr_sign != 0.0) mat_dm_tree_from<MAT>(x, Kt_, At_, sk_, -cfgsign * x.tr_sign * exp(lsc_ - x.tr_log), bin); }
        else if (x.tr_sign != 0.0) mat_dm_from_halves(x, sk_, -cfgsign * x.tr_sign / fabs(wave_sum(pt_)), bin);
    } else {
    // state alive at beta/2 (Trace.F90:6334: operators with tau <= beta/2 have acted)
    int sb2 = x.outer;
    for (int f = 0; f < nf; ++f) {
        int k = x.order(f), c = 0;
        for (int base = 0; base < k; base += 64) {
            int e = base + x.lane; bool v = e < k;
            c += count_pred(v && x.tc(f)[e] <= 0.5 * C.beta) + count_pred(v && x.ta(f)[e] <= 0.5 * C.beta);
        }
        if (c & 1) sb2 ^= 1 << flav_bit(f, nd);
    }
    if (x.lane == 0) atomicAdd(&C.acc_dm[(size_t)bin * C.dmsize + sb2], -cfgsign);
    }
    // MeasOuterHistograms (CTQMC.F90:3285-3339): sign and weight exp(log|partial trace|) of every state of the outer block
    if constexpr (MAT == 0) {
        const double ptot = wave_sum(pairsum);
        const double c = exp(scalar_logtrace(x, &ptot));   // one state per block: the partial trace is the trace
        if (x.lane == 0) {
            atomicAdd(&C.acc_tcstate[(size_t)bin * C.nstates + x.outer], c);
            atomicAdd(&C.acc_signstate[(size_t)bin * C.nstates + x.outer], cfgsign);
        }
    } else {
        const double pt = pt_, lsc = lsc_;                 // lane i < n: partial trace of outer state i on the common scale exp(lsc)
        const int n = uni(x.t_n[x.outer]), off = uni(x.t_off[x.outer]);
        if (x.lane < n) {
            atomicAdd(&C.acc_tcstate[(size_t)bin * C.nstates + off + x.lane], fabs(pt) * exp(lsc));
            atomicAdd(&C.acc_signstate[(size_t)bin * C.nstates + off + x.lane], x.tr_sign * (pt < 0.0 ? -1.0 : 1.0) * cfgsign);
        }
    }
    if (x.lane == 0) {
        C.cfgsign[x.w] = cfgsign;
        atomicAdd(&C.acc_sign[bin], cfgsign);
        atomicAdd(&C.acc_nz[bin], 1.0);
        atomicAdd(&C.acc_outer[(size_t)bin * C.nouter + x.outer], 1.0);
        atomicAdd(&C.acc_signouter[(size_t)bin * C.nouter + x.outer], cfgsign);
    }
    if (x.lane < nf) {      // lane f holds the order of flavour f
        if constexpr (MAT != 0) if (odiag) {      // NOSOper / 2 of the flavour: creators and annihilators counted together
            const int kt = x.order(0); const int *fc = od_fc(x), *fa = od_fa(x);
            int c = 0;
            for (int e = 0; e < kt; ++e) c += (fc[e] == x.lane ? 1 : 0) + (fa[e] == x.lane ? 1 : 0);
            x.kreg = c / 2;
        }
        int h = x.kreg < C.maxhisto ? x.kreg : C.maxhisto;
        atomicAdd(&C.acc_histo[((size_t)bin * nf + x.lane) * (C.maxhisto + 1) + h], 1.0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MeasSusz on the matrix trace path: <n_1(tauref) n_2(tauref + t beta/Ntau)> with the densities inserted into the operator
// sequence (meas_ntau_n0, Trace.F90:6472-6835: densities applied to the stored bra / ket vectors of the outer states).
// tauref is the upper edge of the chain's window; points beyond beta are taken at tauref - (Ntau - t) beta/Ntau as there.
// One wave per chain, lane = (outer state s, component i) as in mat_apply (blocks of up to 8 states).  Per chain:
//   1. kets K_s[j] right after operator j-1 and bras B_s[j] right in front of operator j, j = 0 .. n (each set of vectors
//      scaled to a largest element of one, the log of the scale kept with it), to the chain's slice of a global work area;
//   2. for every flavour f1: the kets with n_f1 inserted at tauref for the positions behind it (K'_s[j], j > jr, and the
//      vectors at tauref itself in slot jr) and the bras with n_f1 inserted for the positions in front of it;
//   3. lane = tau point: the point's interval j is found by bisection over the operator times, ket and bra of that interval are
//      brought to the point with the interval's decay factors, and  sum_s B_s n_f2 K_s  is formed for every f2 with the
//      density matrices N[block][f2] = Op(c+_f2) Op(c_f2) of the eigenbasis (host-made table `dens`):
//          ntau_n0[f1][f2][t] -= cfgsign * Tr[... n_f2 ... n_f1 ...] / Tr[...]
// Cost per chain: O(nflav (n + Ntau) nsm^2 n_outer) instead of nflav^2 Ntau traces from scratch.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void susz_store(double *arr, int ent, int j, int slot, bool act, double v, double l, int lane)
{
    if (act) arr[(size_t)j * ent + slot] = v;
    if (lane == 0) arr[(size_t)j * ent + ent - 1] = l;
}
// v <- N v inside block sst (N = density matrix of flavour f in the block's eigenbasis, row-major [target][source])
__device__ __forceinline__ double susz_dens_ket(Ctx &x, const double *dens, int sst, int f, double v)
{
    const int nsm = x.C.nsm;
    const double *N = dens + (size_t)(sst * x.C.nflav + f) * nsm * nsm;
    const bool act = x.lane < nsm * nsm;
    double nv = 0.0;
    for (int j = 0; j < nsm; ++j) {
        const double vj = __shfl(v, x.ms_ * nsm + j, 64);
        nv = fma(act ? N[x.mi_ * nsm + j] : 0.0, vj, nv);
    }
    return nv;
}
// b <- b N (bra side)
__device__ __forceinline__ double susz_dens_bra(Ctx &x, const double *dens, int sst, int f, double b)
{
    const int nsm = x.C.nsm;
    const double *N = dens + (size_t)(sst * x.C.nflav + f) * nsm * nsm;
    const bool act = x.lane < nsm * nsm;
    double nb = 0.0;
    for (int i = 0; i < nsm; ++i) {
        const double bi = __shfl(b, x.ms_ * nsm + i, 64);
        nb = fma(act ? N[i * nsm + x.mi_] : 0.0, bi, nb);
    }
    return nb;
}
// one step of the bras towards earlier times: decay over dt inside block `src` (the block behind operator `code`), then the
// operator from the right: b[m] <- sum_i b[i] e^{-(E_i - E_0) dt} Op[i][m], Op taking block `sst` into `src`
__device__ __forceinline__ void susz_bra_step(Ctx &x, int sst, int code, int src, double dt, double &b, double &lb)
{
    const DevConst &C = x.C;
    const int nsm = C.nsm, nf2 = 2 * C.nflav;
    const int n = uni(x.t_n[src]), off = uni(x.t_off[src]);
    const double e0 = uni(x.t_eval[off]);
    if (n > 1) b *= exp(-(((x.mi_ < n) ? x.t_eval[off + x.mi_] : e0) - e0) * dt);
    lb -= (e0 - C.egs) * dt;
    const size_t base = (size_t)(sst * nf2 + code) * nsm * nsm;
    const bool act = x.lane < nsm * nsm;
    double nb = 0.0;
    for (int i = 0; i < n; ++i) {
        const double bi = __shfl(b, x.ms_ * nsm + i, 64);
        const double o = act ? (x.t_op ? x.t_op[base + i * nsm + x.mi_] : C.opmat[base + i * nsm + x.mi_]) : 0.0;
        nb = fma(o, bi, nb);
    }
    b = nb;
}
template <int NSMT>
__global__ __launch_bounds__(256) void k_meas_susz_mat(DevConst C, double *__restrict__ acc, int nbs, double *__restrict__ work, const double *__restrict__ dens)
{
    extern __shared__ unsigned char smem[];
    stage_shared_tables(C, smem);
    Ctx x(C);
    ctx_setup(x, smem);
    if (x.w >= C.nwalkers) return;
    ctx_load(x);
    if (x.sector != 1 || (x.err & (ERR_WCAP | ERR_BLOCKSEQ))) return;      // chains in worm space measure no Z-space estimator
    const int nsm = C.nsm, nf = C.nflav, n = x.nops, ntau = C.ntau, ent = nsm * nsm + 1;
    const int outer = x.outer, nout = uni(x.t_n[outer]);
    const double beta = C.beta, tauD = beta / (double)ntau, tauref = x.wmax;
    double *SK = work + (size_t)x.w * 4 * (C.wcap + 1) * ent, *SB = SK + (size_t)(C.wcap + 1) * ent, *SX = SB + (size_t)(C.wcap + 1) * ent, *SY = SX + (size_t)(C.wcap + 1) * ent;
    const bool act = x.lane < nsm * nsm;
    const int slot = x.lane;
    const bool diag = act && x.ms_ == x.mi_ && x.ms_ < nout;
    if (uni(x.osst[n]) != outer) return;                                     // not a closed sequence: nothing to measure
    // 1a. kets
    double v = diag ? 1.0 : 0.0, lacc = 0.0, tprev = 0.0;
    susz_store(SK, ent, 0, slot, act, v, lacc, x.lane);
    for (int j = 0; j < n; ++j) {
        const double tau = uni(x.ot[j]);
        if (mat_apply(x, uni(x.osst[j]), uni(x.om[j]), tau - tprev, v, lacc) < 0) return;
        if (!mat_rescale(v, lacc)) return;                                   // the trace vanishes
        susz_store(SK, ent, j + 1, slot, act, v, lacc, x.lane);
        tprev = tau;
    }
    double ltr0, sign0;
    {
        double vf = v, lf = lacc;
        mat_evolve(x, outer, beta - tprev, vf, lf);
        const double tot = wave_sum(diag ? vf : 0.0);
        if (tot == 0.0 || !(fabs(tot) > 0.0)) return;
        ltr0 = lf + log(fabs(tot)); sign0 = tot < 0.0 ? -1.0 : 1.0;
    }
    // 1b. bras
    {
        double b = diag ? 1.0 : 0.0, lb = 0.0, tnext = beta;
        susz_store(SB, ent, n, slot, act, b, lb, x.lane);
        for (int j = n - 1; j >= 0; --j) {
            const double tau = uni(x.ot[j]);
            susz_bra_step(x, uni(x.osst[j]), uni(x.om[j]), uni(x.osst[j + 1]), tnext - tau, b, lb);
            if (!mat_rescale(b, lb)) return;
            susz_store(SB, ent, j, slot, act, b, lb, x.lane);
            tnext = tau;
        }
    }
    int jr = 0;                                                              // operators in front of tauref
    for (int base = 0; base < n; base += 64) { const int e = base + x.lane; jr += count_pred(e < n && x.ot[e] < tauref); }
    const int br = uni(x.osst[jr]);
    const double t_before = jr > 0 ? uni(x.ot[jr - 1]) : 0.0, t_after = jr < n ? uni(x.ot[jr]) : beta;
    const double cfg = C.cfgsign[x.w];
    const int ab = (x.w / C.walkers_per_bin) % nbs;
    double *out = acc + (size_t)ab * nf * nf * ntau;
    wave_sync();
    for (int f1 = 0; f1 < nf; ++f1) {
        // 2a. kets with n_f1 at tauref: slot jr holds the vectors AT tauref, slots j > jr those right after operator j-1
        bool kets_ok = true, bras_ok = true;
        {
            double k1 = act ? SK[(size_t)jr * ent + slot] : 0.0, lk = uni(SK[(size_t)jr * ent + ent - 1]);
            mat_evolve(x, br, tauref - t_before, k1, lk);
            k1 = susz_dens_ket(x, dens, br, f1, k1);
            kets_ok = mat_rescale(k1, lk);
            if (kets_ok) {
                susz_store(SX, ent, jr, slot, act, k1, lk, x.lane);
                double tp = tauref;
                for (int j = jr; j < n && kets_ok; ++j) {
                    const double tau = uni(x.ot[j]);
                    if (mat_apply(x, uni(x.osst[j]), uni(x.om[j]), tau - tp, k1, lk) < 0) { kets_ok = false; break; }
                    kets_ok = mat_rescale(k1, lk);
                    if (kets_ok) susz_store(SX, ent, j + 1, slot, act, k1, lk, x.lane);
                    tp = tau;
                }
            }
        }
        // 2b. bras with n_f1 at tauref: slot jr holds the bras AT tauref, slots j < jr those right in front of operator j
        {
            double b1 = act ? SB[(size_t)jr * ent + slot] : 0.0, lb = uni(SB[(size_t)jr * ent + ent - 1]);
            mat_evolve(x, br, t_after - tauref, b1, lb);
            b1 = susz_dens_bra(x, dens, br, f1, b1);
            bras_ok = mat_rescale(b1, lb);
            if (bras_ok) {
                susz_store(SY, ent, jr, slot, act, b1, lb, x.lane);
                double tn = tauref;
                for (int j = jr - 1; j >= 0 && bras_ok; --j) {
                    const double tau = uni(x.ot[j]);
                    susz_bra_step(x, uni(x.osst[j]), uni(x.om[j]), uni(x.osst[j + 1]), tn - tau, b1, lb);
                    bras_ok = mat_rescale(b1, lb);
                    if (bras_ok) susz_store(SY, ent, j, slot, act, b1, lb, x.lane);
                    tn = tau;
                }
            }
        }
        wave_sync();
        // 3. lane = tau point
        for (int tb = 0; tb < ntau; tb += 64) {
            const int t = tb + x.lane;
            if (t < ntau) {
                double tauint = tauref + (double)t * tauD;
                const bool wrapped = tauint > beta;
                if (wrapped) tauint = tauref - (double)(ntau - t) * tauD;
                if (wrapped ? bras_ok : kets_ok) {
                    int lo = 0, hi = n;                                      // j = operators with time < tauint
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (x.ot[mid] < tauint) lo = mid + 1; else hi = mid; }
                    const int j = lo;
                    const double *Kp, *Bp; double tk, tbra;
                    if (!wrapped) { Kp = SX + (size_t)j * ent; tk = j == jr ? tauref : x.ot[j - 1]; Bp = SB + (size_t)j * ent; tbra = j < n ? x.ot[j] : beta; }
                    else { Kp = SK + (size_t)j * ent; tk = j > 0 ? x.ot[j - 1] : 0.0; Bp = SY + (size_t)j * ent; tbra = j == jr ? tauref : x.ot[j]; }
                    const int b = x.osst[j], nst = x.t_n[b], off = x.t_off[b];
                    const double e0 = x.t_eval[off];
                    double dk[NSMT], db[NSMT];
#pragma unroll
                    for (int i = 0; i < NSMT; ++i) {
                        const double de = i < nst ? x.t_eval[off + i] - e0 : 0.0;
                        dk[i] = i < nst ? exp(-de * (tauint - tk)) : 0.0; db[i] = i < nst ? exp(-de * (tbra - tauint)) : 0.0;
                    }
                    const double pref = -cfg * sign0 * exp(Kp[ent - 1] + Bp[ent - 1] - (e0 - C.egs) * (tbra - tk) - ltr0);
                    for (int f2 = 0; f2 < nf; ++f2) {
                        const double *N = dens + (size_t)(b * nf + f2) * nsm * nsm;
                        double val = 0.0;
                        for (int sI = 0; sI < nout; ++sI) {
#pragma unroll
                            for (int i = 0; i < NSMT; ++i) {
                                if (i < nst) {
                                    double r = 0.0;
#pragma unroll
                                    for (int m = 0; m < NSMT; ++m) if (m < nst) r = fma(N[i * nsm + m], Kp[sI * nsm + m] * dk[m], r);
                                    val = fma(Bp[sI * nsm + i] * db[i], r, val);
                                }
                            }
                        }
                        if (val != 0.0) atomicAdd(&out[((size_t)f1 * nf + f2) * ntau + t], pref * val);
                    }
                }
            }
        }
        wave_sync();
    }
}

#ifdef W2DYN_TU_MEAS
// Legendre and tau-bin estimators (MeasGtauRem, CTQMC.F90:2252-2440): THE data-parallel kernel.
// One wave per (bin, flavour).  The wave walks the chains of its bin; lanes take (creator, annihilator) pairs; every
// lane runs the Legendre three-term recursion for its pair and accumulates m*P_l into NLEG register accumulators,
// which are reduced across the wave once per launch and added to the bin's accumulator (no atomics: one owner).
//
// Work list: the (creator, annihilator) pairs of all chains of the bin, flattened (prefix sums of k_w^2 in LDS, binary
// search per element), two elements per lane and pass so that two independent recursions are in flight.
// The Legendre moments sum_e m_e P_l(x_e) (CTQMC.F90:2349-2393, recursion 2371-2388: a multiply and two FMAs per element and
// order with order-dependent coefficients) are accumulated as Chebyshev moments c_j = sum_e m_e T_j(x_e) instead:
// T_{j+1} = 2x T_j - T_{j-1} has constant coefficients, so an element and order cost two FMAs and no constants.  The
// accumulators acc_gleg hold c_j; P_l = sum_j A_lj T_j with A_lj = (2 - delta_j0) g_(l-j)/2 g_(l+j)/2 >= 0, sum_j A_lj = 1
// (g_k = binom(2k, k)/4^k; l - j even), a convex combination, turns them into Legendre moments once per run (k_gleg_legendre).
struct GlegElem { double m, x; int idt; };
struct GlegGroup {                 // per group of up to 64 chains of one (bin, flavour), in LDS
    int2 pk[65];                   // (elements before chain j, its order k); pk[64].x = all elements of the group
    unsigned long long mb[64];     // element offset of the chain's M block
    unsigned long long tb[64];     // element offset of its tau arrays
    double csl[64];                // configuration sign (0 for chains that do not measure)
};

// Element g (clamped into the group's range; bit 31 marks a clamped one) -> packed (chain j, creator row ic, annihilator
// column ia).  Up to 8 chains the chain is found by comparing with the 7 inner prefix sums held in scalar registers, else by
// bisection in LDS.  e / k through the hardware reciprocal: (e + 1/2) / k is at least 1/(2k) >= 2^-9 away from an integer
// and the two roundings shift it by less than k 2^-22.
__device__ __forceinline__ int gleg_locate(const GlegGroup &G, int g, int total, int nstep, const int (&thr)[7])
{
    const int out = g >= total ? (int)0x80000000 : 0;
    g = min(g, total - 1);
    int j = 0;
    if (nstep < 8) {
#pragma unroll
        for (int i = 0; i < 7; ++i) j += g >= thr[i] ? 1 : 0;
    } else
        for (int step = nstep; step > 0; step >>= 1) if (G.pk[j + step].x <= g) j += step;
    const int2 c = G.pk[j];
    const int e = g - c.x;
    const int ic = (int)(((float)e + 0.5f) * __builtin_amdgcn_rcpf((float)c.y)), ia = e - ic * c.y;
    return out | (j << 20) | (ic << 10) | ia;
}
__device__ __forceinline__ double gleg_load_m(const DevConst &C, const GlegGroup &G, int packed)
{
    return C.M[G.mb[(packed >> 20) & 63] + (unsigned)(((packed >> 10) & 1023) * C.kmax + (packed & 1023))];
}

// d / beta, correctly rounded, from the correctly rounded reciprocal (made on the host) and two residual corrections
// (Markstein: the first makes the quotient faithful, the second rounds it correctly); the hardware division sequence
// costs ~40 instructions per element, this 5.  beta's significand is never all ones in practice; the tau-bin index it
// feeds is checked against the oracle per chain.
__device__ __forceinline__ double div_by_beta(double d, double beta, double rbeta)
{
    double q = d * rbeta;
    q = fma(fma(-q, beta, d), rbeta, q);
    return fma(fma(-q, beta, d), rbeta, q);
}

__device__ __forceinline__ void gleg_load_tau(const DevConst &C, const GlegGroup &G, int packed, double &ta, double &tc)
{
    const unsigned long long tb = G.tb[(packed >> 20) & 63];
#ifdef W2DYN_GLEG_NOTAU      // diagnostic: no tau loads
    ta = (double)(packed & 1023) * 0.37 + (double)tb * 1e-9; tc = (double)((packed >> 10) & 1023) * 0.21;
#else
    ta = C.TA[tb + (unsigned)(packed & 1023)]; tc = C.TC[tb + (unsigned)((packed >> 10) & 1023)];
#endif
}

__device__ __forceinline__ GlegElem gleg_finish(const DevConst &C, const GlegGroup &G, int packed, double m, double ta, double tc, double two_over_beta)
{
    GlegElem o;
    double d = ta - tc;
    if (!(d > 0.0)) { d = C.beta + ta - tc; m = -m; }
    o.m = packed < 0 ? 0.0 : m * G.csl[(packed >> 20) & 63];
    const int idt = (int)(div_by_beta(d, C.beta, C.rbeta) * (double)C.ntau) + 1;
    o.idt = (idt > C.ntau ? C.ntau : idt) - 1;
    o.x = two_over_beta * d - 1.0;
    return o;
}

// Work list of k_meas_gleg: a segment is the element list (chain, ic, ia) of one (accumulator bin, flavour), numbered
// bin-major; segpre[sg] = number of elements before segment sg, segpre[nbins nflav] = all of them.  A workgroup of 1024
// threads takes GLEG_SEG_BLOCK / nflav bins: the orders K[chain][flavour] of its chains are one contiguous array, read with
// unit stride eight loads at a time and squared into LDS counters, which are then scanned (four consecutive segments per
// thread).  With more than one workgroup the last one to finish (ticket counter, reset for the next launch) adds the
// workgroup offsets.
__global__ __launch_bounds__(1024) void k_gleg_segments(DevConst C, unsigned long long *segpre, unsigned long long *blocktot, unsigned *ticket)
{
    __shared__ unsigned cnt[GLEG_SEG_BLOCK];
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned last;
    const int nf = C.nflav, wpb = C.walkers_per_bin, nseg = C.nbins * nf, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bpb = GLEG_SEG_BLOCK / nf;                       // bins per workgroup
    const int bin0 = blockIdx.x * bpb, bin1 = min(bin0 + bpb, C.nbins), seg0 = bin0 * nf, nloc = (bin1 - bin0) * nf;
    for (int i = tid; i < GLEG_SEG_BLOCK; i += 1024) cnt[i] = 0;
    __syncthreads();
    const int p0 = bin0 * wpb * nf, p1 = min(bin1 * wpb, C.nwalkers) * nf;       // chains x flavours < 2^31 (checked on the host)
    for (int pb = p0 + tid; pb < p1; pb += 8 * 1024) {
        int kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int p = pb + u * 1024; kv[u] = p < p1 ? C.K[p] : 0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = pb + u * 1024;
            if (kv[u] > 0) { const int w = p / nf, f = p - w * nf; atomicAdd(&cnt[(w / wpb - bin0) * nf + f], (unsigned)(kv[u] * kv[u])); }
        }
    }
    __syncthreads();
    unsigned long long c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = 4 * tid + i < nloc ? cnt[4 * tid + i] : 0u;
    const unsigned long long mine = c[0] + c[1] + c[2] + c[3];
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long run = incl - mine;
    for (int i = 0; i < wv; ++i) run += wsum[i];
    if (blockIdx.x == 0 && tid == 0) segpre[0] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { run += c[i]; if (4 * tid + i < nloc) segpre[seg0 + 4 * tid + i + 1] = run; }     // inclusive, within the workgroup
    if (gridDim.x == 1) return;
    if (tid == 1023) blocktot[blockIdx.x] = run;
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    unsigned long long off = 0;
    for (unsigned b = 1; b < gridDim.x; ++b) {
        off += __builtin_nontemporal_load(&blocktot[b - 1]);
        const int sb = (int)b * bpb * nf, sn = (min((int)(b + 1) * bpb, C.nbins) - (int)b * bpb) * nf;
        for (int i = tid; i < sn; i += 1024) segpre[sb + i + 1] = __builtin_nontemporal_load(&segpre[sb + i + 1]) + off;
    }
    if (tid == 0) *ticket = 0;
    (void)nseg;
}

// One wave per workgroup, as many workgroups as the GPU holds at once (two per SIMD: every lane keeps one accumulator per
// order in registers, 2 NLEG VGPRs).  The global element list is cut into equal ranges (multiples of 128), one per wave, so
// that all waves finish together whatever the expansion orders of the chains are; a wave walks the segments its range
// touches, takes each one's elements in passes of 128 (two per lane, two independent recursions in flight; a pass costs
// 2 fp64 FMAs per element and order, see above), and at the end of a segment sums the lane accumulators through an LDS
// transpose and adds them and the segment's tau histogram (LDS) to the bin's accumulators with return-less atomics (a
// segment cut by a range boundary has two owners; and nobody waits for the memory round trip).  Everything per element outside the recursion is kept
// branch-free and short (~60 instructions): it is paid 128 times per pass against 400 instructions of recursion.
template <int NLEG_TILE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NLEG_TILE > 100 ? 1 : 2, 8))) void k_meas_gleg(DevConst C, int lds_hist, const unsigned long long *__restrict__ segpre)
{
    extern __shared__ double hist[];            // [ntau if lds_hist] tau-bin histogram of the current segment, then the reduction tile
    __shared__ GlegGroup G;
    const int lane0 = threadIdx.x, nleg = C.nleg, ntau = C.ntau, nseg = C.nbins * C.nflav;
    const unsigned long long all = segpre[nseg];
    unsigned long long chunk = (all + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 127) / 128 * 128;
    // the 64-bit division runs on the vector unit: bring the (uniform) result back to scalar registers
    chunk = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(chunk >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)chunk);
    const unsigned long long g0 = (unsigned long long)blockIdx.x * chunk, g1 = min(g0 + chunk, all);
    if (g0 >= g1) return;
    int sg = 0;                                 // the segment that holds element g0: last one with segpre[sg] <= g0
    for (int step = 1 << 30; step > 0; step >>= 1) if (sg + step < nseg && segpre[sg + step] <= g0) sg += step;
    double *red = hist + (lds_hist ? ntau : 0);
    if (lds_hist) for (int i = lane0; i < ntau; i += 64) hist[i] = 0.0;
    const double two_over_beta = 2.0 / C.beta;
    for (; sg < nseg; ++sg) {
        const unsigned long long sb = segpre[sg], se = segpre[sg + 1];
        if (sb >= g1) break;
        if (se == sb) continue;
        const int lo = (int)(max(g0, sb) - sb), hi = (int)(min(g1, se) - sb);      // this wave's part of the segment
        int lane = lane0;
        asm volatile("" : "+v"(lane));      // per-segment values derived from the lane index are recomputed, not kept (and spilled) across the pass loop
        const int bin = sg / C.nflav, f = sg - bin * C.nflav;
        double acc[NLEG_TILE];
#pragma unroll
        for (int l = 0; l < NLEG_TILE; ++l) acc[l] = 0.0;
        double *gt = C.acc_gtau + ((size_t)bin * C.nflav + f) * ntau;
        const int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
        int goff = 0;                           // elements of the segment before the current group of 64 chains
        for (int wc = w0; wc < w1 && goff < hi; wc += 64) {
            wave_sync();
            const size_t wf = (size_t)(wc + lane) * C.nflav + f;
            const int km = (wc + lane < w1) ? C.K[wf] : 0, nch = min(w1 - wc, 64);
            int incl = km * km;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            G.pk[lane] = make_int2(incl - km * km, km);
            if (lane == 63) G.pk[64] = make_int2(incl, 0);
            G.mb[lane] = wf * C.kmax * C.kmax; G.tb[lane] = wf * C.kmax;
            G.csl[lane] = (wc + lane < w1) ? C.cfgsign[wc + lane] : 0.0;
            int thr[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) thr[i] = __builtin_amdgcn_readlane(incl, i);
            wave_sync();
            const int gtot = __builtin_amdgcn_readlane(incl, 63);
            const int total = min(gtot, hi - goff);           // local element range [base0, total) of this group
            int base = max(lo - goff, 0);
            goff += gtot;
            if (base >= total) continue;
            int nstep = 32;
            while (nstep >= nch && nstep > 0) nstep >>= 1;    // chains beyond nch have no elements: the prefix is flat there
            // M (HBM, a gather) is requested two passes ahead; the tau values of a pass (reused by k elements each: cache
            // hits) are requested before the pass's M gather (loads complete in order: waiting for them also waits for
            // the M request of the pass before, not for the one just made)
            int iA0 = gleg_locate(G, base + lane, total, nstep, thr), iA1 = gleg_locate(G, base + 64 + lane, total, nstep, thr);
            double mA0 = gleg_load_m(C, G, iA0), mA1 = gleg_load_m(C, G, iA1);
            int iB0 = gleg_locate(G, base + 128 + lane, total, nstep, thr), iB1 = gleg_locate(G, base + 192 + lane, total, nstep, thr);
            double mB0 = gleg_load_m(C, G, iB0), mB1 = gleg_load_m(C, G, iB1);
            for (; base < total; base += 128) {
                double ta0, tc0, ta1, tc1;          // requested before the next M gather: loads complete in order
                gleg_load_tau(C, G, iA0, ta0, tc0); gleg_load_tau(C, G, iA1, ta1, tc1);
                const int iC0 = gleg_locate(G, base + 256 + lane, total, nstep, thr), iC1 = gleg_locate(G, base + 320 + lane, total, nstep, thr);
                const double mC0 = gleg_load_m(C, G, iC0), mC1 = gleg_load_m(C, G, iC1);
                const GlegElem e0 = gleg_finish(C, G, iA0, mA0, ta0, tc0, two_over_beta), e1 = gleg_finish(C, G, iA1, mA1, ta1, tc1, two_over_beta);
#ifndef W2DYN_GLEG_NOHIST
                if (iA0 >= 0) { if (lds_hist) atomicAdd(&hist[e0.idt], e0.m); else atomicAdd(&gt[e0.idt], e0.m); }
                if (iA1 >= 0) { if (lds_hist) atomicAdd(&hist[e1.idt], e1.m); else atomicAdd(&gt[e1.idt], e1.m); }
#endif
                iA0 = iB0; iA1 = iB1; mA0 = mB0; mA1 = mB1; iB0 = iC0; iB1 = iC1; mB0 = mC0; mB1 = mC1;
                // T_{l-1}, T_l of the two elements; T_{l+1} = 2x T_l - T_{l-1}: one FMA for the polynomial, one for the moment
                double p0 = 1.0, q0 = e0.x, p1 = 1.0, q1 = e1.x;
                const double x0 = e0.x + e0.x, x1 = e1.x + e1.x;
                acc[0] += e0.m; acc[0] += e1.m;
                if (NLEG_TILE > 1) { acc[1] = fma(e0.m, q0, acc[1]); acc[1] = fma(e1.m, q1, acc[1]); }
#ifndef W2DYN_GLEG_NOREC      // diagnostic: everything but the recursion
#pragma unroll
#else
#pragma unroll
                for (int il = 2; il < NLEG_TILE; ++il) acc[il] += x0 + x1;
                if (0)
#endif
                for (int il = 2; il < NLEG_TILE; ++il) {               // orders >= nleg are computed but never stored
                    const double n0 = fma(x0, q0, -p0), n1 = fma(x1, q1, -p1);
                    acc[il] = fma(e0.m, n0, acc[il]); acc[il] = fma(e1.m, n1, acc[il]);
                    p0 = q0; q0 = n0; p1 = q1; q1 = n1;
                }
            }
        }
        // cross-lane reduction of the NLEG accumulators through an LDS transpose, 8 orders at a time: lane = (order, eighth)
        // sums 8 lanes' partials, three DPP steps finish the sum.  (100 separate wave reductions cost ~6000 instr.)
        double *gl = C.acc_gleg + ((size_t)bin * C.nflav + f) * nleg;
#pragma unroll
        for (int c0 = 0; c0 < NLEG_TILE; c0 += GLEG_RED_ROWS) {
            wave_sync();
#pragma unroll
            for (int i = 0; i < GLEG_RED_ROWS; ++i)
                if (c0 + i < NLEG_TILE) red[i * GLEG_RED_STRIDE + lane] = acc[c0 + i];
            wave_sync();
            const int lo8 = lane >> 3, part = lane & 7;
            double sres = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) sres += red[lo8 * GLEG_RED_STRIDE + part * 8 + j];
            sres += dpp_d<0xB1>(sres);          // quad_perm [1,0,3,2]
            sres += dpp_d<0x4E>(sres);          // quad_perm [2,3,0,1]
            sres += dpp_d<0x141>(sres);         // row_half_mirror: lanes 0-3 <-> 7-4 of every 8
            const int l = c0 + lo8;
            // atomics without return value even where the segment has one owner: a plain += waits a memory round trip
            // per tile (and per 64 tau bins below) with nothing else to do
            if (part == 0 && l < nleg && l < NLEG_TILE) atomicAdd(&gl[l], sres);
        }
        if (lds_hist) {
            wave_sync();
            for (int i = lane; i < ntau; i += 64) {
                const double h = hist[i];
                if (h != 0.0) { hist[i] = 0.0; atomicAdd(&gt[i], h); }
            }
        }
    }
}

// Chebyshev moments of all (bin, flavour) -> Legendre moments: out[r][l] = sum_{j <= l} tab[l][j] in[r][j]
__global__ __launch_bounds__(256) void k_gleg_legendre(const double *__restrict__ in, const double *__restrict__ tab, int nleg, size_t nrows, double *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * nleg) return;
    const size_t r = i / nleg; const int l = (int)(i - r * nleg);
    double a = 0.0;
    for (int j = l & 1; j <= l; j += 2) a = fma(tab[(size_t)l * nleg + j], in[r * nleg + j], a);
    out[i] = a;
}

// AccumGtau = 2: autocorrelation accumulator of the G(tau) measurement series (autocorracc_add, Accumulators.F90:283-331),
// kept per chain like the reference keeps it per rank.  One workgroup per (chain, flavour): this measurement's G(tau)
// histogram is built in LDS, then thread i folds bin i into the chain's stack of (spill, mean, sqdiff) levels - Welford
// update of level 0; whenever the low bits of the measurement count n are ones, half the spill of the level below rolls
// over into the next level.  Streaming read-modify-write of dat[chain][flavour][level][spill|mean|sqdiff][Ntau] (HBM-bound:
// 48 B per bin for level 0, 96 B amortised over all levels).
__global__ __launch_bounds__(256) void k_meas_gtau_ac(DevConst C, double *dat, int nlev, long long n)
{
    extern __shared__ double hist[];            // [ntau]
    const int w = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, ntau = C.ntau, kmax = C.kmax;
    if (C.ws[w].sector > 1) return;             // chains in worm space measure no Z-space estimator
    for (int i = tid; i < ntau; i += 256) hist[i] = 0.0;
    __syncthreads();
    const size_t wf = (size_t)w * C.nflav + f;
    const int k = C.K[wf];
    const double cs = C.cfgsign[w];
    for (int e = tid; e < k * k; e += 256) {
        const int ic = e / k, ia = e - ic * k;
        double m = C.M[wf * kmax * kmax + (size_t)ic * kmax + ia], d = C.TA[wf * kmax + ia] - C.TC[wf * kmax + ic];
        if (!(d > 0.0)) { d = C.beta + C.TA[wf * kmax + ia] - C.TC[wf * kmax + ic]; m = -m; }
        int idt = (int)(d / C.beta * (double)ntau) + 1;
        if (idt > ntau) idt = ntau;
        atomicAdd(&hist[idt - 1], m * cs);
    }
    __syncthreads();
    double *base = dat + wf * (size_t)nlev * 3 * ntau;
    // two tau bins per thread as 16-byte accesses when the rows are 16-byte aligned (Ntau even), one otherwise
    if ((ntau & 1) == 0) {
        for (int i = 2 * tid; i < ntau; i += 512) {
            double2 val = make_double2(hist[i], hist[i + 1]);
            long long nn = n;
            for (int lvl = 0;; ) {
                double2 *sp = (double2 *)(base + ((size_t)lvl * 3) * ntau + i), *mn = (double2 *)((double *)sp + ntau), *sq = (double2 *)((double *)mn + ntau);
                double2 s_ = *sp, m_ = *mn, q_ = *sq;
                const double rn = (double)(nn + 1);
                const double i1x = val.x - m_.x, i1y = val.y - m_.y;
                s_.x += val.x; s_.y += val.y;
                m_.x += i1x / rn; m_.y += i1y / rn;
                q_.x += i1x * (val.x - m_.x); q_.y += i1y * (val.y - m_.y);
                const bool roll = (nn & 1) && lvl != nlev - 1;
                val = make_double2(s_.x / 2, s_.y / 2);
                if (roll) s_ = make_double2(0.0, 0.0);
                *sp = s_; *mn = m_; *sq = q_;
                if (!roll) break;
                nn >>= 1; ++lvl;
            }
        }
        return;
    }
    for (int i = tid; i < ntau; i += 256) {
        double val = hist[i];
        long long nn = n;
        for (int lvl = 0;; ) {
            double *sp = base + ((size_t)lvl * 3) * ntau + i, *mn = sp + ntau, *sq = mn + ntau;
            *sp += val;
            const double incr1 = val - *mn;
            const double mean = *mn + incr1 / (double)(nn + 1);
            *mn = mean;
            *sq += incr1 * (val - mean);
            if (!(nn & 1) || lvl == nlev - 1) break;
            nn >>= 1; ++lvl;
            val = *sp / 2; *sp = 0.0;
        }
    }
}

// autocorracc_blocks (Accumulators.F90:343-361) averaged over the chains of an accumulator bin:
// out[bin][flavour][level][Ntau] = mean over chains of sqdiff / (n_level - 1), n halved from level to level
__global__ __launch_bounds__(256) void k_gtau_blocks(DevConst C, const double *dat, int nlev, long long n, double *out)
{
    const int bin = blockIdx.x, f = blockIdx.y / nlev, lvl = blockIdx.y % nlev, ntau = C.ntau;
    long long nl = n >> lvl;
    const double den = nl > 1 ? (double)(nl - 1) : 0.0;
    const int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
    for (int i = threadIdx.x; i < ntau; i += 256) {
        double a = 0.0;
        for (int w = w0; w < w1; ++w) a += dat[((((size_t)w * C.nflav + f) * nlev + lvl) * 3 + 2) * ntau + i] / den;
        out[(((size_t)bin * C.nflav + f) * nlev + lvl) * ntau + i] = a / (double)(w1 - w0);
    }
}

// MeasGtau4PntLeg (CTQMC.F90:2609-2763), FourPnt bits 1 (imaginary-time bins, g4tau) and 2 (mixed Legendre / Matsubara
// representation, g4leg): one workgroup per (chain, flavour pair), threads stride over the k1^2 k2^2 quadruples
// (a1, c1, c2, a2): M = M1(c1,a1) M2(c2,a2) - [f1 == f2] M(c2,a1) M(c1,a2) with the fermionic signs of tau12 and tau34;
// no configuration sign.  First version: every quadruple adds its N4leg^2 (2 N4iwb + 1) terms with atomics - the direct
// term factorises into GEMM-shaped contractions, which is the next step for production sizes.
__global__ __launch_bounds__(256) void k_meas_g4leg(DevConst C, double *acc_tau, double *acc_leg, int nbl, int nt, int nl, int nwb)
{
    const int w = blockIdx.x, nf = C.nflav, f1 = blockIdx.y / nf, f2 = blockIdx.y - f1 * nf, kmax = C.kmax;
    if (C.ws[w].sector > 1) return;
    const int bin = (w / C.walkers_per_bin) % nbl, nw = 2 * nwb + 1;
    const size_t w1 = (size_t)w * nf + f1, w2 = (size_t)w * nf + f2;
    const int k1 = C.K[w1], k2 = C.K[w2];
    const double *M1 = C.M + w1 * kmax * kmax, *M2 = C.M + w2 * kmax * kmax;
    const double *tc1 = C.TC + w1 * kmax, *ta1 = C.TA + w1 * kmax, *tc2 = C.TC + w2 * kmax, *ta2 = C.TA + w2 * kmax;
    const double beta = C.beta;
    double *gt = acc_tau ? acc_tau + ((size_t)bin * nf * nf + blockIdx.y) * nt * nt * nt : nullptr;
    double *gl = acc_leg ? acc_leg + ((size_t)bin * nf * nf + blockIdx.y) * nl * nl * nw * 2 : nullptr;
    const long long total = (long long)k1 * k1 * k2 * k2;
    for (long long q = threadIdx.x; q < total; q += 256) {
        const int a2 = (int)(q % k2), c2 = (int)((q / k2) % k2), c1 = (int)((q / ((long long)k2 * k2)) % k1), a1 = (int)(q / ((long long)k2 * k2 * k1));
        double M = M1[(size_t)c1 * kmax + a1] * M2[(size_t)c2 * kmax + a2];
        if (f1 == f2) M -= M1[(size_t)c2 * kmax + a1] * M2[(size_t)c1 * kmax + a2];
        double tau12 = ta1[a1] - tc1[c1], tau14 = ta1[a1] - tc2[c2], tau34 = ta2[a2] - tc2[c2];
        double sg = 1.0;
        if (tau12 < 0.0) { tau12 += beta; sg = -sg; }
        if (tau14 < 0.0) tau14 += beta;
        if (tau34 < 0.0) { tau34 += beta; sg = -sg; }
        M = -sg * M;
        if (gt) {
            const int b12 = (int)(nt * tau12 / beta), b14 = (int)(nt * tau14 / beta), b34 = (int)(nt * tau34 / beta);
            atomicAdd(&gt[((size_t)b12 * nt + b34) * nt + b14], M);
        }
        if (gl) {
            const double x12 = 2.0 / beta * tau12 - 1.0, x34 = 2.0 / beta * tau34 - 1.0;
            double s1, c1p; sincos(2.0 * M_PI * tau14 / beta, &s1, &c1p);
            double p12m = 0.0, p12 = 1.0;                           // P_{l-1}, P_l at x12 (Bonnet's recursion, LegendrePoly.F90:36-44)
            for (int l = 0; l < nl; ++l) {
                if (l == 1) { p12m = 1.0; p12 = x12; }
                else if (l > 1) { const double pn = ((2 * l - 1) * x12 * p12 - (l - 1) * p12m) / l; p12m = p12; p12 = pn; }
                double p34m = 0.0, p34 = 1.0;
                for (int lp = 0; lp < nl; ++lp) {
                    if (lp == 1) { p34m = 1.0; p34 = x34; }
                    else if (lp > 1) { const double pn = ((2 * lp - 1) * x34 * p34 - (lp - 1) * p34m) / lp; p34m = p34; p34 = pn; }
                    const double v = p34 * M * p12;
                    double *g = gl + ((size_t)l * nl + lp) * nw * 2;
                    double cr = 1.0, ci = 0.0;                       // exp(i m phi), m = 0, +-1, ...
                    atomicAdd(&g[2 * nwb], v);
                    for (int m = 1; m <= nwb; ++m) {
                        const double c2r = cr * c1p - ci * s1; ci = cr * s1 + ci * c1p; cr = c2r;
                        atomicAdd(&g[2 * (nwb + m)], v * cr); atomicAdd(&g[2 * (nwb + m) + 1], v * ci);
                        atomicAdd(&g[2 * (nwb - m)], v * cr); atomicAdd(&g[2 * (nwb - m) + 1], -v * ci);
                    }
                }
            }
        }
    }
}

// g4leg without per-term atomics.  For a fixed annihilator a1 (flavour f1) and creator c2 (flavour f2) the phase
// exp(i w_m tau14) is a common factor and the sum over (c1, a2) splits:
//   S(l,l') = -[ A(l) B(l') - [f1 == f2] M1(c2,a1) E(l,l') ],   A(l) = sum_c1 s12 P_l(x12) M1(c1,a1),
//   B(l') = sum_a2 s34 P_l'(x34) M2(c2,a2),   E(l,l') = sum_c1 s12 P_l(x12) T(c1,l'),  T(c1,l') = sum_a2 M2(c1,a2) s34 P_l'(x34)
// (s12, s34: signs of tau12, tau34).  One workgroup per (chain, flavour pair); thread = NPT pairs (l, l'), each with its N4iwb + 1
// non-negative frequencies in registers (S is real: the negative ones are the conjugates), Legendre tables, the
// phases and column a1 of M1 of the current (a1, c2) in LDS (prepared by different waves), one atomic flush per
// workgroup.
template <int NWB1, int NPT>
__global__ __launch_bounds__(256) void k_meas_g4leg_fact(DevConst C, double *acc_leg, int nbl, int nl, int nwb)
{
    extern __shared__ double sm[];
    const int w = blockIdx.x, nf = C.nflav, f1 = blockIdx.y / nf, f2 = blockIdx.y - f1 * nf, kmax = C.kmax, tid = threadIdx.x;
    if (C.ws[w].sector > 1) return;
    const int bin = (w / C.walkers_per_bin) % nbl, nw = 2 * nwb + 1;
    const size_t w1 = (size_t)w * nf + f1, w2 = (size_t)w * nf + f2;
    const int k1 = C.K[w1], k2 = C.K[w2];
    if (k1 == 0 || k2 == 0) return;
    const double *M1 = C.M + w1 * kmax * kmax, *M2 = C.M + w2 * kmax * kmax;
    const double *tc1 = C.TC + w1 * kmax, *ta1 = C.TA + w1 * kmax, *tc2 = C.TC + w2 * kmax, *ta2 = C.TA + w2 * kmax;
    const double beta = C.beta;
    double *L12 = sm, *L34 = L12 + (size_t)k1 * nl, *T = L34 + (size_t)k2 * nl, *B = T + (size_t)k1 * nl;
    double *CA = B + nl, *CB = CA + nl, *Mc = CB + nl, *PH = Mc + k1 + 1;   // PH: 2 * NWB1 doubles
    double *gl = acc_leg + ((size_t)bin * nf * nf + blockIdx.y) * nl * nl * nw * 2;
    // Bonnet recursion with its coefficients tabulated: P_l = CA(l) x P_{l-1} - CB(l) P_{l-2}
    for (int q = tid; q < nl; q += 256) { CA[q] = q ? (double)(2 * q - 1) / q : 0.0; CB[q] = q ? (double)(q - 1) / q : 0.0; }
    auto legendre_row = [&](double x, double sgn, double *row) {
        double pm = 0.0, p = 1.0;
        for (int q = 0; q < nl; ++q) {
            if (q == 1) { pm = 1.0; p = x; }
            else if (q > 1) { const double pn = CA[q] * x * p - CB[q] * pm; pm = p; p = pn; }
            row[q] = sgn * p;
        }
    };
    for (int p0 = 0; p0 < nl * nl; p0 += 256 * NPT) {            // tiles of 256 * NPT (l, l') pairs, NPT per thread
        bool own[NPT]; int l[NPT], lp[NPT];
        double re[NPT][NWB1], im[NPT][NWB1];
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int pr = p0 + u * 256 + tid;
            own[u] = pr < nl * nl; l[u] = own[u] ? pr / nl : 0; lp[u] = own[u] ? pr - l[u] * nl : 0;
#pragma unroll
            for (int m = 0; m < NWB1; ++m) { re[u][m] = 0.0; im[u][m] = 0.0; }
        }
        for (int c2 = 0; c2 < k2; ++c2) {
            __syncthreads();
            for (int a2 = tid; a2 < k2; a2 += 256) {
                double t = ta2[a2] - tc2[c2], sg = 1.0;
                if (t < 0.0) { t += beta; sg = -1.0; }
                legendre_row(2.0 / beta * t - 1.0, sg, L34 + (size_t)a2 * nl);
            }
            __syncthreads();
            if (tid < nl) { double b = 0.0; for (int a2 = 0; a2 < k2; ++a2) b += L34[(size_t)a2 * nl + tid] * M2[(size_t)c2 * kmax + a2]; B[tid] = b; }
            if (f1 == f2)                                        // T(c1, l') = sum_a2 M2(c1, a2) s34 P_l'(x34): once per c2
                for (int e = tid; e < k1 * nl; e += 256) {
                    const int c1 = e / nl, q = e - c1 * nl;
                    double t = 0.0;
                    for (int a2 = 0; a2 < k2; ++a2) t += M2[(size_t)c1 * kmax + a2] * L34[(size_t)a2 * nl + q];
                    T[e] = t;
                }
            for (int a1 = 0; a1 < k1; ++a1) {
                __syncthreads();
                // wave 0: Legendre rows of tau12; wave 1: the phases exp(i w_m tau14); waves 2, 3: column a1 of M1
                if (tid < 64) {
                    for (int c1 = tid; c1 < k1; c1 += 64) {
                        double t = ta1[a1] - tc1[c1], sg = 1.0;
                        if (t < 0.0) { t += beta; sg = -1.0; }
                        legendre_row(2.0 / beta * t - 1.0, sg, L12 + (size_t)c1 * nl);
                    }
                } else if (tid < 128) {
                    const int m = tid - 64;
                    if (m <= nwb) {
                        double t14 = ta1[a1] - tc2[c2];
                        if (t14 < 0.0) t14 += beta;
                        double sn, cs; sincos(2.0 * M_PI * (double)m * t14 / beta, &sn, &cs);
                        PH[2 * m] = cs; PH[2 * m + 1] = sn;
                    }
                } else {
                    for (int c1 = tid - 128; c1 < k1; c1 += 128) Mc[c1] = M1[(size_t)c1 * kmax + a1];
                    if (tid == 255) Mc[k1] = M1[(size_t)c2 * kmax + a1];
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    if (!own[u]) continue;
                    double Aa = 0.0, E = 0.0;
                    if (f1 == f2)
                        for (int c1 = 0; c1 < k1; ++c1) { const double p = L12[(size_t)c1 * nl + l[u]]; Aa += p * Mc[c1]; E += p * T[(size_t)c1 * nl + lp[u]]; }
                    else
                        for (int c1 = 0; c1 < k1; ++c1) Aa += L12[(size_t)c1 * nl + l[u]] * Mc[c1];
                    const double S = Mc[k1] * E - Aa * B[lp[u]];
#pragma unroll
                    for (int m = 0; m < NWB1; ++m)
                        if (m <= nwb) { re[u][m] += S * PH[2 * m]; im[u][m] += S * PH[2 * m + 1]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            if (!own[u]) continue;
            double *g = gl + ((size_t)l[u] * nl + lp[u]) * nw * 2;
#pragma unroll
            for (int m = 0; m < NWB1; ++m) {
                if (m <= nwb) {
                    atomicAdd(&g[2 * (nwb + m)], re[u][m]);
                    if (m > 0) { atomicAdd(&g[2 * (nwb + m) + 1], im[u][m]); atomicAdd(&g[2 * (nwb - m)], re[u][m]); atomicAdd(&g[2 * (nwb - m) + 1], -im[u][m]); }
                }
            }
        }
    }
}

// Matsubara estimator (measure_giw_nfft + naive get_giw, CTQMC.F90:1015-1056, Nfft_z.F90:376-414):
// giw(n) += sum_ij M(c_j,a_i) exp(-i w_n tau_c_j) exp(+i w_n tau_a_i), no configuration sign, w_n = (2n+1) pi/beta.
// M is real, so giw(-w) = conj(giw(w)): only the Niw positive frequencies are computed, the mirror entry is written with
// them.  Two-step transform  giw(n) = sum_c conj(T_c(n)) U_c(n),  U_c(n) = sum_a M(c,a) T_a(n),  T_o(n) = e^{i w_n tau_o}:
// a complex x real multiply-add (two FMAs) per element and frequency, a complex multiply-add per creator and frequency.
// One workgroup per (accumulator bin, flavour, block of 256 * GIW_NP frequencies), thread = frequency (GIW_NP of them,
// 256 apart), the bin's chains looped; everything about a chain is workgroup-uniform: the row of M comes through scalar
// loads, the FMAs take it from SGPRs, and the phases T_a(n) of GIW_TS annihilators stay in registers while the creators
// are looped.  No sincos per element: with n = nbase + 256 p + 16 h + l the phase is the product of three table entries,
// P3[p] = e^{i pi (2 (nbase + 256 p) + 1) x}, P2[h] = e^{2 pi i 16 h x}, P1[l] = e^{2 pi i l x}, x = tau/beta (sincospi:
// the argument reduction is exact), 16 + 16 + GIW_NP sincos per operator and chain instead of 2 k per element.
constexpr int GIW_TS = 24, GIW_NP = 8, GIW_TAB = 32 + GIW_NP;
__device__ __forceinline__ void giw_fill(double2 *tab, int nops, const double *tau, double rbeta, int nbase, int tid)
{
    for (int e = tid; e < nops * GIW_TAB; e += 256) {
        const int o = e / GIW_TAB, j = e - o * GIW_TAB;
        const double x = tau[o] * rbeta;
        const double arg = j < 16 ? (double)(2 * j) * x : (j < 32 ? (double)(32 * (j - 16)) * x : (double)(2 * (nbase + 256 * (j - 32)) + 1) * x);
        double sn, cs; sincospi(arg, &sn, &cs);
        tab[e] = make_double2(cs, sn);
    }
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// one frequency pass over one annihilator tile of 4 NG entries (na of them exist) and nc creators.  The row segments of M
// are read unconditionally, 4 NG doubles as aligned scalar loads: entries beyond na - finite stale numbers of the
// zero-initialised arrays, at the far end up to 184 B past the last row into the allocation's padding (dev_alloc) - meet
// phases that are set to zero
template <int NG>
__device__ __forceinline__ void giw_pass(const double2 *tA, const double2 *tC, const double *Mrow0, int kmax, int na, int nc, int p, int h, int l, double &sr, double &si)
{
    double er[4 * NG], ei[4 * NG];
#pragma unroll
    for (int a = 0; a < 4 * NG; ++a) {
        const int ac = a < na ? a : 0;
        const double2 e = cmul(tA[ac * GIW_TAB + 32 + p], cmul(tA[ac * GIW_TAB + 16 + h], tA[ac * GIW_TAB + l]));
        er[a] = a < na ? e.x : 0.0; ei[a] = a < na ? e.y : 0.0;
    }
    for (int c = 0; c < nc; ++c) {
        const double4 *row = (const double4 *)(Mrow0 + (size_t)c * kmax);
        double ur = 0.0, ui = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const double4 m = row[g];
            ur = fma(m.x, er[4 * g], ur); ui = fma(m.x, ei[4 * g], ui);
            ur = fma(m.y, er[4 * g + 1], ur); ui = fma(m.y, ei[4 * g + 1], ui);
            ur = fma(m.z, er[4 * g + 2], ur); ui = fma(m.z, ei[4 * g + 2], ui);
            ur = fma(m.w, er[4 * g + 3], ur); ui = fma(m.w, ei[4 * g + 3], ui);
        }
        const double2 t = cmul(tC[c * GIW_TAB + 32 + p], cmul(tC[c * GIW_TAB + 16 + h], tC[c * GIW_TAB + l]));
        sr += t.x * ur + t.y * ui; si += t.x * ui - t.y * ur;       // conj(T_c) U_c
    }
}
__global__ __launch_bounds__(256) void k_meas_giw(DevConst C, const double *__restrict__ Mall, const double *__restrict__ TCall,
                                                  const double *__restrict__ TAall, const int *__restrict__ Kall, int ccap)
{
    extern __shared__ double2 giw_lds[];
    double2 *tA = giw_lds, *tC = giw_lds + GIW_TS * GIW_TAB;        // phase tables of the current annihilator tile / creator chunk
    const int bin = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, l = tid & 15, h = tid >> 4;
    const int niw = C.niw2 >> 1, nbase = blockIdx.z * 256 * GIW_NP, kmax = C.kmax, nf = C.nflav;
    double are[GIW_NP], aim[GIW_NP];
#pragma unroll
    for (int p = 0; p < GIW_NP; ++p) { are[p] = 0.0; aim[p] = 0.0; }
    const int w0 = bin * C.walkers_per_bin, w1 = min(w0 + C.walkers_per_bin, C.nwalkers);
    for (int w = w0; w < w1; ++w) {
        const int k = Kall[(size_t)w * nf + f];
        if (k == 0) continue;
        const double *tcp = TCall + ((size_t)w * nf + f) * kmax, *tap = TAall + ((size_t)w * nf + f) * kmax;
        const double *Mf = Mall + ((size_t)w * nf + f) * kmax * kmax;
        for (int c0 = 0; c0 < k; c0 += ccap) {
            const int nc = min(ccap, k - c0);
            for (int a0 = 0; a0 < k; a0 += GIW_TS) {
                const int na = min(GIW_TS, k - a0);
                __syncthreads();                                     // the tables' readers of the previous round are done
                if (a0 == 0) giw_fill(tC, nc, tcp + c0, C.rbeta, nbase, tid);
                giw_fill(tA, na, tap + a0, C.rbeta, nbase, tid);
                __syncthreads();
                const double *Mrow0 = Mf + (size_t)c0 * kmax + a0;
#pragma unroll
                for (int p = 0; p < GIW_NP; ++p) {
                    if (nbase + 256 * p >= niw) continue;            // uniform
                    double sr = 0.0, si = 0.0;
                    if (na <= 8) giw_pass<2>(tA, tC, Mrow0, kmax, na, nc, p, h, l, sr, si);
                    else if (na <= 16) giw_pass<4>(tA, tC, Mrow0, kmax, na, nc, p, h, l, sr, si);
                    else giw_pass<GIW_TS / 4>(tA, tC, Mrow0, kmax, na, nc, p, h, l, sr, si);
                    are[p] += sr; aim[p] += si;
                }
            }
        }
    }
    double *g = C.acc_giw + ((size_t)bin * nf + f) * C.niw2 * 2;
#pragma unroll
    for (int p = 0; p < GIW_NP; ++p) {
        const int n = nbase + 256 * p + tid;
        if (n < niw) {
            double *gp = g + (size_t)(niw + n) * 2, *gm = g + (size_t)(niw - 1 - n) * 2;
            gp[0] += are[p]; gp[1] += aim[p]; gm[0] += are[p]; gm[1] -= aim[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Z-space two-particle Matsubara estimators (measure_g4iw_nfft with the exact-sum transform, CTQMC.F90:2130-2234,
// Nfft_z.F90:416-453)
// ---------------------------------------------------------------------------------------------------------------

// g2(wa, wc) = sum_{ic,ia} exp(-i nu_wc tau_c(ic)) M(ic,ia) exp(+i nu_wa tau_a(ia)) for one chain and flavour, as
// U(ic,wa) = sum_ia M(ic,ia) Ea(ia,wa) followed by g2(wa,wc) = sum_ic Ec(ic,wc) U(ic,wa): 8 k n2 (k + n2) flop
// instead of 8 k^2 n2^2.  grid (chains, flavours), 256 threads; LDS 3 * kmax * G2_FC complex.
// G2 layout [chain][flavour][wa][wc], frequencies nu_n = (2n+1) pi/beta, n = -n2/2 .. n2/2-1.
// Behind the n2 x n2 table of a (chain, flavour) lie its 2 nwb + 1 central diagonals once more, as rows:
// diag[(a - c + nwb) n2 + c] = g2(a, c) for |a - c| <= nwb - the operands A(w1) = g2(w1 + wb, w1), B(w2) = g2(w2, w2 + wb) of the
// matrix product in k_meas_g4iw, which would otherwise be gathered one 16-byte element per cache line.
__global__ __launch_bounds__(256) void k_meas_g2iw(DevConst C, int n2, int nwb, double2 *G2)
{
    extern __shared__ double2 g2s[];
    const int w = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, kmax = C.kmax;
    const int k = C.K[(size_t)w * C.nflav + f];
    double2 *out = G2 + ((size_t)w * C.nflav + f) * g2_stride(n2, nwb);
    if (k == 0) { for (int e = tid; e < (int)g2_stride(n2, nwb); e += 256) out[e] = make_double2(0.0, 0.0); return; }
    double2 *Ea = g2s, *Ec = g2s + (size_t)kmax * G2_FC, *U = Ec + (size_t)kmax * G2_FC;
    const double *tcp = C.TC + ((size_t)w * C.nflav + f) * kmax, *tap = C.TA + ((size_t)w * C.nflav + f) * kmax;
    const double *Mf = C.M + ((size_t)w * C.nflav + f) * kmax * kmax;
    const double pib = M_PI / C.beta;
    for (int a0 = 0; a0 < n2; a0 += G2_FC) {
        const int na = min(G2_FC, n2 - a0);
        for (int e = tid; e < k * na; e += 256) {
            int ia = e / na, wa = e - ia * na;
            double sn, cs; sincos(pib * (double)(2 * (a0 + wa + 1) - n2 - 1) * tap[ia], &sn, &cs);
            Ea[ia * G2_FC + wa] = make_double2(cs, sn);
        }
        __syncthreads();
        for (int e = tid; e < k * na; e += 256) {
            int ic = e / na, wa = e - ic * na;
            double re = 0.0, im = 0.0;
            for (int ia = 0; ia < k; ++ia) { double m = Mf[(size_t)ic * kmax + ia]; double2 z = Ea[ia * G2_FC + wa]; re = fma(m, z.x, re); im = fma(m, z.y, im); }
            U[ic * G2_FC + wa] = make_double2(re, im);
        }
        for (int c0 = 0; c0 < n2; c0 += G2_FC) {
            const int nc = min(G2_FC, n2 - c0);
            __syncthreads();
            for (int e = tid; e < k * nc; e += 256) {
                int ic = e / nc, wc = e - ic * nc;
                double sn, cs; sincos(-pib * (double)(2 * (c0 + wc + 1) - n2 - 1) * tcp[ic], &sn, &cs);
                Ec[ic * G2_FC + wc] = make_double2(cs, sn);
            }
            __syncthreads();
            for (int e = tid; e < na * nc; e += 256) {
                int wa = e / nc, wc = e - wa * nc;
                double re = 0.0, im = 0.0;
                for (int ic = 0; ic < k; ++ic) {
                    double2 a = Ec[ic * G2_FC + wc], b = U[ic * G2_FC + wa];
                    re += a.x * b.x - a.y * b.y; im += a.x * b.y + a.y * b.x;
                }
                out[(size_t)(a0 + wa) * n2 + (c0 + wc)] = make_double2(re, im);
                const int dd = (a0 + wa) - (c0 + wc);
                if (dd >= -nwb && dd <= nwb) out[(size_t)n2 * n2 + (size_t)(dd + nwb) * n2 + (c0 + wc)] = make_double2(re, im);
            }
        }
        __syncthreads();
    }
}

// Assembly of the two-particle Green's function from products of two g2 (CTQMC.F90:2175-2222) and accumulation of g2.
// acc4 / acc4pp layout [nb4][f1][f2][wb][w1][w2] (complex; the host turns it into the reference's (w1, w2, wb) order),
// acc2 [nb4][f][wa][wc]; chains of sample bin b go to b % nb4.  Three kernels add to acc4 / acc4pp one after the other:
//
// k_meas_g4iw: the direct term of the particle-hole channel.  For fixed (f1, f2, wb), with A_ch(w1) = g2_ch[f1](w1 + wb, w1)
// and B_ch(w2) = g2_ch[f2](w2, w2 + wb), it is  sum_ch A_ch(w1) B_ch(w2)  - a complex matrix product over the CHAINS of the
// accumulator bin, (2 N4iwf) x chains times chains x (2 N4iwf).  That is a real fp64 contraction of a size the matrix
// cores take: four v_mfma_f64_16x16x4_f64 per 16 x 16 tile and four chains (Re += Ar Br - Ai Bi, Im += Ar Bi + Ai Br),
// instead of four FMAs and two gathered loads per element and chain.  One workgroup = (f1, f2, wb, accumulator bin), its
// four waves cover a 64 x 64 region of (w1, w2) in 2 x 2 tiles of 16 x 16 each (16 accumulator registers of 4 doubles per
// lane); the operands A (lane = (w1, chain)) and B (lane = (chain, w2)) are contiguous rows of the chains' g2 tables (the
// diagonals k_meas_g2iw stores behind the matrix).  C/D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 reg.
// k_meas_g4iw_exch / k_meas_g4iw_pp: the terms that do not factorise (below).  The accumulators cross HBM once per kernel.
typedef double g4_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k_meas_g4iw(DevConst C, int nwf, int nwb, int zph, int nb4, const double2 *__restrict__ G2, double2 *__restrict__ acc)
{
    const int nf = C.nflav, n2 = 2 * (nwf + nwb), h = n2 / 2, nw = 2 * nwf, nb = 2 * nwb + 1;
    const size_t n4 = (size_t)nf * nf * nw * nw * nb, g2sz = g2_stride(n2, nwb);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nreg = (nw + 63) / 64;                               // 64 x 64 regions per direction
    // Workgroups go to the eight XCDs round-robin in launch order, and every XCD has an L2 of its own: all workgroups of an
    // accumulator bin read the same g2 tables, so they are dealt to ONE XCD (bin = 8 * group + XCD), one (f1, f2) after the
    // other - with the launch order as it comes every L2 fetched every table (measured: 18.8 GB from the fabric for 2.1 GB of
    // tables).  The grid's z extent is nb4 rounded up to eight.
    const unsigned per_bin = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned slot = lin >> 3, inner = slot % per_bin;
    const int b4 = (int)((lin & 7u) + 8u * (slot / per_bin));
    if (b4 >= nb4) return;
    const int bx = (int)(inner % gridDim.x);
    const int reg_i = bx / nreg, reg_j = bx - reg_i * nreg;
    int r = (int)(inner / gridDim.x);
    const int wbi = r % nb; r /= nb;
    const int f2 = r % nf, f1 = r / nf;
    const int wb = wbi - nwb;
    const int r0 = reg_i * 64 + (wv >> 1) * 32, c0 = reg_j * 64 + (wv & 1) * 32;      // first row / column index (0-based) of this wave
    if (r0 >= nw || c0 >= nw) return;
    const int li = lane & 15, lk = lane >> 4;
    // operand positions inside a chain's g2 table: A of row tile t at w1 = r0 + 16 t + li, B of column tile t at w2 = c0 + 16 t + li
    unsigned offA[2], offB[2]; bool okA[2], okB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i1 = r0 + 16 * t + li, i2 = c0 + 16 * t + li, w1 = i1 - nwf, w2 = i2 - nwf;
        okA[t] = i1 < nw; okB[t] = i2 < nw;
        // A = g2(a1, c1) with a1 - c1 = wb, B = g2(a2, c2) with a2 - c2 = -wb, both conventions: rows wb + nwb and -wb + nwb of the diagonals
        const int c1 = zph == 0 ? w1 : w1 - wb, c2 = zph == 0 ? w2 + wb : w2;
        offA[t] = okA[t] ? (unsigned)(n2 * n2 + (wb + nwb) * n2 + (c1 + h)) : 0u; offB[t] = okB[t] ? (unsigned)(n2 * n2 + (nwb - wb) * n2 + (c2 + h)) : 0u;
    }
    g4_d4 re[2][2], im[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { re[a][b] = (g4_d4)(0.0); im[a][b] = (g4_d4)(0.0); }
    for (int bin = b4; bin < C.nbins; bin += nb4) {
        const int w0 = bin * C.walkers_per_bin, w1e = min(w0 + C.walkers_per_bin, C.nwalkers);
        for (int w4 = w0; w4 < w1e; w4 += 4) {
            const int wl = w4 + lk;
            const bool wok = wl < w1e;
            const double2 *g1 = G2 + ((size_t)(wok ? wl : w0) * nf + f1) * g2sz, *g2p = G2 + ((size_t)(wok ? wl : w0) * nf + f2) * g2sz;
            double ar[2], ai[2], br[2], bi[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double2 a = g1[offA[t]], b = g2p[offB[t]];
                ar[t] = (wok && okA[t]) ? a.x : 0.0; ai[t] = (wok && okA[t]) ? a.y : 0.0;
                br[t] = (wok && okB[t]) ? b.x : 0.0; bi[t] = (wok && okB[t]) ? b.y : 0.0;
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    re[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ti], br[tj], re[ti][tj], 0, 0, 0);
                    re[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai[ti], bi[tj], re[ti][tj], 0, 0, 0);
                    im[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ti], bi[tj], im[ti][tj], 0, 0, 0);
                    im[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[ti], br[tj], im[ti][tj], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i1 = r0 + 16 * ti + lk + 4 * q, i2 = c0 + 16 * tj + li;
                if (i1 >= nw || i2 >= nw) continue;
                const size_t e = ((((size_t)f1 * nf + f2) * nb + wbi) * nw + i1) * nw + i2;      // w2 fastest: the wave writes runs of 16 elements
                double2 *d = acc + (size_t)b4 * n4 + e; double2 o = *d; o.x += re[ti][tj][q]; o.y += im[ti][tj][q]; *d = o;
            }
}
// Exchange term of the particle-hole channel (equal flavours):  - g(w2, w1) g(w1 + wb, w2 + wb)  (zph = 0)  or
// - g(w1, w2) g(w2 - wb, w1 - wb)  (zph = 1).  One factor does not depend on wb, the other is the same table read wb places
// further along the diagonal: a workgroup owns a 16 x 16 tile of (w1, w2) for G4X_WB consecutive bosonic frequencies, thread =
// element, and per chain stages the tile of the fixed factor (16 x 16) and the window of the shifted one ((16 + G4X_WB - 1)^2)
// in LDS once - each laid out so that the lanes of a wave (w2 fastest) read consecutive addresses, i.e. the factor whose ROW runs
// with w2 is transposed on its way in.  The element-wise version inside k_meas_g4iw read two 64 x 64 windows per chain and
// bosonic frequency (~40 GB of L2 traffic per launch at N4iwf = 30, N4iwb = 10, 4096 chains; this one: 9 GB) and waited for
// them with two waves per SIMD.  Runs behind k_meas_g4iw<false> on the same stream and adds to the same accumulators.
constexpr int G4X_WB = 7, G4X_WIN = 16 + G4X_WB - 1;
__global__ __launch_bounds__(256) void k_meas_g4iw_exch(DevConst C, int nwf, int nwb, int zph, int nb4, const double2 *__restrict__ G2, double2 *__restrict__ acc)
{
    __shared__ double2 fixt[16][17], win[G4X_WIN][G4X_WIN + 1];
    const int nf = C.nflav, n2 = 2 * (nwf + nwb), h = n2 / 2, nw = 2 * nwf, nb = 2 * nwb + 1;
    const size_t n4 = (size_t)nf * nf * nw * nw * nb, g2sz = g2_stride(n2, nwb);
    const int nt = (nw + 15) / 16, nch = (nb + G4X_WB - 1) / G4X_WB;
    // same dealing of accumulator bins to XCDs as in k_meas_g4iw
    const unsigned per_bin = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned slot = lin >> 3, inner = slot % per_bin;
    const int b4 = (int)((lin & 7u) + 8u * (slot / per_bin));
    if (b4 >= nb4) return;
    const int bx = (int)(inner % gridDim.x), f = (int)(inner / gridDim.x);
    const int ch = bx % nch, tile = bx / nch, ti = tile / nt, tj = tile - ti * nt;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = ti * 16, c0 = tj * 16;                     // first w1 / w2 index (0-based) of the tile
    const int wbi0 = ch * G4X_WB, wb0 = wbi0 - nwb;            // first bosonic frequency of this chunk
    const bool ok = r0 + ty < nw && c0 + tx < nw;
    // table positions (index + h): fixed factor rows / columns and the window's first row / column
    //   zph 0: fixed = g(w2, w1) -> fixt[w1][w2] (transposed on the way in); window(a, b) = g(R + wb0 + a, Cc + wb0 + b), read at (ty + j, tx + j)
    //   zph 1: fixed = g(w1, w2) -> fixt[w1][w2];  window stored transposed: win[a][b] = g(Cc - wb0 - (G4X_WB - 1) + b, R - wb0 - (G4X_WB - 1) + a),
    //          read at (ty + G4X_WB - 1 - j, tx + G4X_WB - 1 - j)
    const int R = r0 - nwf + h, Cc = c0 - nwf + h;
    double re[G4X_WB], im[G4X_WB];
#pragma unroll
    for (int j = 0; j < G4X_WB; ++j) { re[j] = 0.0; im[j] = 0.0; }
    auto at = [&](const double2 *g, int row, int col) { return (row >= 0 && row < n2 && col >= 0 && col < n2) ? g[(unsigned)(row * n2 + col)] : make_double2(0.0, 0.0); };
    for (int bin = b4; bin < C.nbins; bin += nb4) {
        const int w0 = bin * C.walkers_per_bin, w1e = min(w0 + C.walkers_per_bin, C.nwalkers);
        for (int w = w0; w < w1e; ++w) {
            const double2 *g = G2 + ((size_t)w * nf + f) * g2sz;
            __syncthreads();
            if (zph == 0) {
                fixt[tx][ty] = at(g, Cc + ty, R + tx);                         // row w2 = ty, column w1 = tx (coalesced), stored as [w1][w2]
                for (int e = threadIdx.x; e < G4X_WIN * G4X_WIN; e += 256) { const int a = e / G4X_WIN, b = e - a * G4X_WIN; win[a][b] = at(g, R + wb0 + a, Cc + wb0 + b); }
            } else {
                fixt[ty][tx] = at(g, R + ty, Cc + tx);
                for (int e = threadIdx.x; e < G4X_WIN * G4X_WIN; e += 256) { const int b = e / G4X_WIN, a = e - b * G4X_WIN; win[a][b] = at(g, Cc - wb0 - (G4X_WB - 1) + b, R - wb0 - (G4X_WB - 1) + a); }
            }
            __syncthreads();
            const double2 fx = fixt[ty][tx];
#pragma unroll
            for (int j = 0; j < G4X_WB; ++j) {
                const double2 v = zph == 0 ? win[ty + j][tx + j] : win[ty + G4X_WB - 1 - j][tx + G4X_WB - 1 - j];
                re[j] -= fx.x * v.x - fx.y * v.y; im[j] -= fx.x * v.y + fx.y * v.x;
            }
        }
    }
    if (!ok) return;
#pragma unroll
    for (int j = 0; j < G4X_WB; ++j) {
        const int wbi = wbi0 + j;
        if (wbi >= nb) break;
        const size_t e = ((((size_t)f * nf + f) * nb + wbi) * nw + (r0 + ty)) * nw + (c0 + tx);
        double2 *d = acc + (size_t)b4 * n4 + e; double2 o = *d; o.x += re[j]; o.y += im[j]; *d = o;
    }
}
// Particle-particle channel (CTQMC.F90:2200-2222, ZPHConvention = 0 only), same tiling: with K = wb - 1 (+ 2h in table positions)
//   direct    g_f1(K - w2, w1) g_f2(w2, K - w1)         X: row falls with w2 and rises with wb, Y: column falls with w1, rises with wb
//   exchange  - g_f1(w2, w1) g_f1(K - w2, K - w1)       (equal flavours) fixed tile times a window that moves along the anti-diagonal
// Four LDS arrays per chain, each stored so that a wave (w2 = tx fastest) reads consecutive addresses: xs[w1][K - w2],
// ys[K - w1][w2], fs[w1][w2], ws[K - w1][K - w2] (window coordinates 15 + j - t for bosonic frequency j of the chunk).
__global__ __launch_bounds__(256) void k_meas_g4iw_pp(DevConst C, int nwf, int nwb, int nb4, const double2 *__restrict__ G2, double2 *__restrict__ acc)
{
    __shared__ double2 xs[16][G4X_WIN + 1], ys[G4X_WIN][17], fs[16][17], ws[G4X_WIN][G4X_WIN + 1];
    const int nf = C.nflav, n2 = 2 * (nwf + nwb), h = n2 / 2, nw = 2 * nwf, nb = 2 * nwb + 1;
    const size_t n4 = (size_t)nf * nf * nw * nw * nb, g2sz = g2_stride(n2, nwb);
    const int nt = (nw + 15) / 16, nch = (nb + G4X_WB - 1) / G4X_WB;
    const unsigned per_bin = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned slot = lin >> 3, inner = slot % per_bin;
    const int b4 = (int)((lin & 7u) + 8u * (slot / per_bin));
    if (b4 >= nb4) return;
    const int bx = (int)(inner % gridDim.x), ff = (int)(inner / gridDim.x), f1 = ff / nf, f2 = ff - f1 * nf;
    const int ch = bx % nch, tile = bx / nch, ti = tile / nt, tj = tile - ti * nt;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = ti * 16, c0 = tj * 16;
    const int wbi0 = ch * G4X_WB, wb0 = wbi0 - nwb;
    const bool ok = r0 + ty < nw && c0 + tx < nw, same = f1 == f2;
    const int R = r0 - nwf + h, Cc = c0 - nwf + h, K0 = wb0 - 1 + 2 * h;      // table positions: w1 + h = R + ty, w2 + h = Cc + tx, (K - w) + h = K0 + j - (w + h)
    double re[G4X_WB], im[G4X_WB];
#pragma unroll
    for (int j = 0; j < G4X_WB; ++j) { re[j] = 0.0; im[j] = 0.0; }
    auto at = [&](const double2 *g, int row, int col) { return (row >= 0 && row < n2 && col >= 0 && col < n2) ? g[(unsigned)(row * n2 + col)] : make_double2(0.0, 0.0); };
    for (int bin = b4; bin < C.nbins; bin += nb4) {
        const int w0 = bin * C.walkers_per_bin, w1e = min(w0 + C.walkers_per_bin, C.nwalkers);
        for (int w = w0; w < w1e; ++w) {
            const double2 *g1 = G2 + ((size_t)w * nf + f1) * g2sz, *g2p = G2 + ((size_t)w * nf + f2) * g2sz;
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * G4X_WIN; e += 256) {
                { const int u = e / 16, c = e - u * 16; xs[c][u] = at(g1, K0 - Cc - 15 + u, R + c); }                    // column (w1) fastest in memory
                { const int t = e / G4X_WIN, v = e - t * G4X_WIN; ys[v][t] = at(g2p, Cc + t, K0 - R - 15 + v); }         // column (K - w1) fastest
            }
            if (same) {
                fs[tx][ty] = at(g1, Cc + ty, R + tx);
                for (int e = threadIdx.x; e < G4X_WIN * G4X_WIN; e += 256) { const int b = e / G4X_WIN, a = e - b * G4X_WIN; ws[a][b] = at(g1, K0 - Cc - 15 + b, K0 - R - 15 + a); }
            }
            __syncthreads();
            const double2 fx = same ? fs[ty][tx] : make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < G4X_WB; ++j) {
                const double2 xv = xs[ty][15 + j - tx], yv = ys[15 + j - ty][tx];
                re[j] += xv.x * yv.x - xv.y * yv.y; im[j] += xv.x * yv.y + xv.y * yv.x;
                if (same) { const double2 v = ws[15 + j - ty][15 + j - tx]; re[j] -= fx.x * v.x - fx.y * v.y; im[j] -= fx.x * v.y + fx.y * v.x; }
            }
        }
    }
    if (!ok) return;
#pragma unroll
    for (int j = 0; j < G4X_WB; ++j) {
        const int wbi = wbi0 + j;
        if (wbi >= nb) break;
        const size_t e = ((((size_t)f1 * nf + f2) * nb + wbi) * nw + (r0 + ty)) * nw + (c0 + tx);
        double2 *d = acc + (size_t)b4 * n4 + e; double2 o = *d; o.x += re[j]; o.y += im[j]; *d = o;
    }
}
// g2 itself (MeasG2iw): thread = element, workgroup = (256 elements, accumulator bin, chunk of G2ACC_CHUNK sample bins of that
// accumulator bin); the chunk's chains are summed in registers (reads: unit stride across the wave) and added with one
// return-less atomic per element - a thread per element alone leaves the GPU to 25600 threads that each walk all chains
constexpr int G2ACC_CHUNK = 16;
__global__ __launch_bounds__(256) void k_meas_g2acc(DevConst C, int n2, int nwb, int nb4, const double2 *__restrict__ G2, double2 *__restrict__ acc2)
{
    const int nf = C.nflav;
    const size_t ng2 = (size_t)nf * n2 * n2, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b4 = blockIdx.y;
    if (e >= ng2) return;
    const int f = (int)(e / ((size_t)n2 * n2)); const size_t q = e % ((size_t)n2 * n2);
    double re = 0.0, im = 0.0;
    const int first = b4 + (int)blockIdx.z * G2ACC_CHUNK * nb4;
    for (int c = 0, bin = first; c < G2ACC_CHUNK && bin < C.nbins; ++c, bin += nb4) {
        const int w0 = bin * C.walkers_per_bin, w1e = min(w0 + C.walkers_per_bin, C.nwalkers);
        for (int w = w0; w < w1e; ++w) { const double2 x = G2[((size_t)w * nf + f) * g2_stride(n2, nwb) + q]; re += x.x; im += x.y; }
    }
    double *d = (double *)(acc2 + (size_t)b4 * ng2 + e);
    atomicAdd(d, re); atomicAdd(d + 1, im);
}
#endif  // W2DYN_TU_MEAS

}  // namespace w2dyn
