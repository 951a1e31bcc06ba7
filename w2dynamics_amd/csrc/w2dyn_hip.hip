// Host side of libw2dyn_hip.so: the C ABI declared in include/w2dyn_hip.h.
// Mirrors the call sequence of the reference's f2py module (CTQMC.F90:7894-9426) for the density-density /
// diagonal-hybridisation Z-space path; refuses every other mode loudly.  No CPU fallback exists: all sampling and
// all estimators run in the kernels of cthyb_kernels.hip.h.
#include "../../include/w2dyn_hip.h"
#include "launch.hpp"
#include "local_problem.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

using namespace w2dyn;

namespace {

struct Param { std::string key, value; };

struct KTimer {
    double total_ms = 0.0; int64_t launches = 0;
};

// Scratch of w2dyn_measure_once_noalloc: the per-chain g2(nu, nu') table that k_meas_g2iw writes and k_meas_g4iw reads
// (the other entry points keep it in a hipMalloc'ed buffer sized at init_solver).  One copy per GPU, held until the process
// ends.  `w2dyn_scratch` is a host-side stand-in for the symbol, not a device address: the address of each GPU's copy is
// looked up once per device with hipGetSymbolAddress (scratch_address below) and only that pointer is handed to kernels.
constexpr size_t W2DYN_SCRATCH_BYTES = (size_t)32 << 20;
constexpr int W2DYN_MAX_DEVICES = 64;
void *g_scratch_ptr[W2DYN_MAX_DEVICES] = {nullptr};
std::mutex g_scratch_mutex;

// Soft abort (Signals.F90:117-137): SIGTERM / SIGINT / SIGUSR1 are trapped from init_solver to the end of ctqmc_measure
// (CTQMC.F90:8076-8078, 9352-9355); the handler only raises this flag, the sampling loops look at it between launches.
// The handlers are process-wide, the contexts are not: every context that has a solver holds one reference, the first
// one installs the handlers, the last one puts the previous ones back (acquire_traps / release_traps below, which also
// run on every error exit of ctqmc_measure and in w2dyn_destroy).  The stepping entry points (w2dyn_sweep,
// w2dyn_measure_once[_noalloc]) have no loop of their own to leave early: they return W2DYN_EINTERRUPTED once the flag is up.
volatile sig_atomic_t g_signal_fired = 0;
int g_trap_users = 0;
std::mutex g_trap_mutex;
struct sigaction g_prev_term, g_prev_int, g_prev_usr1;
void trap_notify(int) { g_signal_fired = 1; }
void install_traps()
{
    std::lock_guard<std::mutex> lock(g_trap_mutex);
    if (g_trap_users++ > 0) return;
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = trap_notify; sigemptyset(&sa.sa_mask);
    sigaction(SIGTERM, &sa, &g_prev_term); sigaction(SIGINT, &sa, &g_prev_int); sigaction(SIGUSR1, &sa, &g_prev_usr1);
}
void clear_traps()      // reset_notify + clear_trap, CTQMC.F90:9352-9355
{
    std::lock_guard<std::mutex> lock(g_trap_mutex);
    if (g_trap_users == 0 || --g_trap_users > 0) return;
    sigaction(SIGTERM, &g_prev_term, nullptr); sigaction(SIGINT, &g_prev_int, nullptr); sigaction(SIGUSR1, &g_prev_usr1, nullptr);
    g_signal_fired = 0;
}

}  // namespace

static __device__ __align__(256) unsigned char w2dyn_scratch[W2DYN_SCRATCH_BYTES];

// address of the current device's copy of w2dyn_scratch; `device` must be the current device
static hipError_t scratch_address(int device, void **out)
{
    if (device < 0 || device >= W2DYN_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(w2dyn_scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// one per-bin result as w2dyn_reduce_pack needs it: accumulator on the device, bins, doubles per bin, the factor that
// normalises each bin (host-made in measure_finish), and Nleg if the element factor sqrt(2l+1) applies
struct PackDesc { const double *dev; int B; size_t per; std::vector<double> scale; int nleg; };

struct w2dyn_solver {
    int device = 0;
    void *scratch = nullptr;                         // this GPU's copy of w2dyn_scratch (looked up in w2dyn_create)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // layout
    int nwalkers = 1024, kmax_req = 0, walkers_per_bin = 8;
    int64_t first_walker = 0;
    int64_t seed = 0;
    int fourpnt = 0;
    std::vector<Param> paras;
    bool have_paras = false, have_solver = false;
    bool has_traps = false;                          // this context holds a reference on the signal handlers
    // parameters picked up at init_solver
    int nd = 0, nflav = 0, nstates = 0, nftau = 0, ntau = 0, nleg = 0, niw = 0, meas_giw = 0, ncorr = 0, maxhisto = 0;
    int64_t nmeas = 0;
    double beta = 0, p_glob = 0, p_shift = 0, taudiffmax_param = 0;
    double p_4op = 0, p_outer = 0;       // Percentage4OperatorMove, PercentageOuterMove
    bool flavc = false;                  // flavourchange_moves = 1 (with offdiag = 1)
    bool offdiag = false;                // offdiag = 1: one hybridisation matrix over all flavours (csrc/cthyb_offdiag.hip.h)
    double *acc_gleg_full = nullptr, *acc_gtau_full = nullptr;      // [nbins][nflav][nflav][nleg | ntau]
    int has_szt = 0;
    std::vector<double> energies, udd, eps, outer_w, outer_cum, ftab;
    double egs = 0;
    LocalProblem lp;                 // blocks, eigenbasis and operators of H_loc
    std::vector<int> dmoff;          // offsets of the blocks inside the block-diagonal density matrix
    bool mat = false;                // matrix local-trace path (blocks with more than one state)
    // Z-space two-particle Matsubara estimators (FourPnt & 4)
    int n4iwf = 0, n4iwb = 0, meas_g2iw = 0, meas_g4iwpp = 0, zph = 0, nb4 = 0;
    int n4tau = 0, n4leg = 0, nbl = 0; double *acc_g4tau = nullptr, *acc_g4leg = nullptr;     // FourPnt bits 1, 2 (MeasGtau4PntLeg)
    double2 *G2 = nullptr, *acc4 = nullptr, *acc4pp = nullptr, *acc2 = nullptr;
    // worm sampling (sector 4): move probabilities, eta, estimator switches, accumulator bins, sector of the running call
    double p_wins = 0, p_wrep = 0, worm_eta = 0;
    int worm_meas_g4iw = 0, g4ph = 0, g4pp = 0, wormphconv = 0, nbw = 0, cur_isector = 1;
    // G(tau) accumulator (AccumGtau, CTQMC.F90:492-500): 0 = G(tau)/G(l) not measured, 1 = mean, 2 = per-chain autocorrelation stack
    bool meas_ssq = false;                                               // MeasSsquare: <S^2> per electron number from the density matrix
    int step_series = 0; double *ser = nullptr;                          // Gtau_mean_step (1) | Gtau_mid_step (2) | sign_step (4): [2 nflav + 1][Nmeas] chain-mean series
    bool meas_susz = false; int nbs = 0; double *acc_susz = nullptr;     // MeasSusz: <n(tau) n(0)>, accumulator bins [nbs][nflav][nflav][Ntau]
    double *susz_work = nullptr; const double *susz_dens = nullptr;                   // matrix trace path: per-chain bra / ket vectors, density matrices of the eigenbasis
    int accum_gtau = 1, ac_lev = 0; long long ac_n = 0; double *ac_dat = nullptr, *ac_out = nullptr;
    int res_isector = 1;                             // sector of the run the result table was made from
    int worm_meas_giw = 0, worm_meas_gtau = 0;       // sector 2: WormMeasGiw, WormMeasGtau (CTQMC.F90:597-612)
    // sectors 6 - 9 (equal-time pairs): WormMeasP2iwPH, P2iwPP, P2tauPH, P2tauPP, P3iwPH, P3iwPP; N2iwb, N3iwf, N3iwb (CTQMC.F90:654-726)
    int worm_meas_eq[6] = {0, 0, 0, 0, 0, 0}, n2iwb = 0, n3iwf = 0, n3iwb = 0, nbe = 0;
    std::vector<double> umat_h;                      // U matrix as handed to init_solver
    int worm_meas_gsigma = 0;                        // sector 3: WormMeasGSigmaiw (CTQMC.F90:619-623); shares the giw_worm accumulator
    double *acc_weq_iw = nullptr, *acc_weq_tau = nullptr, *acc_weq_p3 = nullptr;
    // worm sectors 5, 10 - 16 (CTQMC.F90:379-430, 624-652): WormMeasH4iw, WormMeasQQ, WormMeasQQtau, WormMeasQQQQ, WormMeasNQQdag, WormMeasQQdd, WormMeasUcaca,
    // WormMeasUcacatau, WormMeasUccaa, WormMeasUccaatau, WormMeasQUDdag; one sector is sampled per run, so the estimators share one complex and one tau accumulator
    int worm_meas_more[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nbie = 0;
    size_t wie_n = 0;                                // doubles per bin of acc_wie_iw
    double *acc_wie_iw = nullptr, *acc_wie_tau = nullptr;
    double wie_sparsity = 1.0;                       // u_sparsity of the sector and component being sampled (CTQMC.F90:9013-9343)
    bool no_sampling = false;
    std::vector<long long> steps_worm_chains;
    // configurations of all chains kept on the host across init_solver calls (w2dyn_save_mc_configs / w2dyn_restore_mc_configs)
    std::vector<WalkerScalars> snap_ws; std::vector<int> snap_K; std::vector<double> snap_TC, snap_TA; int snap_nf = 0, snap_kmax = 0;
    // window calibration (ctqmc_calibrate): default AccPair histogram and the one on the calibration grid
    unsigned long long *accpair_default = nullptr;
    std::vector<double> accpairtau, accpairtau_chains;
    DevConst C{};
    std::vector<void *> allocs;
    // accumulators bookkeeping
    int64_t meas_done = 0, steps_done = 0;
    int aborted = 0;
    double mean_sign = 0, time_warmup = 0, time_sim = 0, time_sim_steps = 0;
    std::map<std::string, std::vector<double>> res;      // normalised results (doubles / complex interleaved)
    std::vector<long long> res_accpair;
    double *gleg_leg = nullptr, *gleg_tab = nullptr;            // Legendre moments made from acc_gleg (Chebyshev moments) by measure_finish; conversion table
    unsigned long long *gleg_segpre = nullptr; int ncu = 256;   // work list of k_meas_gleg; compute units of the device
    std::map<std::string, PackDesc> pack;                // filled by measure_finish, dropped with the accumulators
    double *packbuf = nullptr, *packscale = nullptr; size_t packbuf_n = 0, packscale_n = 0;   // w2dyn_reduce_pack; live until w2dyn_destroy
    std::vector<double> counters;                        // summed over walkers, C_NCOUNT
    std::map<std::string, KTimer> timers;
    std::vector<hipEvent_t> ev_pool; std::vector<std::pair<std::string, size_t>> ev_pending;   // see drain_timers
    int last_count_sampling = 0;
    bool timing = false;                                  // HIP-event timing of every launch (a host sync per launch): bench / profiling only, w2dyn_set_timing
    // position of every chain in its random stream, kept across init_solver / dest_ctqmc until the next init_rnd: the
    // reference seeds once (impurity.py:387, CTQMC.F90:9404-9408) and its generator state survives init_solver
    std::vector<unsigned long long> stream_pos;
};

namespace {
void acquire_traps(w2dyn_solver *s) { if (!s->has_traps) { install_traps(); s->has_traps = true; } }
void release_traps(w2dyn_solver *s) { if (s->has_traps) { clear_traps(); s->has_traps = false; } }

int fail(w2dyn_solver *s, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (s) s->err = buf;
    return code;
}

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(s, W2DYN_EHIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

// ---- parameter string (Parameters.F90:281-320; lookup by prefix of the stored identifier, 77-105) ----
const Param *find_param(const w2dyn_solver *s, const char *id)
{
    size_t n = strlen(id);
    for (const auto &p : s->paras) if (p.key.compare(0, n, id) == 0) return &p;
    return nullptr;
}
bool get_int(w2dyn_solver *s, const char *id, long long &out)
{
    const Param *p = find_param(s, id);
    if (!p) { fail(s, W2DYN_EINVAL, "%s not found in parameter list!", id); return false; }
    if (p->value == "True") { out = 1; return true; }
    if (p->value == "False" || p->value.empty()) { out = 0; return true; }
    char *end = nullptr;
    double v = strtod(p->value.c_str(), &end);
    if (end == p->value.c_str()) { fail(s, W2DYN_EINVAL, "%s is not an integer", id); return false; }
    out = (long long)v; return true;
}
bool get_real(w2dyn_solver *s, const char *id, double &out)
{
    const Param *p = find_param(s, id);
    if (!p) { fail(s, W2DYN_EINVAL, "%s not found in parameter list!", id); return false; }
    out = p->value.empty() ? 0.0 : strtod(p->value.c_str(), nullptr); return true;
}
bool get_str(w2dyn_solver *s, const char *id, std::string &out)
{
    const Param *p = find_param(s, id);
    if (!p) { fail(s, W2DYN_EINVAL, "%s not found in parameter list!", id); return false; }
    out = p->value; return true;
}

template <class T> int dev_alloc(w2dyn_solver *s, T *&ptr, size_t count)
{
    void *p = nullptr;
    // 256 bytes of zeroed padding behind every array: k_meas_giw reads rows of M in aligned segments of 4 doubles per
    // scalar load and may run up to 184 bytes past the last row (the values meet zero phases)
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 256;
    HIPCHK(hipMalloc(&p, bytes));
    HIPCHK(hipMemsetAsync(p, 0, bytes, s->stream));
    s->allocs.push_back(p);
    ptr = (T *)p;
    return 0;
}
template <class T> int dev_upload(w2dyn_solver *s, const T *&ptr, const std::vector<T> &v)
{
    T *p = nullptr;
    int rc = dev_alloc(s, p, v.size());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
    ptr = p;
    return 0;
}

// remember where every chain stands in its random stream (see w2dyn_solver::stream_pos)
void stash_streams(w2dyn_solver *s)
{
    if (!s->have_solver || !s->C.ws) return;
    std::vector<WalkerScalars> ws(s->nwalkers);
    if (hipStreamSynchronize(s->stream) != hipSuccess) return;
    if (hipMemcpy(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost) != hipSuccess) return;
    s->stream_pos.resize(ws.size());
    for (size_t w = 0; w < ws.size(); ++w) s->stream_pos[w] = ws[w].ndraw;
}

void free_device(w2dyn_solver *s)
{
    stash_streams(s);
    for (void *p : s->allocs) (void)hipFree(p);
    s->allocs.clear();
    s->pack.clear();
    s->have_solver = false;
}

// memory budget of each family of two-particle accumulator bins (g4iw / g4iw_pp / g2iw, g4tau / g4leg, g4iw_worm): when one
// bin per sample bin would exceed it, several sample bins share an accumulator bin (fewer, larger samples for the error bar).
// W2DYN_ACC_BYTES overrides the default of 2 GiB.
size_t acc_budget()
{
    if (const char *e = getenv("W2DYN_ACC_BYTES")) { const long long v = atoll(e); if (v > 0) return (size_t)v; }
    return (size_t)2 << 30;
}

inline int flav_bit_h(int f, int nd) { return (f >> 1) + nd * (f & 1); }

// window parameters, Trace.F90:1608-1648
void host_init_window(w2dyn_solver *s, double tdm, int &num_windows, double &taudiff_max)
{
    if (tdm > 0.0) {
        taudiff_max = tdm;
        int nw = (int)std::floor(s->beta / tdm) - 1;
        num_windows = nw > 1 ? nw : 1;
    } else { taudiff_max = s->beta; num_windows = 1; }
}


// Everything that follows from the expansion-order capacity kmax: window-list capacity, order limit of the LDS inversion,
// LDS layout of the sweep kernels (init_solver, and again whenever the capacity is enlarged)
int configure_kmax(w2dyn_solver *s, int kmax)
{
    DevConst &C = s->C; const LocalProblem &lp = s->lp; const int nf = s->nflav;
    C.kmax = kmax;
    C.wcap = std::max(kmax, std::min(2 * nf * kmax, 256));
    C.kgj = std::min(kmax, 32);
    C.shared_lds = 0; C.op_in_lds = 0;
    if (s->mat) {
        const size_t nop = (size_t)lp.nsst * nf * 2 * C.nsm * C.nsm;
        size_t ints = (size_t)lp.nsst * nf * 4 * sizeof(short) + 2 * (size_t)lp.nsst * sizeof(int);
        ints = (ints + 7) & ~(size_t)7;
        size_t small = ints + (size_t)s->nstates * sizeof(double);
        // Occupancy first: when two workgroups (2 waves per SIMD) fit into the CU's 160 KB with the operator matrices left
        // in global memory (they are L2 resident) and a 16x16 Gauss-Jordan tile, take that layout: measured 1.55x on the
        // t2g Kanamori problem against operators staged in LDS at one wave per SIMD.  W2DYN_MAT_OPS_IN_LDS=1/0 overrides.
        const size_t per_wave16 = lds_bytes_per_wave(nf, kmax, C.wcap, std::min(C.kgj, 16), C.nsm);
        bool two_groups = ((small + 15) & ~(size_t)15) + 4 * per_wave16 <= 80 * 1024;
        if (const char *e = getenv("W2DYN_MAT_OPS_IN_LDS")) two_groups = (atoi(e) == 0);
        if (two_groups) C.kgj = std::min(C.kgj, 16);
        C.op_in_lds = (!two_groups && small + nop * sizeof(double) <= 64 * 1024) ? 1 : 0;
        C.shared_lds = (int)((small + (C.op_in_lds ? nop * sizeof(double) : 0) + 15) & ~(size_t)15);
    }
    // three waves per SIMD pay up to three orbitals (measured: C1, C2, C3 +6-10 %, C5Z -23 %); W2DYN_SWEEP_WPE=2|3 overrides
    C.sweep_wpe3 = (!s->mat && nf <= 6) ? 1 : 0;
    if (const char *e = getenv("W2DYN_SWEEP_WPE")) C.sweep_wpe3 = (!s->mat && atoi(e) == 3) ? 1 : 0;
    C.lds_per_wave = (int)lds_bytes_per_wave(nf, kmax, C.wcap, C.kgj, C.mat ? C.nsm : 0);
    if ((size_t)C.shared_lds + 4 * (size_t)C.lds_per_wave > 160 * 1024 && C.kgj > 16) { C.kgj = 16; C.lds_per_wave = (int)lds_bytes_per_wave(nf, kmax, C.wcap, C.kgj, C.mat ? C.nsm : 0); }
    if ((size_t)C.shared_lds + 4 * (size_t)C.lds_per_wave > 160 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "kmax=%d needs %d bytes of LDS per chain: too large", kmax, C.lds_per_wave);
    if ((s->fourpnt & 4) && (size_t)3 * kmax * G2_FC * sizeof(double2) > 160 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "kmax=%d too large for the two-particle transform", kmax);
    return 0;
}

LaunchCfg walker_launch(const w2dyn_solver *s)
{
    LaunchCfg l;
    // Chains (waves) per workgroup.  A workgroup's wave slots are free again only when its slowest chain is done, and chains
    // differ by the global moves they happen to draw (35 step-times each on C2): one chain per workgroup on the scalar trace
    // path, where a workgroup shares nothing (measured against four: C2 +11 %, C3 +12 %, C1 +6 %, C5Z +28 %, C5 worm +24 %).
    // The matrix trace path stages the local problem's tables once per workgroup: four chains share them (C4: 1.81e8 steps/s
    // with 4, 1.51e8 with 2, 1.80e8 with 1; C5K: 4.2e7, 2.4e7, 2.6e7).  W2DYN_WPG = 1 | 2 | 4 overrides.
    int wpg = s->mat ? 4 : 1;
    if (const char *e = getenv("W2DYN_WPG")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) wpg = v; }
    l.block = dim3(64 * wpg);
    l.grid = dim3((s->nwalkers + wpg - 1) / wpg);
    l.lds = (size_t)s->C.shared_lds + (size_t)wpg * (size_t)s->C.lds_per_wave;
    l.stream = s->stream;
    return l;
}

// worm sectors 5, 10 - 16: index (into worm_meas_more) of the sector's complex array and of its tau histogram (-1: none), the array's axes, and
// whether the sector counts its measurements only when its array was asked for (CTQMC.F90:8600-8640)
struct WieSector { int q_iw, q_tau, n1, n2, n3; bool needs_iw; };
static WieSector wie_sector(const w2dyn_solver *s, int is)
{
    const int f4 = 2 * s->n4iwf, b4 = 2 * s->n4iwb + 1, f3 = 2 * s->n3iwf, b3 = 2 * s->n3iwb + 1, b2 = 2 * s->n2iwb + 1, fg = 2 * s->niw;
    switch (is) {
    case 5: return {0, -1, f4, f4, b4, true};
    case 10: return {1, 2, fg, 1, 1, true};
    case 11: return {3, -1, f4, f4, b4, true};
    case 12: return {4, -1, f3, b3, 1, true};
    case 13: return {5, -1, f3, f3, 1, true};
    case 14: return {6, 7, b2, 1, 1, false};
    case 15: return {8, 9, b2, 1, 1, false};
    default: return {10, -1, fg, 1, 1, false};
    }
}
static bool is_wie_sector(int is) { return is == 5 || (is >= 10 && is <= 16); }
static const char *const kWieNames[11] = {"h4iw_worm", "qq_worm", "qqtau_worm", "qqqq_worm", "nqqdag_worm", "qqdd_worm", "ucaca_worm", "ucacatau_worm", "uccaa_worm",
                                          "uccaatau_worm", "quddag_worm"};
int mat_variant(const w2dyn_solver *s) { return !s->mat ? 0 : (s->lp.nsmax <= 2 ? 2 : (s->lp.nsmax <= 4 ? s->lp.nsmax : 9)); }

// ---- expansion-order capacity --------------------------------------------------------------------------------------
// The reference's matrices grow with the expansion order; here every flavour has kmax slots.  A chain whose flavour fills
// up stops before its next step (ERR_KMAX, WalkerScalars::todo = steps it still owes of that launch); the host then moves
// all chains to larger arrays and lets the stopped ones finish.  Trajectories do not depend on kmax.
int read_flags(w2dyn_solver *s, int &max_order, int &paused)
{
    int f[4] = {0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(f, s->C.flags, sizeof f, hipMemcpyDeviceToHost));
    max_order = f[0]; paused = f[1];
    return 0;
}

int grow_kmax(w2dyn_solver *s, int knew)
{
    DevConst &C = s->C; const int kold = C.kmax, nf = s->nflav; const size_t W = s->nwalkers;
    if (s->offdiag) return fail(s, W2DYN_EOVERFLOW, "offdiag = 1: a chain reached the capacity of %d pairs; start with a larger kmax (w2dyn_set_layout)", kold);
    knew = std::min((knew + 7) & ~7, (int)KMAX_LIMIT);
    if (knew <= kold) return fail(s, W2DYN_EOVERFLOW, "a chain reached the expansion-order capacity kmax=%d, the largest this library handles", kold);
    HIPCHK(hipStreamSynchronize(s->stream));
    double *tc = C.TC, *ta = C.TA, *m = C.M, *ts = C.TS, *mb = C.MB;
    int rc;
    if ((rc = configure_kmax(s, knew))) { configure_kmax(s, kold); return fail(s, W2DYN_EOVERFLOW, "a chain reached the expansion-order capacity kmax=%d and the LDS layout does not allow more", kold); }
    if ((rc = dev_alloc(s, C.TC, W * nf * knew))) return rc;
    if ((rc = dev_alloc(s, C.TA, W * nf * knew))) return rc;
    if ((rc = dev_alloc(s, C.TS, W * nf * 2 * knew))) return rc;
    if ((rc = dev_alloc(s, C.M, W * nf * knew * knew))) return rc;
    if ((rc = dev_alloc(s, C.MB, W * nf * knew * knew))) return rc;
    launch_relayout(C, s->stream, kold, knew, tc, ta, m, C.TC, C.TA, C.M);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    for (double *p : {tc, ta, m, ts, mb}) {
        (void)hipFree(p);
        s->allocs.erase(std::remove(s->allocs.begin(), s->allocs.end(), (void *)p), s->allocs.end());
    }
    return 0;
}

// After a sweep launch the host has waited for: enlarge the capacity while chains are stopped at it and let them finish the
// steps they owe.  kind: 0 = Z-space sweep, 1 = calibration sweep, 2 = worm sweep of the current sector.
int finish_paused(w2dyn_solver *s, int kind)
{
    if (kind == 0 && s->cur_isector != 1) kind = 2;
    for (int round = 0; round < 64; ++round) {
        int maxk = 0, paused = 0, rc;
        if ((rc = read_flags(s, maxk, paused))) return rc;
        if (paused == 0) return 0;
        if ((rc = grow_kmax(s, 2 * s->C.kmax))) return rc;
        std::vector<WalkerScalars> ws(s->nwalkers);
        HIPCHK(hipMemcpy(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
        for (auto &w : ws) w.err &= ~ERR_KMAX;
        HIPCHK(hipMemcpy(s->C.ws, ws.data(), ws.size() * sizeof(WalkerScalars), hipMemcpyHostToDevice));
        const int zero[4] = {maxk, 0, 0, 0};
        HIPCHK(hipMemcpy(s->C.flags, zero, sizeof zero, hipMemcpyHostToDevice));
        LaunchCfg l = walker_launch(s);
        if (kind == 2) launch_sweep_worm(s->cur_isector, mat_variant(s), l, s->C, 0, 1);
        else launch_sweep(mat_variant(s), kind == 1, l, s->C, 0, s->last_count_sampling, 1);
        HIPCHK(hipGetLastError());
    }
    return fail(s, W2DYN_EOVERFLOW, "expansion-order capacity: chains still stopped after repeated enlargement");
}

// between launches of a long run: enlarge early when the largest order comes within a quarter of the capacity, so that a
// chain stopping in the middle of the asynchronous measurement loop (which cannot be repaired afterwards) stays exceptional
int grow_if_close(w2dyn_solver *s)
{
    int maxk = 0, paused = 0, rc;
    if ((rc = read_flags(s, maxk, paused))) return rc;
    if (paused) return fail(s, W2DYN_EOVERFLOW, "a chain reached the expansion-order capacity kmax=%d between two checks of a measurement run; start with a larger kmax (w2dyn_set_layout)", s->C.kmax);
    if (maxk + std::max(8, s->C.kmax / 4) > s->C.kmax && s->C.kmax < KMAX_LIMIT) return grow_kmax(s, 2 * s->C.kmax);
    return 0;
}

// Kernel timing (w2dyn_set_timing): every launch is bracketed by a pair of HIP events recorded on the context's stream; the
// host does not wait for them - the pairs are read out when the stream is waited for anyway (w2dyn_sync, w2dyn_kernel_time)
// or when the pool of pairs runs out.
constexpr size_t TIMER_POOL = 2048;
int drain_timers(w2dyn_solver *s)
{
    if (s->ev_pending.empty()) return 0;
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto &pe : s->ev_pending) {
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, s->ev_pool[2 * pe.second], s->ev_pool[2 * pe.second + 1]));
        auto &t = s->timers[pe.first]; t.total_ms += ms; t.launches += 1;
    }
    s->ev_pending.clear();
    return 0;
}
int time_begin(w2dyn_solver *s)
{
    if (!s->timing) return 0;
    if (s->ev_pending.size() >= TIMER_POOL) { int rc = drain_timers(s); if (rc) return rc; }
    while (s->ev_pool.size() < 2 * (s->ev_pending.size() + 1)) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); s->ev_pool.push_back(e); }
    HIPCHK(hipEventRecord(s->ev_pool[2 * s->ev_pending.size()], s->stream));
    return 0;
}
int time_end(w2dyn_solver *s, const char *name)
{
    if (!s->timing) return 0;
    HIPCHK(hipEventRecord(s->ev_pool[2 * s->ev_pending.size() + 1], s->stream));
    s->ev_pending.emplace_back(name, s->ev_pending.size());
    return 0;
}

int check_walker_errors(w2dyn_solver *s)
{
    std::vector<WalkerScalars> ws(s->nwalkers);
    HIPCHK(hipMemcpyAsync(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    int e = 0;
    for (auto &w : ws) e |= w.err;
    if (e & ERR_KMAX) return fail(s, W2DYN_EOVERFLOW, "a chain stopped at the expansion-order capacity kmax=%d outside a phase that enlarges it; raise it with w2dyn_set_layout", s->C.kmax);
    if (e & ERR_WCAP) return fail(s, W2DYN_EOVERFLOW, "more than %d operators inside the sliding window; reduce TaudiffMax or raise kmax", s->C.wcap);
    if (e & ERR_DETRAT) return fail(s, W2DYN_ENUMERIC, "DetRat too small in removal");
    if (e & ERR_SINGULAR) return fail(s, W2DYN_ENUMERIC, "singular hybridisation matrix in full rebuild");
    if (e & ERR_BLOCKSEQ) return fail(s, W2DYN_ENUMERIC, "internal error: inconsistent block sequence in the local trace");
    return 0;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

namespace w2dyn {
// matrix trace path: the sweep kernels take their chains from DevConst::queue; no more workgroups than can share the GPU twice over
static LaunchCfg queue_launch(const LaunchCfg &l, const DevConst &C)
{
    LaunchCfg q = l;
    (void)hipMemsetAsync(C.queue, 0, sizeof(int), l.stream);
    q.grid = dim3(std::min<unsigned>(l.grid.x, 2048u));
    return q;
}
void launch_sweep(int mat, bool calib, const LaunchCfg &l0, const DevConst &C, long long nsteps, int count, int resume)
{
    const LaunchCfg l = mat ? queue_launch(l0, C) : l0;
    if (C.offdiag) { if (mat <= 4) launch_sweep_od_tree(mat, l, C, nsteps, count, resume); else launch_sweep_od_seq(l, C, nsteps, count, resume); }
    else if (C.xmoves && !calib) {
        if (mat == 0) launch_sweep_x_scalar(l, C, nsteps, count, resume);
        else if (mat <= 4) launch_sweep_x_tree(mat, l, C, nsteps, count, resume);
        else launch_sweep_x_seq(l, C, nsteps, count, resume);
    }
    else if (mat == 0) launch_sweep_scalar(calib, l, C, nsteps, count, resume);
    else if (mat <= 4) launch_sweep_tree(mat, calib, l, C, nsteps, count, resume);
    else launch_sweep_seq(calib, l, C, nsteps, count, resume);
}
void launch_sweep_worm(int isector, int mat, const LaunchCfg &l0, const DevConst &C, long long nsteps, int resume)
{
    const LaunchCfg l = mat ? queue_launch(l0, C) : l0;
    const int nworm = worm_nops_of(isector);
    // every tree-product kernel is instantiated for the block size it runs with (mat = 2, 3, 4): a worm kernel with the extra moves built for
    // blocks of four states faulted on blocks of two (cause not found; the kernels of exact size retrace the oracle)
    if (C.xmoves) { if (nworm == 2) launch_sweep_worm_x2(mat, l, C, nsteps, resume); else launch_sweep_worm_x4(mat, l, C, nsteps, resume); }   // enter_sector: matrix trace path, sectors 2, 3, 4, 6 - 9
    else if (isector == 5 || isector >= 10) {                  // U-matrix factors beyond sector 3, up to twelve operators: kernels whose context holds twelve worm operators, both trace paths
        if (nworm == 4) launch_sweep_worm_big4(mat, l, C, nsteps, resume);
        else if (nworm == 6) launch_sweep_worm_big6(mat, l, C, nsteps, resume);
        else if (nworm == 8) launch_sweep_worm_big8(mat, l, C, nsteps, resume);
        else launch_sweep_worm_big12(mat, l, C, nsteps, resume);
    }
    else if (mat == 0) launch_sweep_worm_scalar(nworm, l, C, nsteps, resume);
    else if (mat <= 4) launch_sweep_worm_tree(nworm, mat, l, C, nsteps, resume);
    else launch_sweep_worm_seq(nworm, l, C, nsteps, resume);
}
void launch_meas_misc(int mat, const LaunchCfg &l, const DevConst &C)
{
    if (mat == 0) launch_meas_misc_scalar(l, C);
    else if (mat <= 4) launch_meas_misc_tree(mat, l, C);
    else launch_meas_misc_seq(l, C);
}
}  // namespace w2dyn

extern "C" {

int w2dyn_create(w2dyn_solver **out, int device)
{
    if (!out) return W2DYN_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return W2DYN_EHIP;
    w2dyn_solver *s = new w2dyn_solver();
    s->device = device;
    if (hipSetDevice(device) != hipSuccess || scratch_address(device, &s->scratch) != hipSuccess || hipStreamCreate(&s->stream) != hipSuccess ||
        hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { w2dyn_destroy(s); return W2DYN_EHIP; }      // frees what exists
    if (hipDeviceGetAttribute(&s->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || s->ncu <= 0) s->ncu = 256;
    *out = s;
    return 0;
}

int w2dyn_destroy(w2dyn_solver *s)
{
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    free_device(s);
    release_traps(s);
    if (s->packbuf) (void)hipFree(s->packbuf);
    if (s->packscale) (void)hipFree(s->packscale);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return 0;
}

const char *w2dyn_last_error(const w2dyn_solver *s) { return s ? s->err.c_str() : "null solver"; }

int w2dyn_set_layout(w2dyn_solver *s, int nwalkers, int64_t first_walker, int kmax, int walkers_per_bin)
{
    if (!s) return W2DYN_EINVAL;
    if (s->have_solver) return fail(s, W2DYN_EINVAL, "set_layout must precede init_solver");
    if (nwalkers <= 0 || nwalkers > (1 << 26) || first_walker < 0 || kmax < 0 || kmax > KMAX_LIMIT || walkers_per_bin <= 0)
        return fail(s, W2DYN_EINVAL, "bad layout (nwalkers=%d kmax=%d walkers_per_bin=%d)", nwalkers, kmax, walkers_per_bin);
    s->nwalkers = nwalkers; s->first_walker = first_walker; s->kmax_req = kmax; s->walkers_per_bin = walkers_per_bin;
    return 0;
}

int w2dyn_init_rnd(w2dyn_solver *s, int64_t nseed)
{
    if (!s) return W2DYN_EINVAL;
    s->seed = nseed; s->stream_pos.clear();      // a new seed starts every chain's stream at its beginning
    return 0;
}

int w2dyn_set_timing(w2dyn_solver *s, int on) { if (!s) return W2DYN_EINVAL; s->timing = on != 0; return 0; }

int w2dyn_init_paras(w2dyn_solver *s, const char *parastr)
{
    if (!s || !parastr) return W2DYN_EINVAL;
    s->paras.clear();
    const char *p = parastr;
    while (*p) {
        const char *eol = strchr(p, '\n');
        std::string line = eol ? std::string(p, eol) : std::string(p);
        p = eol ? eol + 1 : p + line.size();
        size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string key = line.substr(0, eq), val = line.substr(eq + 1);
        size_t hash = val.find('#'); if (hash != std::string::npos) val.resize(hash);
        auto trim = [](std::string &t) { size_t a = t.find_first_not_of(" \t\r"), b = t.find_last_not_of(" \t\r"); t = (a == std::string::npos) ? "" : t.substr(a, b - a + 1); };
        trim(key); trim(val);
        if (key.empty() || key[0] == '#') continue;
        s->paras.push_back({key, val});
    }
    s->have_paras = true;
    return 0;
}

int w2dyn_dest_paras(w2dyn_solver *s) { if (!s) return W2DYN_EINVAL; s->paras.clear(); s->have_paras = false; return 0; }
int w2dyn_set_fourpnt(w2dyn_solver *s, int fourpnt) { if (!s) return W2DYN_EINVAL; s->fourpnt = fourpnt; return 0; }

int w2dyn_init_solver(w2dyn_solver *s, const double *U, const double *ftau_full, const double *mu_full,
                      const double *screening, int iter_no, int nftau, int nbands)
{
    (void)iter_no;
    if (!s || !U || !ftau_full || !mu_full) return W2DYN_EINVAL;
    if (!s->have_paras) return fail(s, W2DYN_EINVAL, "init_paras must precede init_solver");
    HIPCHK(hipSetDevice(s->device));
    free_device(s);
    long long iv; double rv; std::string sv;
    // ---- modes this build refuses ----
    struct { const char *key; long long allowed; } zero_int[] = {
        {"segment", 0}, {"statesampling", 0}, {"Phonon", 0},
        {"Screening", 0}, {"Uw", 0}, {"MeasGtauDetRat", 0},
        {"MeasExpResDensityMatrix", 0},
        {"MeasDensityMatrix", 1}, {"Eigenbasis", 1}};
    for (auto &z : zero_int) {
        if (!get_int(s, z.key, iv)) return W2DYN_EINVAL;
        if (iv != z.allowed) return fail(s, W2DYN_EUNSUPPORTED, "%s = %lld is not supported by the HIP solver (only %lld)", z.key, iv, z.allowed);
    }
    // off-diagonal hybridisation (CTQMC.F90:461-479).  full_offdiag = 1 asks the reference to build the matrix anew at every pair
    // move, for debugging: the result is that of the fast updates up to rounding, which is what runs here either way
    if (!get_int(s, "offdiag", iv)) return W2DYN_EINVAL; s->offdiag = iv != 0;
    if (!get_int(s, "full_offdiag", iv)) return W2DYN_EINVAL;
    // flavourchange_moves (b_exch, CTQMC.F90:481-485): StepFlavourchange_general in place of the two-pair moves.  It gives two operators
    // independent new flavours, which only a hybridisation matrix over all flavours can take: offdiag = 1 only
    if (!get_int(s, "flavourchange_moves", iv)) return W2DYN_EINVAL; s->flavc = iv != 0;
    if (s->flavc && !s->offdiag) return fail(s, W2DYN_EUNSUPPORTED, "flavourchange_moves = 1 needs offdiag = 1 (the move leaves a flavour with unequal numbers of creators and annihilators)");
    // moves outside the default mix (CTQMC.F90:332-333): two pairs at once, pair moves that change the outer block
    if (!get_real(s, "Percentage4OperatorMove", s->p_4op)) return W2DYN_EINVAL;
    if (!get_real(s, "PercentageOuterMove", s->p_outer)) return W2DYN_EINVAL;
    if (s->p_4op < 0.0 || s->p_4op > 1.0 || s->p_outer < 0.0 || s->p_outer > 1.0)
        return fail(s, W2DYN_EINVAL, "Percentage4OperatorMove = %g, PercentageOuterMove = %g: probabilities", s->p_4op, s->p_outer);
    if (!get_str(s, "EnableOuterTruncation", sv)) return W2DYN_EINVAL;
    if (sv == "YES") return fail(s, W2DYN_EUNSUPPORTED, "EnableOuterTruncation = YES is not supported");
    if (s->fourpnt & ~15)
        return fail(s, W2DYN_EUNSUPPORTED, "FourPnt = %d is not supported (bits 1 = imaginary-time, 2 = Legendre, 4 = Matsubara two-particle Green's function in Z space, 8 = worm estimators)", s->fourpnt);
    if (!get_int(s, "N4tau", iv)) return W2DYN_EINVAL; s->n4tau = (int)iv;
    if (!get_int(s, "N4leg", iv)) return W2DYN_EINVAL; s->n4leg = (int)iv;
    if ((s->fourpnt & 3) && s->n4tau < 1) return fail(s, W2DYN_EINVAL, "N4tau=%d", s->n4tau);
    if ((s->fourpnt & 2) && (s->n4leg < 2 || s->n4leg > s->n4tau))       // G4leg is allocated with N4tau (CTQMC.F90:549) and filled up to N4leg
        return fail(s, W2DYN_EINVAL, "FourPnt & 2 needs 2 <= N4leg <= N4tau (N4leg=%d, N4tau=%d)", s->n4leg, s->n4tau);
    if (!get_real(s, "PercentageWormInsert", s->p_wins)) return W2DYN_EINVAL;
    if (!get_real(s, "PercentageWormReplace", s->p_wrep)) return W2DYN_EINVAL;
    if (!get_real(s, "WormEta", s->worm_eta)) return W2DYN_EINVAL;                      // CTQMC.F90:434
    if (!get_int(s, "WormMeasG4iw", iv)) return W2DYN_EINVAL; s->worm_meas_g4iw = iv != 0;
    if (!get_int(s, "WormMeasGiw", iv)) return W2DYN_EINVAL; s->worm_meas_giw = iv != 0;
    if (!get_int(s, "WormMeasGtau", iv)) return W2DYN_EINVAL; s->worm_meas_gtau = iv != 0;
    if (!get_int(s, "WormMeasGSigmaiw", iv)) return W2DYN_EINVAL; s->worm_meas_gsigma = iv != 0;
    {
        static const char *keys[6] = {"WormMeasP2iwPH", "WormMeasP2iwPP", "WormMeasP2tauPH", "WormMeasP2tauPP", "WormMeasP3iwPH", "WormMeasP3iwPP"};
        for (int q = 0; q < 6; ++q) { if (!get_int(s, keys[q], iv)) return W2DYN_EINVAL; s->worm_meas_eq[q] = iv != 0; }
        if (!get_int(s, "N2iwb", iv)) return W2DYN_EINVAL; s->n2iwb = (int)iv;
        if (!get_int(s, "N3iwf", iv)) return W2DYN_EINVAL; s->n3iwf = (int)iv;
        if (!get_int(s, "N3iwb", iv)) return W2DYN_EINVAL; s->n3iwb = (int)iv;
    }
    {
        static const char *keys[11] = {"WormMeasH4iw", "WormMeasQQ", "WormMeasQQtau", "WormMeasQQQQ", "WormMeasNQQdag", "WormMeasQQdd", "WormMeasUcaca", "WormMeasUcacatau",
                                       "WormMeasUccaa", "WormMeasUccaatau", "WormMeasQUDdag"};
        for (int q = 0; q < 11; ++q) { if (!get_int(s, keys[q], iv)) return W2DYN_EINVAL; s->worm_meas_more[q] = iv != 0; }
        if (!(s->fourpnt & 8)) s->worm_meas_more[0] = s->worm_meas_more[3] = 0;         // CTQMC.F90:379, 390
    }
    if (!get_int(s, "G4ph", iv)) return W2DYN_EINVAL; s->g4ph = iv != 0;
    if (!get_int(s, "G4pp", iv)) return W2DYN_EINVAL; s->g4pp = iv != 0;
    if (!get_int(s, "WormPHConvention", iv)) return W2DYN_EINVAL; s->wormphconv = iv != 0;
    if (!((s->fourpnt & 8) && s->worm_meas_g4iw)) s->g4ph = s->g4pp = 0;                 // CTQMC.F90:339-363
    if (!get_int(s, "N4iwf", iv)) return W2DYN_EINVAL; s->n4iwf = (int)iv;
    if (!get_int(s, "N4iwb", iv)) return W2DYN_EINVAL; s->n4iwb = (int)iv;
    if (!get_int(s, "MeasG2iw", iv)) return W2DYN_EINVAL; s->meas_g2iw = (int)iv;
    if (!get_int(s, "MeasG4iwPP", iv)) return W2DYN_EINVAL; s->meas_g4iwpp = (int)iv;
    if (!get_int(s, "ZPHConvention", iv)) return W2DYN_EINVAL; s->zph = iv != 0;
    if ((s->fourpnt & 12) && (s->n4iwf < 1 || s->n4iwb < 0)) return fail(s, W2DYN_EINVAL, "N4iwf=%d N4iwb=%d", s->n4iwf, s->n4iwb);
    if ((s->fourpnt & 4) && s->meas_g4iwpp && s->zph) return fail(s, W2DYN_EUNSUPPORTED, "G4iwPP measurement is not implemented for ZPHConvention != 0 (CTQMC.F90:2220)");
    if (screening) { /* must be zero; checked below once nd is known */ }

    if (!get_int(s, "Nd", iv)) return W2DYN_EINVAL; s->nd = (int)iv;
    if (nbands != s->nd) return fail(s, W2DYN_EINVAL, "nbands=%d differs from Nd=%d", nbands, s->nd);
    if (s->nd < 1 || 2 * s->nd > MAXFLAV) return fail(s, W2DYN_EUNSUPPORTED, "Nd=%d out of range", s->nd);
    s->nflav = 2 * s->nd; s->nstates = 1 << (2 * s->nd);
    if (!get_real(s, "beta", s->beta)) return W2DYN_EINVAL;
    if (!get_int(s, "Ntau", iv)) return W2DYN_EINVAL; s->ntau = (int)iv;
    if (!get_int(s, "NLegMax", iv)) return W2DYN_EINVAL; s->nleg = (int)iv;
    if (!get_int(s, "Niw", iv)) return W2DYN_EINVAL; s->niw = (int)iv;
    if (!get_int(s, "MeasGiw", iv)) return W2DYN_EINVAL; s->meas_giw = (int)iv;
    // NLookup_nfft (configspec:71, default 10000) only sizes the reference's NFFT lookup buffer (CTQMC.F90:578-595,
    // Nfft_z.F90:205-278); G(iw) is the exact sum here whatever its value (it differs from the NFFT result by the NFFT
    // truncation error, ~1e-12), so the key is read and otherwise ignored
    if (s->meas_giw) { if (!get_int(s, "NLookup_nfft", iv)) return W2DYN_EINVAL; }
    if (!get_int(s, "NCorr", iv)) return W2DYN_EINVAL; s->ncorr = (int)iv;
    if (!get_int(s, "Nmeas", iv)) return W2DYN_EINVAL; s->nmeas = iv;
    if (!get_int(s, "AccumGtau", iv)) return W2DYN_EINVAL;
    if (iv < 0 || iv > 2) return fail(s, W2DYN_EINVAL, "invalid value %lld for AccumGtau (0, 1, 2)", iv);     // CTQMC.F90:502
    s->accum_gtau = (int)iv;
    s->step_series = 0;                  // CTQMC.F90:510-512
    if (!get_int(s, "Gtau_mean_step", iv)) return W2DYN_EINVAL; if (iv != 0) s->step_series |= 1;
    if (!get_int(s, "Gtau_mid_step", iv)) return W2DYN_EINVAL; if (iv != 0) s->step_series |= 2;
    if (!get_int(s, "sign_step", iv)) return W2DYN_EINVAL; if (iv != 0) s->step_series |= 4;
    if (!get_int(s, "MeasSsquare", iv)) return W2DYN_EINVAL;
    s->meas_ssq = iv != 0;               // CTQMC.F90:8009-8010
    if (!get_int(s, "MeasSusz", iv)) return W2DYN_EINVAL;
    s->meas_susz = iv == 1;              // CTQMC.F90:534-540 (matrix solver: b_meas_susz_mat)
    if (!get_int(s, "MaxHisto", iv)) return W2DYN_EINVAL; s->maxhisto = (int)iv;
    if (!get_real(s, "PercentageGlobalMove", s->p_glob)) return W2DYN_EINVAL;
    if (!get_real(s, "PercentageTauShiftMove", s->p_shift)) return W2DYN_EINVAL;
    if (!get_real(s, "TaudiffMax", s->taudiffmax_param)) return W2DYN_EINVAL;
    if (!get_str(s, "QuantumNumbers", sv)) return W2DYN_EINVAL;
    s->has_szt = sv.find("Szt") != std::string::npos;
    const std::string qns = sv;
    double epsevec = 1e-15;
    if (!get_real(s, "EPSEVEC", epsevec)) return W2DYN_EINVAL;
    if (s->nleg < 1 || s->nleg > 128) return fail(s, W2DYN_EUNSUPPORTED, "NLegMax=%d out of range [1,128]", s->nleg);
    s->nftau = nftau;
    if (nftau < 2) return fail(s, W2DYN_EINVAL, "nftau=%d", nftau);

    const int nd = s->nd, nf = s->nflav;
    // ---- local problem (CTQMC.F90:7957-8031) ----
    bool dd = true;        // H_loc diagonal in the occupation basis (density-density U, diagonal muimp)
    for (int a = 0; a < nf && dd; ++a) for (int b = 0; b < nf && dd; ++b) for (int c = 0; c < nf && dd; ++c) for (int d = 0; d < nf; ++d) {
        double u = U[((a * nf + b) * nf + c) * nf + d];
        if (u == 0.0) continue;
        bool diag = (a == c && b == d) || (a == d && b == c);
        if (!diag && a != b && c != d) { dd = false; break; }
    }
    for (int b1 = 0; b1 < nd; ++b1) for (int s1 = 0; s1 < 2; ++s1) for (int b2 = 0; b2 < nd; ++b2) for (int s2 = 0; s2 < 2; ++s2) {
        size_t i = ((size_t)(b1 * 2 + s1) * nd + b2) * 2 + s2;
        if ((b1 != b2 || s1 != s2) && mu_full[i] != 0.0) dd = false;
        if (screening) for (int t = 0; t < nftau; ++t) if (screening[i * nftau + t] != 0.0) return fail(s, W2DYN_EUNSUPPORTED, "non-zero screening function is not supported");
    }
    double epsblock = 1e-13;
    if (!get_real(s, "EPSBLOCK", epsblock)) return W2DYN_EINVAL;
    if (qns.find("All") != std::string::npos && nd > 5)
        return fail(s, W2DYN_EUNSUPPORTED, "QuantumNumbers = All (automatic partitioning) needs the dense 4^Nd x 4^Nd H_loc: Nd = %d > 5", nd);
    s->lp.build(nd, U, mu_full, qns, epsevec, epsblock);
    const LocalProblem &lp = s->lp;
    // one-state blocks in Fock-state order: only the diagonal of H_loc survives (terms between blocks are dropped,
    // Operator.F90:521, 1083) and the closed-form trace with eps / udd below is exact, whatever U and muimp hold
    if (s->offdiag) {      // what the off-diagonal sampler does not do (the reference measures gtau_full / gleg_full only and skips G(iw), CTQMC.F90:8562-8575)
        long long a = 0, b = 0, c = 0, d = 0, e = 0;
        if (!get_int(s, "MeasSusz", a) || !get_int(s, "AccumGtau", b) || !get_int(s, "Gtau_mean_step", c) || !get_int(s, "Gtau_mid_step", d) || !get_int(s, "sign_step", e)) return W2DYN_EINVAL;
        if (s->fourpnt != 0 || a != 0 || b == 2 || c != 0 || d != 0 || e != 0 || (s->p_4op > 0.0 && !s->flavc) || s->p_outer > 0.0)
            return fail(s, W2DYN_EUNSUPPORTED, "offdiag = 1 runs the pair, shift and global moves and measures gtau_full / gleg_full, sign, histograms and the density matrix: FourPnt, MeasSusz, AccumGtau = 2, the step series, PercentageOuterMove and Percentage4OperatorMove without flavourchange_moves = 1 are not supported with it");
        s->meas_giw = 0;
    }
    // worm sampling together with the two-pair / outer-block moves: the worm kernels make those moves on the operator list only
    const bool worm_xmoves = s->p_wins > 0.0 && (s->p_4op > 0.0 || s->p_outer > 0.0);
    s->mat = !lp.identity_blocks || s->offdiag || worm_xmoves;      // off-diagonal hybridisation: always the matrix trace path (list of operators with their flavours)
    (void)dd;
    if (s->mat && lp.nsmax > 16) return fail(s, W2DYN_EUNSUPPORTED, "largest block of H_loc has %d states with QuantumNumbers '%s': the HIP solver handles up to 16", lp.nsmax, qns.c_str());
    s->eps.assign(nf, 0.0); s->udd.assign((size_t)nf * nf, 0.0);
    for (int f = 0; f < nf; ++f) { int b = f >> 1, sp = f & 1; s->eps[f] = mu_full[((size_t)(b * 2 + sp) * nd + b) * 2 + sp]; }
    auto Ut = [&](int a, int b) { return U[((a * nf + b) * nf + a) * nf + b] - U[((a * nf + b) * nf + b) * nf + a]; };
    for (int f = 0; f < nf; ++f) for (int g = 0; g < nf; ++g) if (f != g) s->udd[f * nf + g] = 0.5 * (Ut(f, g) + Ut(g, f));
    s->energies = lp.eval;                                                    // eigenvalues in block order
    s->egs = lp.egs;                                                          // Trace.F90:576-583
    const int nouter = lp.nsst;
    s->dmoff.assign(nouter + 1, 0);
    for (int b = 0; b < nouter; ++b) s->dmoff[b + 1] = s->dmoff[b] + lp.sst_n[b] * lp.sst_n[b];
    {   // empty-trace weights of the outer blocks (Trace.F90:648-663): sum over the block's states
        s->outer_w.assign(nouter, 0.0); s->outer_cum.assign(nouter, 0.0);
        double sum = 0.0;
        for (int b = 0; b < nouter; ++b) {
            double w = 0.0;
            for (int i = 0; i < lp.sst_n[b]; ++i) w += std::exp(-(lp.eval[lp.sst_off[b] + i] - s->egs) * s->beta);
            s->outer_w[b] = w; sum += w;
        }
        double acc = 0.0;
        for (int b = 0; b < nouter; ++b) { s->outer_w[b] /= sum; acc += s->outer_w[b]; s->outer_cum[b] = acc; }
    }
    s->ftab.assign((size_t)nf * nftau, 0.0);
    for (int b = 0; b < nd; ++b) for (int sp = 0; sp < 2; ++sp)      // CTQMC.F90:7950-7955
        memcpy(&s->ftab[(size_t)(2 * b + sp) * nftau], ftau_full + ((((size_t)b * 2 + sp) * nd + b) * 2 + sp) * nftau, sizeof(double) * nftau);

    // ---- device state ----
    DevConst &C = s->C;
    C = DevConst{};
    C.nd = nd; C.nflav = nf; C.nstates = s->nstates; C.nftau = nftau; C.ntau = s->ntau; C.nleg = s->nleg; C.niw2 = 2 * s->niw;
    int kmax = s->kmax_req;
    if (kmax == 0) { kmax = (int)(0.4 * s->beta) + 24; kmax = (kmax + 7) & ~7; if (kmax > KMAX_LIMIT) kmax = KMAX_LIMIT; }   // overflow is detected and reported
    if (s->offdiag) kmax = std::min(kmax * nf, 128);      // all flavours share the slots of flavour 0 (no growth later: w2dyn_set_layout)
    C.nwalkers = s->nwalkers; C.has_szt = s->has_szt; C.maxhisto = s->maxhisto;
    C.walkers_per_bin = s->walkers_per_bin; C.nbins = (s->nwalkers + s->walkers_per_bin - 1) / s->walkers_per_bin;
    { int G = 1, sh = 6; while (G < nf) { G <<= 1; --sh; } C.lane_shift = sh; }
    C.nouter = nouter; C.dmsize = s->dmoff[nouter]; C.mat = s->mat ? 1 : 0; C.nsst = lp.nsst; C.nsm = lp.nsmax;
    std::vector<double> opmat;
    if (s->mat) {
        const size_t nop = (size_t)lp.nsst * nf * 2 * C.nsm * C.nsm;
        opmat.assign(nop, 0.0);
        for (int b = 0; b < lp.nsst; ++b) for (int f = 0; f < nf; ++f) for (int ca = 1; ca <= 2; ++ca) {
            int tb = lp.connect(b, f, ca); if (tb < 0) continue;
            const std::vector<double> &O = lp.oper(b, f, ca); const int n1 = lp.sst_n[b], n2 = lp.sst_n[tb];
            double *dst = &opmat[((size_t)(b * nf + f) * 2 + (ca - 1)) * C.nsm * C.nsm];
            for (int i = 0; i < n2; ++i) for (int j = 0; j < n1; ++j) dst[i * C.nsm + j] = O[i + (size_t)n2 * j];
        }
        if (lp.nsst > 32767) return fail(s, W2DYN_EUNSUPPORTED, "%d blocks: more than the 16-bit block tables hold", lp.nsst);
    }
    std::vector<double> opmat_pad;
    if (s->mat && C.nsm > 4) {          // the sequential evaluator reads whole rows of a fixed length: stride mat_seq_pad(nsm)
        const int nt = mat_seq_pad(C.nsm);
        opmat_pad.assign(((size_t)lp.nsst * nf * 2 + 1) * nt * nt, 0.0);       // one more: the unit matrix (fills groups of operators, mat_trace_seq_core)
        for (int i = 0; i < nt; ++i) opmat_pad[((size_t)lp.nsst * nf * 2 * nt + i) * nt + i] = 1.0;
        for (size_t m = 0; m < (size_t)lp.nsst * nf * 2; ++m)
            for (int i = 0; i < C.nsm; ++i) for (int j = 0; j < C.nsm; ++j)
                opmat_pad[(m * nt + i) * nt + j] = opmat[(m * C.nsm + i) * C.nsm + j];
    }
    { int rc0 = configure_kmax(s, kmax); if (rc0) return rc0; }
    C.beta = s->beta; C.rbeta = 1.0 / s->beta; C.p_glob = s->p_glob; C.p_shift = s->p_shift; C.egs = s->egs;
    C.p_4op = s->p_4op; C.p_outer = s->p_outer; C.xmoves = (!s->offdiag && (s->p_4op > 0.0 || s->p_outer > 0.0)) ? 1 : 0;
    C.offdiag = s->offdiag ? 1 : 0;
    C.seed = (unsigned long long)s->seed; C.first_walker = (unsigned long long)s->first_walker;
    host_init_window(s, s->taudiffmax_param, C.num_windows, C.taudiff_max);           // Trace.F90:605
    C.nslide = std::max(2 * nd, (int)std::floor((double)s->ncorr / (double)C.num_windows));   // CTQMC.F90:8069
    int rc;
    if ((rc = dev_upload(s, C.ftab, s->ftab))) return rc;
    {   // flavours with the same F(tau) table, to the last bit (step_glob carries the inverse matrices along a relabelling between them)
        std::vector<int> same(nf, 0);
        for (int f = 0; f < nf; ++f)
            for (int g = 0; g < nf; ++g)
                if (!memcmp(&s->ftab[(size_t)f * s->nftau], &s->ftab[(size_t)g * s->nftau], (size_t)s->nftau * sizeof(double))) same[f] |= 1 << g;
        const int *ds = nullptr; if ((rc = dev_upload(s, ds, same))) return rc; C.fsame = ds;
    }
    if ((rc = dev_upload(s, C.energies, s->energies))) return rc;
    if ((rc = dev_upload(s, C.udd, s->udd))) return rc;
    if ((rc = dev_upload(s, C.eps, s->eps))) return rc;
    if ((rc = dev_upload(s, C.outer_cum, s->outer_cum))) return rc;
    if ((rc = dev_upload(s, C.outer_w, s->outer_w))) return rc;
    if ((rc = dev_upload(s, C.sst_n, lp.sst_n))) return rc;
    if ((rc = dev_upload(s, C.sst_off, lp.sst_off))) return rc;
    if ((rc = dev_upload(s, C.conn, lp.conn))) return rc;
    if ((rc = dev_upload(s, C.shift, lp.shift))) return rc;
    if ((rc = dev_upload(s, C.sst_of_state, lp.sst_of_state))) return rc;
    if ((rc = dev_upload(s, C.dmoff, s->dmoff))) return rc;
    if ((rc = dev_upload(s, C.sst_eval, lp.eval))) return rc;
    {
        std::vector<int> first(lp.nsst);
        for (int b = 0; b < lp.nsst; ++b) first[b] = lp.states[lp.sst_off[b]];
        if ((rc = dev_upload(s, C.sst_first, first))) return rc;
    }
    if (s->mat) { if ((rc = dev_upload(s, C.opmat, opmat))) return rc; }
    if (!opmat_pad.empty()) { if ((rc = dev_upload(s, C.opmat_pad, opmat_pad))) return rc; }
    if (!opmat_pad.empty() && !getenv("W2DYN_NO_CHECKPOINTS")) {      // checkpoints of the window part of the sequential trace
        const int nt = mat_seq_pad(C.nsm);
        if ((rc = dev_alloc(s, C.kcache, (size_t)s->nwalkers * MAT_CK_ENTRIES * mat_ck_doubles(nt)))) return rc;
    }
    const size_t W = s->nwalkers;
    if ((rc = dev_alloc(s, C.ws, W))) return rc;
    if ((rc = dev_alloc(s, C.K, W * nf))) return rc;
    if ((rc = dev_alloc(s, C.TC, W * nf * kmax))) return rc;
    if ((rc = dev_alloc(s, C.TA, W * nf * kmax))) return rc;
    if ((rc = dev_alloc(s, C.TS, W * nf * 2 * kmax))) return rc;
    if ((rc = dev_alloc(s, C.M, W * nf * kmax * kmax))) return rc;
    if ((rc = dev_alloc(s, C.MB, W * nf * kmax * kmax))) return rc;
    if ((rc = dev_alloc(s, C.cfgsign, W))) return rc;
    if ((rc = dev_alloc(s, C.flags, 4))) return rc;
    if (s->offdiag) {      // HybrF_full as handed over (CTQMC.F90:7950-7955), the flavours of the slots, the full-matrix accumulators
        std::vector<double> ff(ftau_full, ftau_full + (size_t)nf * nf * nftau);
        if ((rc = dev_upload(s, C.ffull, ff))) return rc;
        {   // classes of identical two-flavour tables (step_glob_od: a relabelling that maps every table onto an identical one leaves the matrix as it is)
            std::vector<int> cls((size_t)nf * nf, 0);
            for (int e = 0; e < nf * nf; ++e) {
                cls[e] = e;
                for (int d = 0; d < e; ++d) if (!memcmp(&ff[(size_t)e * nftau], &ff[(size_t)d * nftau], (size_t)nftau * sizeof(double))) { cls[e] = d; break; }
            }
            const int *dc = nullptr; if ((rc = dev_upload(s, dc, cls))) return rc; C.fclass = dc;
        }
        if ((rc = dev_alloc(s, C.FC, W * kmax))) return rc;
        if ((rc = dev_alloc(s, C.FA, W * kmax))) return rc;
        HIPCHK(hipMemsetAsync(C.FC, 0, sizeof(int) * W * kmax, s->stream)); HIPCHK(hipMemsetAsync(C.FA, 0, sizeof(int) * W * kmax, s->stream));
        if ((rc = dev_alloc(s, s->acc_gleg_full, (size_t)C.nbins * nf * nf * s->nleg))) return rc;
        if ((rc = dev_alloc(s, s->acc_gtau_full, (size_t)C.nbins * nf * nf * s->ntau))) return rc;
    } else { s->acc_gleg_full = s->acc_gtau_full = nullptr; C.fclass = nullptr; }
    const size_t B = C.nbins;
    if ((rc = dev_alloc(s, C.acc_gleg, B * nf * s->nleg))) return rc;
    {   // prefix sums, workgroup totals, ticket of k_gleg_segments (layout: tu_meas.hip)
        const size_t bpb = GLEG_SEG_BLOCK / nf, n = B * nf + 1 + (B + bpb - 1) / bpb + 1;
        if ((rc = dev_alloc(s, s->gleg_segpre, n))) return rc;
        HIPCHK(hipMemsetAsync(s->gleg_segpre, 0, sizeof(unsigned long long) * n, s->stream));
    }
    if ((rc = dev_alloc(s, s->gleg_leg, B * nf * s->nleg))) return rc;
    if ((rc = dev_alloc(s, s->gleg_tab, (size_t)s->nleg * s->nleg))) return rc;
    {   // Legendre polynomials in Chebyshev polynomials: P_l = sum_k g_k g_(l-k) T_|l-2k|, g_k = binom(2k, k) / 4^k
        const int nl = s->nleg;
        std::vector<long double> g(nl + 1, 1.0L);
        for (int k = 1; k <= nl; ++k) g[k] = g[k - 1] * (long double)(2 * k - 1) / (long double)(2 * k);
        std::vector<double> tab((size_t)nl * nl, 0.0);
        for (int l = 0; l < nl; ++l)
            for (int k = 0; 2 * k <= l; ++k) tab[(size_t)l * nl + (l - 2 * k)] = (double)((2 * k == l ? 1.0L : 2.0L) * g[k] * g[l - k]);
        HIPCHK(hipMemcpy(s->gleg_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if ((rc = dev_alloc(s, C.acc_gtau, B * nf * s->ntau))) return rc;
    if ((rc = dev_alloc(s, C.acc_giw, B * nf * C.niw2 * 2))) return rc;
    if ((rc = dev_alloc(s, C.acc_histo, B * nf * (s->maxhisto + 1)))) return rc;
    if ((rc = dev_alloc(s, C.acc_dm, B * C.dmsize))) return rc;
    if ((rc = dev_alloc(s, C.acc_outer, B * nouter))) return rc;
    if ((rc = dev_alloc(s, C.acc_signouter, B * nouter))) return rc;
    if ((rc = dev_alloc(s, C.acc_sign, B))) return rc;
    if ((rc = dev_alloc(s, C.acc_signstate, B * s->nstates))) return rc;
    if ((rc = dev_alloc(s, C.acc_tcstate, B * s->nstates))) return rc;
    if ((rc = dev_alloc(s, C.accpair, (size_t)nf * s->ntau))) return rc;
    s->accpair_default = C.accpair; C.naccpair = s->ntau; C.accpair_max = s->beta; C.calib = 0;
    s->accpairtau.clear(); s->accpairtau_chains.clear();
    if ((rc = dev_alloc(s, C.acc_nz, B))) return rc;
    if ((rc = dev_alloc(s, C.acc_nw, B))) return rc;
    C.nwf = s->n4iwf; C.nwb = s->n4iwb; C.wormphconv = s->wormphconv; C.g4ph = s->g4ph; C.g4pp = s->g4pp;
    C.p_wins = 0.0; C.p_wrep = s->p_wrep; C.worm_eta = s->worm_eta; C.isector = 1; C.icomponent = 0;
    { int *qp = nullptr; if ((rc = dev_alloc(s, qp, 1))) return rc; C.queue = qp; }
    {   // the full U matrix: proposal factor of the GSigma worm (sector 3)
        s->umat_h.assign(U, U + (size_t)nf * nf * nf * nf);
        const double *du = nullptr; if ((rc = dev_upload(s, du, s->umat_h))) return rc; C.umat = du;
    }      // chain queue of the matrix-path sweep kernels
    {   // user symmetry moves of the global move: the listed permutations and their inverses (Trace.F90:471-478)
        if (!get_int(s, "NSymMove", iv)) return W2DYN_EINVAL;
        const int nsym = (int)iv;
        if (nsym < 0 || nsym > 99) return fail(s, W2DYN_EINVAL, "NSymMove = %d", nsym);
        C.nsym = nsym; C.symmoves = nullptr;
        if (nsym > 0) {
            std::vector<int> tab((size_t)2 * nsym * nf, 0);
            for (int i = 0; i < nsym; ++i) {
                char key[16]; snprintf(key, sizeof key, "SymMove%02d", i + 1);
                if (!get_str(s, key, sv)) return W2DYN_EINVAL;
                std::vector<int> row; { const char *p = sv.c_str(); char *e; for (long v = strtol(p, &e, 10); e != p; p = e, v = strtol(p, &e, 10)) row.push_back((int)v); }
                std::vector<char> seen(nf + 1, 0);
                if ((int)row.size() != nf) return fail(s, W2DYN_EINVAL, "%s: %zu entries, %d flavours", key, row.size(), nf);
                for (int j = 0; j < nf; ++j) {
                    if (row[j] < 1 || row[j] > nf || seen[row[j]]) return fail(s, W2DYN_EINVAL, "%s is not a permutation of 1..%d", key, nf);
                    seen[row[j]] = 1;
                    tab[(size_t)i * nf + j] = row[j]; tab[(size_t)(nsym + i) * nf + row[j] - 1] = j + 1;
                }
            }
            const int *dtab = nullptr;
            if ((rc = dev_upload(s, dtab, tab))) return rc;
            C.symmoves = dtab;
        }
    }
    s->nbw = 0; s->cur_isector = 1; s->no_sampling = false; s->steps_worm_chains.clear();
    s->ac_dat = s->ac_out = nullptr; s->ac_lev = 0; s->ac_n = 0;
    if (s->accum_gtau == 2) {        // autocorracc_create(ncomp, ilog2(Nmeas)), CTQMC.F90:497-500, Accumulators.F90:89-101
        int lev = -1; for (long long t = s->nmeas; t > 0; t /= 2) ++lev;
        if (lev < 1) return fail(s, W2DYN_EINVAL, "AccumGtau = 2 needs Nmeas >= 2 (Nmeas = %lld)", (long long)s->nmeas);
        const size_t n = W * (size_t)nf * lev * 3 * s->ntau;
        if (n * sizeof(double) > ((size_t)64 << 30)) return fail(s, W2DYN_EUNSUPPORTED, "AccumGtau = 2: %zu bytes of per-chain accumulator levels; use fewer chains or tau bins", n * sizeof(double));
        if ((size_t)s->ntau * sizeof(double) > 64 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "AccumGtau = 2: Ntau = %d too large", s->ntau);
        s->ac_lev = lev;
        if ((rc = dev_alloc(s, s->ac_dat, n))) return rc;
        if ((rc = dev_alloc(s, s->ac_out, B * (size_t)nf * lev * s->ntau))) return rc;
    }
    s->acc_susz = nullptr; s->nbs = 0;
    s->ser = nullptr;
    if (s->step_series && s->nmeas > 0) { if ((rc = dev_alloc(s, s->ser, (size_t)(2 * nf + 1) * (size_t)s->nmeas))) return rc; }
    if (s->meas_ssq && !s->lp.build_s2())
        return fail(s, W2DYN_EUNSUPPORTED, "MeasSsquare = 1 needs blocks that S^2 does not leave (quantum numbers '%s' do not conserve the total spin: "
                                           "the reference's result is undefined there)", qns.c_str());
    if (s->meas_susz) {
        // meas_ntau_n0 walks stored bra / ket vectors of the eigenbasis (Trace.F90:6472-6835); with one state per block the
        // densities are numbers and the estimator is an occupation count, which is what k_meas_susz does
        // the matrix trace path (k_meas_susz_mat) inserts the densities into kets and bras kept per operator position
        if (s->mat && lp.nsmax > 8) return fail(s, W2DYN_EUNSUPPORTED, "MeasSusz = 1 on the matrix trace path handles blocks of up to 8 states (largest block here: %d)", lp.nsmax);
        if (s->ntau > 2048) return fail(s, W2DYN_EUNSUPPORTED, "MeasSusz = 1: Ntau = %d too large (at most 2048)", s->ntau);
        const size_t per = (size_t)nf * nf * s->ntau * sizeof(double);
        s->nbs = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / per));
        if ((rc = dev_alloc(s, s->acc_susz, (size_t)s->nbs * nf * nf * s->ntau))) return rc;
        s->susz_work = nullptr; s->susz_dens = nullptr;
        if (s->mat) {
            // N[block][f] = Op(c+_f : block' -> block) Op(c_f : block -> block') in the eigenbasis, zero where c_f annihilates the block
            const int nsm = C.nsm;
            std::vector<double> dens((size_t)lp.nsst * nf * nsm * nsm, 0.0);
            for (int b = 0; b < lp.nsst; ++b) for (int f = 0; f < nf; ++f) {
                const int tb = lp.connect(b, f, 2);
                if (tb < 0 || lp.connect(tb, f, 1) != b) continue;
                const double *A = &opmat[((size_t)(b * nf + f) * 2 + 1) * nsm * nsm];        // c_f : b -> tb, [target][source]
                const double *Cr = &opmat[((size_t)(tb * nf + f) * 2 + 0) * nsm * nsm];      // c+_f : tb -> b
                double *N = &dens[(size_t)(b * nf + f) * nsm * nsm];
                for (int i = 0; i < nsm; ++i) for (int m = 0; m < nsm; ++m) { double a = 0.0; for (int k = 0; k < nsm; ++k) a += Cr[i * nsm + k] * A[k * nsm + m]; N[i * nsm + m] = a; }
            }
            if ((rc = dev_upload(s, s->susz_dens, dens))) return rc;
            const size_t wdoubles = (size_t)W * 4 * (C.wcap + 1) * (nsm * nsm + 1);
            if (wdoubles * sizeof(double) > ((size_t)32 << 30)) return fail(s, W2DYN_EUNSUPPORTED, "MeasSusz = 1 on the matrix trace path: %zu bytes of per-chain bra / ket vectors; use fewer chains", wdoubles * sizeof(double));
            if ((rc = dev_alloc(s, s->susz_work, wdoubles))) return rc;
        }
    }
    s->acc_g4tau = s->acc_g4leg = nullptr; s->nbl = 0;
    if (s->fourpnt & 3) {
        const size_t pt = (s->fourpnt & 1) ? (size_t)nf * nf * s->n4tau * s->n4tau * s->n4tau : 0;
        const size_t pl = (s->fourpnt & 2) ? (size_t)nf * nf * s->n4leg * s->n4leg * (2 * s->n4iwb + 1) * 2 : 0;
        s->nbl = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / std::max<size_t>((pt + pl) * sizeof(double), 1)));
        if (pt) { if ((rc = dev_alloc(s, s->acc_g4tau, (size_t)s->nbl * pt))) return rc; }
        if (pl) { if ((rc = dev_alloc(s, s->acc_g4leg, (size_t)s->nbl * pl))) return rc; }
    }
    C.acc_giww = C.acc_gtauw = nullptr;
    if (s->worm_meas_giw || s->worm_meas_gsigma) { if ((rc = dev_alloc(s, C.acc_giww, B * (size_t)C.niw2 * 2))) return rc; }
    if (s->worm_meas_gtau) {
        if ((rc = dev_alloc(s, C.acc_gtauw, B * (size_t)s->ntau))) return rc;
        if ((size_t)s->ntau * sizeof(double) > 64 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "Ntau=%d too large for the gtau_worm estimator", s->ntau);
    }
    s->acc_weq_iw = s->acc_weq_tau = s->acc_weq_p3 = nullptr; s->nbe = 0;
    {   // one sector is sampled per run: the ph and pp estimators of a kind share their accumulator
        const bool iw = s->worm_meas_eq[0] || s->worm_meas_eq[1], tau = s->worm_meas_eq[2] || s->worm_meas_eq[3], p3 = s->worm_meas_eq[4] || s->worm_meas_eq[5];
        if (iw || tau || p3) {
            if (s->n2iwb < 0 || s->n3iwf < 1 || s->n3iwb < 0) return fail(s, W2DYN_EINVAL, "N2iwb=%d N3iwf=%d N3iwb=%d", s->n2iwb, s->n3iwf, s->n3iwb);
            const size_t per = ((iw ? (size_t)2 * (2 * s->n2iwb + 1) : 0) + (tau ? (size_t)s->ntau : 0) + (p3 ? (size_t)4 * s->n3iwf * (2 * s->n3iwb + 1) : 0)) * sizeof(double);
            s->nbe = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / per));
            if (iw) { if ((rc = dev_alloc(s, s->acc_weq_iw, (size_t)s->nbe * 2 * (2 * s->n2iwb + 1)))) return rc; }
            if (tau) { if ((rc = dev_alloc(s, s->acc_weq_tau, (size_t)s->nbe * s->ntau))) return rc; }
            if (p3) { if ((rc = dev_alloc(s, s->acc_weq_p3, (size_t)s->nbe * 4 * s->n3iwf * (2 * s->n3iwb + 1)))) return rc; }
        }
    }
    s->acc_wie_iw = s->acc_wie_tau = nullptr; s->nbie = 0; s->wie_n = 0;
    {   // worm sectors 5, 10 - 16: the largest complex array asked for, and a tau histogram
        const int *m = s->worm_meas_more;
        const size_t n4 = (size_t)4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1) * 2;
        const size_t sizes[11] = {n4, (size_t)4 * s->niw, 0, n4, (size_t)4 * s->n3iwf * (2 * s->n3iwb + 1), (size_t)8 * s->n3iwf * s->n3iwf, (size_t)2 * (2 * s->n2iwb + 1), 0,
                                  (size_t)2 * (2 * s->n2iwb + 1), 0, (size_t)4 * s->niw};
        bool any = false, tau = m[2] || m[7] || m[9];
        for (int q = 0; q < 11; ++q) if (m[q]) { any = true; s->wie_n = std::max(s->wie_n, sizes[q]); }
        if (any) {
            if ((m[0] || m[3]) && (s->n4iwf < 1 || s->n4iwb < 0)) return fail(s, W2DYN_EINVAL, "N4iwf=%d N4iwb=%d", s->n4iwf, s->n4iwb);
            if ((m[4] || m[5]) && (s->n3iwf < 1 || s->n3iwb < 0)) return fail(s, W2DYN_EINVAL, "N3iwf=%d N3iwb=%d", s->n3iwf, s->n3iwb);
            if ((m[6] || m[8]) && s->n2iwb < 0) return fail(s, W2DYN_EINVAL, "N2iwb=%d", s->n2iwb);
            const size_t per = (s->wie_n + (tau ? (size_t)s->ntau : 0)) * sizeof(double);
            s->nbie = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / std::max<size_t>(per, 1)));
            if (s->wie_n) { if ((rc = dev_alloc(s, s->acc_wie_iw, (size_t)s->nbie * s->wie_n))) return rc; }
            if (tau) { if ((rc = dev_alloc(s, s->acc_wie_tau, (size_t)s->nbie * s->ntau))) return rc; }
        }
    }
    if (s->g4ph || s->g4pp) {
        const size_t n = (size_t)4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1) * 2 * ((s->g4ph ? 1 : 0) + (s->g4pp ? 1 : 0));
        s->nbw = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / (n * sizeof(double))));
        const size_t one = (size_t)s->nbw * 4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1) * 2;
        if (s->g4ph) { if ((rc = dev_alloc(s, C.acc_g4w, one))) return rc; }
        if (s->g4pp) { if ((rc = dev_alloc(s, C.acc_g4wpp, one))) return rc; }
        if ((size_t)(4 * s->n4iwf + 2 * s->n4iwb + 1) * sizeof(double2) > 64 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "N4iwf=%d N4iwb=%d too large for the worm estimator", s->n4iwf, s->n4iwb);
    }
    s->G2 = s->acc4 = s->acc4pp = s->acc2 = nullptr; s->nb4 = 0;
    if (s->fourpnt & 4) {
        const size_t n2 = 2 * (size_t)(s->n4iwf + s->n4iwb), n4 = (size_t)nf * nf * 4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1), ng2 = (size_t)nf * n2 * n2;
        const size_t g2all = W * nf * g2_stride((int)n2, s->n4iwb);          // per chain and flavour: the n2 x n2 table, then its 2 N4iwb + 1 central diagonals as rows
        const size_t g2bytes = g2all * sizeof(double2);
        if (g2bytes > ((size_t)32 << 30)) return fail(s, W2DYN_EUNSUPPORTED, "two-particle measurement: %zu bytes of per-chain g2 scratch; use fewer chains or frequencies", g2bytes);
        const size_t per = (n4 * (1 + (s->meas_g4iwpp ? 1 : 0)) + (s->meas_g2iw ? ng2 : 0)) * sizeof(double2);
        s->nb4 = (int)std::max<size_t>(1, std::min<size_t>(B, acc_budget() / std::max<size_t>(per, 1)));
        if ((rc = dev_alloc(s, s->G2, g2all))) return rc;
        if ((rc = dev_alloc(s, s->acc4, (size_t)s->nb4 * n4))) return rc;
        if (s->meas_g4iwpp) { if ((rc = dev_alloc(s, s->acc4pp, (size_t)s->nb4 * n4))) return rc; }
        if (s->meas_g2iw) { if ((rc = dev_alloc(s, s->acc2, (size_t)s->nb4 * ng2))) return rc; }
        if ((size_t)3 * kmax * G2_FC * sizeof(double2) > 160 * 1024) return fail(s, W2DYN_EUNSUPPORTED, "kmax=%d too large for the two-particle transform", kmax);
    }
    // initial chain state: empty configuration, outer superstate drawn with the chain's first random number
    // (streams go on from where the previous solve of this context left them; a context whose chain count changed starts
    // every chain behind the furthest position reached so far, so that no chain ever re-uses draws)
    unsigned long long pos_max = 0;
    for (unsigned long long v : s->stream_pos) pos_max = std::max(pos_max, v);
    std::vector<WalkerScalars> ws(W);
    for (size_t w = 0; w < W; ++w) {
        WalkerScalars &x = ws[w];
        memset(&x, 0, sizeof x);
        w2dyn_rng_state r{(uint64_t)s->seed, (uint64_t)s->first_walker + w, s->stream_pos.size() == W ? s->stream_pos[w] : pos_max};
        double u = w2dyn_uniform(&r);
        int sel = nouter - 1;
        for (int st = 0; st < nouter; ++st) if (s->outer_cum[st] >= u && s->outer_w[st] != 0.0) { sel = st; break; }
        x.ndraw = r.ndraw; x.outer = sel; x.npairs = 0; x.wpos = 0; x.det_sign = 1.0; x.det_log = 0.0;
    }
    HIPCHK(hipMemcpyAsync(C.ws, ws.data(), W * sizeof(WalkerScalars), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->have_solver = true; s->aborted = 0; s->meas_done = 0; s->steps_done = 0;
    s->res.clear(); s->counters.assign(C_NCOUNT, 0.0);
    acquire_traps(s);                                                                 // CTQMC.F90:8076-8078
    return 0;
}

int w2dyn_set_taudiffmax(w2dyn_solver *s, double tdm)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "set_taudiffmax before init_solver");
    HIPCHK(hipSetDevice(s->device));
    host_init_window(s, tdm, s->C.num_windows, s->C.taudiff_max);     // note: NSlide is NOT recomputed (CTQMC.F90:8069 vs 8180)
    std::vector<WalkerScalars> ws(s->nwalkers);
    HIPCHK(hipMemcpy(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    for (auto &w : ws) { w.wpos = 0; w.npairs = -1; }
    HIPCHK(hipMemcpy(s->C.ws, ws.data(), ws.size() * sizeof(WalkerScalars), hipMemcpyHostToDevice));
    return 0;
}

int w2dyn_init_counters(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "init_counters before init_solver");
    HIPCHK(hipSetDevice(s->device));
    std::vector<WalkerScalars> ws(s->nwalkers);
    HIPCHK(hipMemcpy(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    for (auto &w : ws) memset(w.cnt, 0, sizeof w.cnt);
    HIPCHK(hipMemcpy(s->C.ws, ws.data(), ws.size() * sizeof(WalkerScalars), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->C.accpair, 0, sizeof(unsigned long long) * s->nflav * s->C.naccpair));
    if (s->C.apt_cnt) HIPCHK(hipMemset(s->C.apt_cnt, 0, sizeof(int) * s->nwalkers));           // apt_index = 1, CTQMC.F90:888
    if (s->C.calib_pairs) HIPCHK(hipMemset(s->C.calib_pairs, 0, sizeof(int) * s->nwalkers * s->nflav));
    return 0;
}

// AccPair back to Ntau bins over beta (ctqmc_warmup / ctqmc_measure, CTQMC.F90:8199-8205, 8387-8393)
static void accpair_to_default(w2dyn_solver *s)
{
    s->C.accpair = s->accpair_default; s->C.naccpair = s->ntau; s->C.accpair_max = s->beta; s->C.calib = 0;
}

static int interrupted_call(w2dyn_solver *s, const char *what)
{
    return fail(s, W2DYN_EINTERRUPTED, "%s: a signal (SIGINT / SIGTERM / SIGUSR1) arrived while this context's solver exists; nothing was launched", what);
}

static int sweep_impl(w2dyn_solver *s, int64_t nsteps, int count_as_sampling);
int w2dyn_sweep(w2dyn_solver *s, int64_t nsteps, int count_as_sampling)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "sweep before init_solver");
    if (g_signal_fired) return interrupted_call(s, "w2dyn_sweep");
    return sweep_impl(s, nsteps, count_as_sampling);
}
static int sweep_impl(w2dyn_solver *s, int64_t nsteps, int count_as_sampling)
{
    if (nsteps <= 0) return 0;
    HIPCHK(hipSetDevice(s->device));
    LaunchCfg l = walker_launch(s);
    int rc;
    if ((rc = time_begin(s))) return rc;
    // instantiations: 0 = one-state blocks (closed-form trace); 2, 3, 4 = matrix path with blocks padded to that size
    // (tree product); 9 = matrix path, sequential propagation (blocks of 5..8 states)
    if (s->cur_isector != 1) launch_sweep_worm(s->cur_isector, mat_variant(s), l, s->C, (long long)nsteps, 0);   // counts its own steps
    else launch_sweep(mat_variant(s), false, l, s->C, (long long)nsteps, count_as_sampling, 0);
    s->last_count_sampling = count_as_sampling;
    HIPCHK(hipGetLastError());
    if ((rc = time_end(s, s->cur_isector != 1 ? "sweep_worm" : "sweep"))) return rc;
    s->steps_done += nsteps;
    return 0;
}

int w2dyn_sync(w2dyn_solver *s)
{
    if (!s) return W2DYN_EINVAL;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return drain_timers(s);
}

int w2dyn_ctqmc_calibrate(w2dyn_solver *s, int list, int phase1pairnum, int phase2taugrid, int phase2pairnum, int nprogress)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "ctqmc_calibrate before init_solver");
    if (phase1pairnum < 1 || phase2taugrid < 1 || phase2pairnum < 0 || nprogress < 1) return fail(s, W2DYN_EINVAL, "ctqmc_calibrate: bad arguments");
    if (s->offdiag) return fail(s, W2DYN_EUNSUPPORTED, "ctqmc_calibrate with offdiag = 1 is not supported: pass the window with set_taudiffmax");
    if (s->C.xmoves) return fail(s, W2DYN_EUNSUPPORTED, "ctqmc_calibrate with Percentage4OperatorMove or PercentageOuterMove > 0 is not supported: calibrate the window with both at 0 and hand it over with set_taudiffmax");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    DevConst &C = s->C; const size_t W = s->nwalkers, nf = s->nflav;
    int rc;
    if (list) {      // CTQMC.F90:8100-8101: the list is allocated once; the histogram keeps its current shape
        if (!C.apt_tau || C.napt != phase1pairnum) {
            if ((rc = dev_alloc(s, C.apt_tau, W * (size_t)phase1pairnum))) return rc;
            if (!C.apt_cnt) { if ((rc = dev_alloc(s, C.apt_cnt, W))) return rc; }
            else HIPCHK(hipMemsetAsync(C.apt_cnt, 0, sizeof(int) * W, s->stream));
            C.napt = phase1pairnum;
        }
        C.calib = 1;
    } else {         // CTQMC.F90:8103-8108
        unsigned long long *h = nullptr;
        if ((rc = dev_alloc(s, h, nf * (size_t)phase2taugrid))) return rc;
        C.accpair = h; C.naccpair = phase2taugrid; C.accpair_max = C.taudiff_max;
        if (!C.calib_pairs) { if ((rc = dev_alloc(s, C.calib_pairs, W * nf))) return rc; }
        else HIPCHK(hipMemsetAsync(C.calib_pairs, 0, sizeof(int) * W * nf, s->stream));
        C.phase2pairnum = phase2pairnum; C.calib = 2;
    }
    const long long chunk = std::min(1000, nprogress);          // the completion test runs every min(1000, nprogress) steps
    LaunchCfg l = walker_launch(s);
    std::vector<int> cnt(list ? W : W * nf);
    bool done = false;
    for (long long it = 0; it < 1000000 && !done && !g_signal_fired; ++it) {      // CTQMC.F90:8123
        if ((rc = time_begin(s))) return rc;
        launch_sweep(mat_variant(s), true, l, s->C, chunk, 0, 0);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "sweep_calibrate"))) return rc;
        if ((rc = finish_paused(s, 1))) return rc;
        HIPCHK(hipMemcpyAsync(cnt.data(), list ? C.apt_cnt : C.calib_pairs, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        done = true;
        for (int v : cnt) if (v < (list ? phase1pairnum : phase2pairnum)) { done = false; break; }
        if ((it & 15) == 15 || done) { if ((rc = check_walker_errors(s))) return rc; }
    }
    if (!done && !g_signal_fired) return fail(s, W2DYN_ENUMERIC, "ctqmc_calibrate: not every chain completed phase %d within 10^9 steps", list ? 1 : 2);
    if (list) {
        // every chain owns a list like every MPI rank of the reference does; `accpairtau` is the mean over chains of the
        // sorted lists, so that sorted[int(0.995 n)] is the chain average of the percentile (impurity.py:1159-1162)
        s->accpairtau_chains.resize(W * (size_t)phase1pairnum);
        HIPCHK(hipMemcpy(s->accpairtau_chains.data(), C.apt_tau, s->accpairtau_chains.size() * sizeof(double), hipMemcpyDeviceToHost));
        s->accpairtau.assign(phase1pairnum, 0.0);
        std::vector<double> tmp(phase1pairnum);
        for (size_t w = 0; w < W; ++w) {
            std::copy(s->accpairtau_chains.begin() + w * phase1pairnum, s->accpairtau_chains.begin() + (w + 1) * phase1pairnum, tmp.begin());
            std::sort(tmp.begin(), tmp.end());
            for (int i = 0; i < phase1pairnum; ++i) s->accpairtau[i] += tmp[i] / (double)W;
        }
    }
    s->res_accpair.resize(nf * (size_t)C.naccpair);
    HIPCHK(hipMemcpy(s->res_accpair.data(), C.accpair, s->res_accpair.size() * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

int w2dyn_ctqmc_warmup(w2dyn_solver *s, int64_t n)
{
    int rc;
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "ctqmc_warmup before init_solver");
    accpair_to_default(s);
    if ((rc = w2dyn_init_counters(s))) return rc;
    double t0 = now_s();
    // launches of at most WARMUP_CHUNK steps: a chain's trajectory does not depend on how its steps are cut into launches,
    // and a signal (CTQMC.F90:8214) is noticed between two of them
    const int64_t WARMUP_CHUNK = 200000;
    for (int64_t done = 0; done < n && !g_signal_fired; done += WARMUP_CHUNK) {
        if ((rc = sweep_impl(s, std::min<int64_t>(WARMUP_CHUNK, n - done), 0))) return rc;
        if ((rc = finish_paused(s, 0))) return rc;
    }
    if ((rc = grow_if_close(s))) return rc;
    if ((rc = check_walker_errors(s))) return rc;
    s->time_warmup = now_s() - t0;
    return 0;
}

int w2dyn_measure_begin(w2dyn_solver *s)
{
    int rc;
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "measure before init_solver");
    accpair_to_default(s);
    if ((rc = w2dyn_init_counters(s))) return rc;
    const DevConst &C = s->C; const size_t B = C.nbins; const int nf = s->nflav;
    if (s->acc_gleg_full) {
        HIPCHK(hipMemsetAsync(s->acc_gleg_full, 0, sizeof(double) * B * nf * nf * s->nleg, s->stream));
        HIPCHK(hipMemsetAsync(s->acc_gtau_full, 0, sizeof(double) * B * nf * nf * s->ntau, s->stream));
    }
    HIPCHK(hipMemsetAsync(C.acc_gleg, 0, sizeof(double) * B * nf * s->nleg, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_gtau, 0, sizeof(double) * B * nf * s->ntau, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_giw, 0, sizeof(double) * B * nf * C.niw2 * 2, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_histo, 0, sizeof(double) * B * nf * (s->maxhisto + 1), s->stream));
    HIPCHK(hipMemsetAsync(C.acc_dm, 0, sizeof(double) * B * C.dmsize, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_outer, 0, sizeof(double) * B * C.nouter, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_signouter, 0, sizeof(double) * B * C.nouter, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_sign, 0, sizeof(double) * B, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_signstate, 0, sizeof(double) * B * s->nstates, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_tcstate, 0, sizeof(double) * B * s->nstates, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_nz, 0, sizeof(double) * B, s->stream));
    HIPCHK(hipMemsetAsync(C.acc_nw, 0, sizeof(double) * B, s->stream));
    {
        const size_t one = (size_t)s->nbw * 4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1) * 2;
        if (C.acc_g4w) HIPCHK(hipMemsetAsync(C.acc_g4w, 0, sizeof(double) * one, s->stream));
        if (C.acc_g4wpp) HIPCHK(hipMemsetAsync(C.acc_g4wpp, 0, sizeof(double) * one, s->stream));
    }
    if (s->acc_g4tau) HIPCHK(hipMemsetAsync(s->acc_g4tau, 0, sizeof(double) * s->nbl * nf * nf * s->n4tau * s->n4tau * s->n4tau, s->stream));
    if (s->acc_g4leg) HIPCHK(hipMemsetAsync(s->acc_g4leg, 0, sizeof(double) * s->nbl * nf * nf * s->n4leg * s->n4leg * (2 * s->n4iwb + 1) * 2, s->stream));
    if (C.acc_giww) HIPCHK(hipMemsetAsync(C.acc_giww, 0, sizeof(double) * B * C.niw2 * 2, s->stream));
    if (C.acc_gtauw) HIPCHK(hipMemsetAsync(C.acc_gtauw, 0, sizeof(double) * B * s->ntau, s->stream));
    if (s->acc_wie_iw) HIPCHK(hipMemsetAsync(s->acc_wie_iw, 0, sizeof(double) * (size_t)s->nbie * s->wie_n, s->stream));
    if (s->acc_wie_tau) HIPCHK(hipMemsetAsync(s->acc_wie_tau, 0, sizeof(double) * (size_t)s->nbie * s->ntau, s->stream));
    if (s->acc_weq_iw) HIPCHK(hipMemsetAsync(s->acc_weq_iw, 0, sizeof(double) * (size_t)s->nbe * 2 * (2 * s->n2iwb + 1), s->stream));
    if (s->acc_weq_tau) HIPCHK(hipMemsetAsync(s->acc_weq_tau, 0, sizeof(double) * (size_t)s->nbe * s->ntau, s->stream));
    if (s->acc_weq_p3) HIPCHK(hipMemsetAsync(s->acc_weq_p3, 0, sizeof(double) * (size_t)s->nbe * 4 * s->n3iwf * (2 * s->n3iwb + 1), s->stream));
    if (s->acc_susz) HIPCHK(hipMemsetAsync(s->acc_susz, 0, sizeof(double) * (size_t)s->nbs * nf * nf * s->ntau, s->stream));
    if (s->ser) HIPCHK(hipMemsetAsync(s->ser, 0, sizeof(double) * (size_t)(2 * nf + 1) * (size_t)s->nmeas, s->stream));
    if (s->fourpnt & 4) {
        const size_t n2 = 2 * (size_t)(s->n4iwf + s->n4iwb), n4 = (size_t)nf * nf * 4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1), ng2 = (size_t)nf * n2 * n2;
        HIPCHK(hipMemsetAsync(s->acc4, 0, sizeof(double2) * s->nb4 * n4, s->stream));
        if (s->acc4pp) HIPCHK(hipMemsetAsync(s->acc4pp, 0, sizeof(double2) * s->nb4 * n4, s->stream));
        if (s->acc2) HIPCHK(hipMemsetAsync(s->acc2, 0, sizeof(double2) * s->nb4 * ng2, s->stream));
    }
    s->meas_done = 0; s->time_sim_steps = 0;
    return 0;
}

// one pass of the estimators; g2buf: per-chain g2 table of the Matsubara two-particle estimators (FourPnt & 4)
static int measure_once_impl(w2dyn_solver *s, double2 *g2buf)
{
    HIPCHK(hipSetDevice(s->device));
    LaunchCfg l = walker_launch(s);
    int rc;
    if ((rc = time_begin(s))) return rc;
    launch_meas_misc(mat_variant(s), l, s->C);
    HIPCHK(hipGetLastError());
    if ((rc = time_end(s, "meas_misc"))) return rc;
    if (s->cur_isector != 1) {      // component sampling: no Z-space Green's function estimators (CTQMC.F90:8554), worm estimator instead
        if (s->cur_isector == 2 || s->cur_isector == 3) {
            if ((s->cur_isector == 2 && (s->C.acc_giww || s->C.acc_gtauw)) || (s->cur_isector == 3 && s->worm_meas_gsigma)) {
                if ((rc = time_begin(s))) return rc;
                launch_meas_gw(s->C, s->stream);
                HIPCHK(hipGetLastError());
                if ((rc = time_end(s, "meas_gw"))) return rc;
            }
        } else if (is_wie_sector(s->cur_isector)) {
            const WieSector e = wie_sector(s, s->cur_isector);
            double *iw = s->worm_meas_more[e.q_iw] ? s->acc_wie_iw : nullptr;
            double *tau = (e.q_tau >= 0 && s->worm_meas_more[e.q_tau] && (iw || !e.needs_iw)) ? s->acc_wie_tau : nullptr;
            if (iw || tau) {
                if ((rc = time_begin(s))) return rc;
                launch_meas_wie(s->C, s->stream, s->nbie, iw, tau, e.n1, e.n2, e.n3);
                HIPCHK(hipGetLastError());
                if ((rc = time_end(s, "meas_wie"))) return rc;
            }
        } else if (s->cur_isector >= 6) {
            // sector 6 measures p2iw / p2tau (ph), 7 the pp ones, 8 p3iw, 9 p3iwpp - each only when asked for (CTQMC.F90:8603-8616)
            const int is = s->cur_isector;
            double *iw = (is == 6 && s->worm_meas_eq[0]) || (is == 7 && s->worm_meas_eq[1]) ? s->acc_weq_iw : nullptr;
            double *tau = (is == 6 && s->worm_meas_eq[2]) || (is == 7 && s->worm_meas_eq[3]) ? s->acc_weq_tau : nullptr;
            double *p3 = (is == 8 && s->worm_meas_eq[4]) || (is == 9 && s->worm_meas_eq[5]) ? s->acc_weq_p3 : nullptr;
            if (iw || tau || p3) {
                if ((rc = time_begin(s))) return rc;
                launch_meas_weq(s->C, s->stream, s->nbe, iw, tau, p3, s->n2iwb, s->n3iwf, s->n3iwb);
                HIPCHK(hipGetLastError());
                if ((rc = time_end(s, "meas_weq"))) return rc;
            }
        } else if (s->C.acc_g4w || s->C.acc_g4wpp) {
            if ((rc = time_begin(s))) return rc;
            if (s->C.acc_g4w) launch_meas_g4w(s->C, s->stream, 0, s->nbw);
            if (s->C.acc_g4wpp) launch_meas_g4w(s->C, s->stream, 1, s->nbw);
            HIPCHK(hipGetLastError());
            if ((rc = time_end(s, "meas_g4w"))) return rc;
        }
        s->meas_done += 1;
        return 0;
    }
    if (s->offdiag) {                  // CTQMC.F90:8562-8575: MeasGtauRem_full and nothing else of the Green's function estimators
        if ((rc = time_begin(s))) return rc;
        launch_meas_full(s->C, s->stream, s->acc_gleg_full, s->acc_gtau_full);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_full"))) return rc;
        s->meas_done += 1;
        return 0;
    }
    if (s->accum_gtau != 0) {          // AccumGtau = 0: MeasGtauRem returns at once, neither G(l) nor G(tau) (CTQMC.F90:2267)
    if ((rc = time_begin(s))) return rc;
    launch_meas_gleg(s->C, s->stream, s->gleg_segpre, s->ncu);
    HIPCHK(hipGetLastError());
    if ((rc = time_end(s, "meas_gleg"))) return rc;
    }
    if (s->ser && s->accum_gtau != 0 && s->meas_done < s->nmeas) {     // entry g_inmeas of the series (CTQMC.F90:2394-2402, 8424)
        launch_meas_steps(s->C, s->stream, s->ser, (int)s->nmeas, (int)s->meas_done, s->step_series);
        HIPCHK(hipGetLastError());
    }
    if (s->accum_gtau == 2) {
        if ((rc = time_begin(s))) return rc;
        launch_meas_gtau_ac(s->C, s->stream, s->ac_dat, s->ac_lev, s->ac_n);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_gtau_ac"))) return rc;
        s->ac_n += 1;
    }
    if (s->acc_susz) {                // CTQMC.F90:8556-8560
        if ((rc = time_begin(s))) return rc;
        if (s->mat) launch_meas_susz_mat(l, s->C, s->acc_susz, s->nbs, s->susz_work, s->susz_dens);
        else launch_meas_susz(s->C, s->stream, s->acc_susz, s->nbs);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_susz"))) return rc;
    }
    if (s->meas_giw) {
        if ((rc = time_begin(s))) return rc;
        launch_meas_giw(s->C, s->stream);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_giw"))) return rc;
    }
    if (s->fourpnt & 3) {           // CTQMC.F90:8578
        if ((rc = time_begin(s))) return rc;
        // g4leg: factorised kernel (registers hold the N4iwb + 1 non-negative frequencies of one (l, l')) when its tables fit
        // into LDS and N4iwb < 24; otherwise, and for g4tau, the quadruple kernel with one atomic add per term
        const size_t lds_f = g4leg_fact_lds(s->C, s->n4leg);
        const bool fact = s->acc_g4leg && s->n4iwb < 24 && lds_f <= 64 * 1024 && !getenv("W2DYN_G4LEG_ATOMICS");
        if (s->acc_g4tau || !fact) launch_meas_g4leg(s->C, s->stream, s->acc_g4tau, fact ? nullptr : s->acc_g4leg, s->nbl, s->n4tau, s->n4leg, s->n4iwb);
        if (fact) launch_meas_g4leg_fact(s->C, s->stream, s->acc_g4leg, s->nbl, s->n4leg, s->n4iwb);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_g4leg"))) return rc;
    }
    if (s->fourpnt & 4) {           // CTQMC.F90:8579
        const int n2 = 2 * (s->n4iwf + s->n4iwb);
        if ((rc = time_begin(s))) return rc;
        launch_meas_g2iw(s->C, s->stream, n2, s->n4iwb, g2buf);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_g2iw"))) return rc;
        if ((rc = time_begin(s))) return rc;
        launch_meas_g4iw(s->C, s->stream, s->n4iwf, s->n4iwb, s->zph, s->nb4, g2buf, s->acc4, s->acc4pp, s->acc2);
        HIPCHK(hipGetLastError());
        if ((rc = time_end(s, "meas_g4iw"))) return rc;
    }
    s->meas_done += 1;
    return 0;
}

int w2dyn_measure_once(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "measure before init_solver");
    if (g_signal_fired) return interrupted_call(s, "w2dyn_measure_once");
    return measure_once_impl(s, s->G2);
}

int w2dyn_measure_once_noalloc(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "measure before init_solver");
    if (!(s->fourpnt & 4) || s->cur_isector != 1) return fail(s, W2DYN_EINVAL, "measure_once_noalloc: a Z-space run with FourPnt & 4 (Matsubara two-particle estimators) is required");
    const size_t n2 = 2 * (size_t)(s->n4iwf + s->n4iwb), need = (size_t)s->nwalkers * s->nflav * g2_stride((int)n2, s->n4iwb) * sizeof(double2);
    if (need > W2DYN_SCRATCH_BYTES)
        return fail(s, W2DYN_EINVAL, "measure_once_noalloc: %zu bytes of per-chain g2 tables (chains x flavours x n2 (n2 + 2 N4iwb + 1) x 16, n2 = 2 (N4iwf + N4iwb)) exceed the static scratch of %zu bytes", need, (size_t)W2DYN_SCRATCH_BYTES);
    if (!s->scratch) return fail(s, W2DYN_EHIP, "measure_once_noalloc: no scratch address for device %d", s->device);
    if (g_signal_fired) return interrupted_call(s, "w2dyn_measure_once_noalloc");
    // nothing below allocates, frees or waits for the GPU: kernel launches on the context's stream only (timing is
    // skipped).  k_meas_g2iw writes every entry of every chain's table, so the array's old contents never matter.
    const bool timing = s->timing;
    s->timing = false;
    const int rc = measure_once_impl(s, (double2 *)s->scratch);
    s->timing = timing;
    return rc;
}

int w2dyn_measure_finish(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "measure_finish before init_solver");
    HIPCHK(hipSetDevice(s->device));
    int rc;
    if ((rc = check_walker_errors(s))) return rc;
    const DevConst &C = s->C;
    const int nf = s->nflav, nd = s->nd, B = C.nbins, W = s->nwalkers, ns = s->nstates;
    hipError_t fetch_err = hipSuccess;
    auto fetch = [&](const double *dev, size_t n) {
        std::vector<double> v(n);
        hipError_t e = hipMemcpy(v.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess && fetch_err == hipSuccess) fetch_err = e;
        return v;
    };
    launch_gleg_legendre(s->stream, C.acc_gleg, s->gleg_tab, s->nleg, (size_t)B * nf, s->gleg_leg);
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<double> gleg = fetch(s->gleg_leg, (size_t)B * nf * s->nleg), gtau = fetch(C.acc_gtau, (size_t)B * nf * s->ntau);
    std::vector<double> giw = fetch(C.acc_giw, (size_t)B * nf * C.niw2 * 2), histo = fetch(C.acc_histo, (size_t)B * nf * (s->maxhisto + 1));
    const int no = C.nouter, dms = C.dmsize;
    const LocalProblem &lp = s->lp;
    std::vector<double> dm = fetch(C.acc_dm, (size_t)B * dms), outer = fetch(C.acc_outer, (size_t)B * no), souter = fetch(C.acc_signouter, (size_t)B * no);
    std::vector<double> sgn = fetch(C.acc_sign, B);
    std::vector<double> sgnstate = fetch(C.acc_signstate, (size_t)B * ns), tcstate = fetch(C.acc_tcstate, (size_t)B * ns);
    if (fetch_err != hipSuccess) return fail(s, W2DYN_EHIP, "measure_finish: copying the accumulators to the host: %s", hipGetErrorString(fetch_err));
    std::vector<WalkerScalars> ws(W);
    HIPCHK(hipMemcpy(ws.data(), C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    s->res_accpair.resize((size_t)nf * s->ntau);
    HIPCHK(hipMemcpy(s->res_accpair.data(), C.accpair, s->res_accpair.size() * sizeof(long long), hipMemcpyDeviceToHost));
    s->counters.assign(C_NCOUNT, 0.0);
    for (auto &w : ws) for (int i = 0; i < C_NCOUNT; ++i) s->counters[i] += (double)w.cnt[i];
    if (s->meas_done == 0 && !s->no_sampling) { s->aborted = 2; return 0; }          // CTQMC.F90:8653-8657, 8665
    // measurements in Z space per bin: every chain at every measurement in a Z run, the chains found in Z space in a
    // worm run (CntMeas(SectorZ), CTQMC.F90:8541-8545)
    std::vector<double> nz = fetch(C.acc_nz, B), nwm = fetch(C.acc_nw, B);
    if (fetch_err != hipSuccess) return fail(s, W2DYN_EHIP, "measure_finish: copying the accumulators to the host: %s", hipGetErrorString(fetch_err));
    double nmeas_tot = 0.0; for (double v : nz) nmeas_tot += v;
    double ssum = 0.0; for (double v : sgn) ssum += v;
    s->mean_sign = ssum / nmeas_tot;                               // CTQMC.F90:8670
    const double ms = s->mean_sign, beta = s->beta;
    auto binw = [&](int b) { return nz[b]; };
    auto &R = s->res;
    R.clear();
    // per-bin normalised arrays and their walker-weighted mean
    s->pack.clear();
    auto reduce_bins = [&](const std::vector<double> &acc, size_t per, const char *name, auto norm, const double *dev = nullptr, int nleg_fac = 0) {
        std::vector<double> bins((size_t)B * per), tot(per, 0.0);
        if (dev) {     // every normalisation is linear in the accumulated value: its factor per bin feeds w2dyn_reduce_pack
            PackDesc d{dev, B, per, std::vector<double>(B), nleg_fac};
            for (int b = 0; b < B; ++b) d.scale[b] = norm(1.0, binw(b), 0);
            s->pack[name] = d;
        }
        for (int b = 0; b < B; ++b) for (size_t i = 0; i < per; ++i) { tot[i] += acc[(size_t)b * per + i]; bins[(size_t)b * per + i] = norm(acc[(size_t)b * per + i], binw(b), i); }
        for (size_t i = 0; i < per; ++i) tot[i] = norm(tot[i], nmeas_tot, i);
        R[std::string(name) + "_bins"] = bins; R[name] = tot;
    };
    const int nleg = s->nleg, ntau = s->ntau;
    // GLeg*sqrt(2l+1)/beta, then -GLeg*zScaling with zScaling = -1/(N*mean_sign)          (CTQMC.F90:8697, 8766-8771)
    reduce_bins(gleg, (size_t)nf * nleg, "gleg", [&](double v, double n, size_t i) { int l = (int)(i % nleg); double zs = -1.0 / (n * ms); return -(v * std::sqrt((double)(2 * l) + 1.0) / beta) * zs; }, s->gleg_leg, nleg);
    // Gtau = -mean/(-mean_sign) * Ntau/beta^2                                              (CTQMC.F90:8726-8731)
    reduce_bins(gtau, (size_t)nf * ntau, "gtau", [&](double v, double n, size_t) { return -(v / n) / (-ms) * (double)ntau / (beta * beta); }, C.acc_gtau);
    reduce_bins(giw, (size_t)nf * C.niw2 * 2, "giw", [&](double v, double n, size_t) { double zs = -1.0 / (n * ms); return v * zs / beta; }, C.acc_giw);   // 8815-8817
    reduce_bins(histo, (size_t)nf * (s->maxhisto + 1), "histo", [&](double v, double n, size_t) { return v / n; }, C.acc_histo);                   // 8755
    reduce_bins(dm, dms, "dmblocks", [&](double v, double n, size_t) { double zs = -1.0 / (n * ms); return v * zs; });                             // 8706-8708
    reduce_bins(outer, no, "outersuperstatehisto", [&](double v, double n, size_t) { return v / n; });                                             // 8760
    {   // MeasOuterHistograms (CTQMC.F90:3285-3339), normalised as in 8757-8764: sign sums over visit counts (0/0 = NaN
        // for blocks never visited, as in the reference), visits and trace contributions over the number of measurements
        auto outer_hists = [&](const double *ou, const double *so_, const double *ss_, const double *tc_, double n,
                               double *sign_sst, double *ostate, double *sign_state, double *tc_sst, double *tc_state) {
            for (int b = 0; b < no; ++b) {
                sign_sst[b] = so_[b] / ou[b]; tc_sst[b] = 0.0;
                for (int i = 0; i < lp.sst_n[b]; ++i) {       // every state of the outer block counts as visited (3298-3300)
                    const int st = lp.sst_off[b] + i;
                    ostate[st] = ou[b] / n; sign_state[st] = ss_[st] / ou[b]; tc_state[st] = tc_[st] / n; tc_sst[b] += tc_[st] / n;
                }
            }
        };
        std::vector<double> ou(no, 0.0), so(no, 0.0), ss(ns, 0.0), tc(ns, 0.0);
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < no; ++i) { ou[i] += outer[(size_t)b * no + i]; so[i] += souter[(size_t)b * no + i]; }
            for (int i = 0; i < ns; ++i) { ss[i] += sgnstate[(size_t)b * ns + i]; tc[i] += tcstate[(size_t)b * ns + i]; }
        }
        std::vector<double> a1(no), a2(ns), a3(ns), a4(no), a5(ns);
        outer_hists(ou.data(), so.data(), ss.data(), tc.data(), nmeas_tot, a1.data(), a2.data(), a3.data(), a4.data(), a5.data());
        R["signhistosuperstates"] = a1; R["outerstatehisto"] = a2; R["signhistostates"] = a3; R["tracecontribsuperstates"] = a4; R["tracecontribstates"] = a5;
        std::vector<double> b1((size_t)B * no), b2((size_t)B * ns), b3((size_t)B * ns), b4((size_t)B * no), b5((size_t)B * ns);
        for (int b = 0; b < B; ++b)
            outer_hists(&outer[(size_t)b * no], &souter[(size_t)b * no], &sgnstate[(size_t)b * ns], &tcstate[(size_t)b * ns], binw(b),
                        &b1[(size_t)b * no], &b2[(size_t)b * ns], &b3[(size_t)b * ns], &b4[(size_t)b * no], &b5[(size_t)b * ns]);
        R["signhistosuperstates_bins"] = b1; R["outerstatehisto_bins"] = b2; R["signhistostates_bins"] = b3;
        R["tracecontribsuperstates_bins"] = b4; R["tracecontribstates_bins"] = b5;
    }
    {   // diagonal of the density matrix in the occupation basis, block order (Transform_DensityMatrix_EB_to_OCCB, 3163-3179)
        auto to_occ_diag = [&](const double *blocks, double *out) {
            for (int b = 0; b < no; ++b) {
                const int n = lp.sst_n[b]; const double *Rb = blocks + s->dmoff[b]; const std::vector<double> &V = lp.evec[b];
                for (int i = 0; i < n; ++i) { double a = 0.0; for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) a += V[i + (size_t)n * k] * Rb[k * n + l] * V[i + (size_t)n * l]; out[lp.sst_off[b] + i] = a; }
            }
        };
        std::vector<double> d(ns), db((size_t)B * ns);
        to_occ_diag(R["dmblocks"].data(), d.data());
        for (int b = 0; b < B; ++b) to_occ_diag(&R["dmblocks_bins"][(size_t)b * dms], &db[(size_t)b * ns]);
        R["dmdiag"] = d; R["dmdiag_bins"] = db;
    }
    // density matrix and its contractions (CTQMC.F90:2842-3060 for one-state blocks; eigenbasis == occupation basis)
    auto occ_of = [&](const double *d, std::vector<double> &occ, std::vector<double> &rho1, std::vector<double> &rho2) {
        occ.assign((size_t)nf * nf, 0.0); rho1.assign((size_t)nf * nf, 0.0); rho2.assign((size_t)nf * nf * nf * nf, 0.0);
        for (int idx = 0; idx < ns; ++idx) {
            double wgt = d[idx]; if (wgt == 0.0) continue;
            const int st = lp.states[idx];
            for (int a = 0; a < nf; ++a) {
                if (!((st >> flav_bit_h(a, nd)) & 1)) continue;
                rho1[(size_t)a * nf + a] += wgt;
                for (int b = 0; b < nf; ++b) {
                    if (!((st >> flav_bit_h(b, nd)) & 1)) continue;
                    occ[(size_t)a * nf + b] += wgt;
                    if (a != b) {
                        rho2[(((size_t)a * nf + b) * nf + b) * nf + a] += wgt;      // <c+_a c+_b c_b c_a>
                        rho2[(((size_t)b * nf + a) * nf + b) * nf + a] -= wgt;      // <c+_b c+_a c_b c_a>
                    }
                }
            }
        }
    };
    if (!s->mat) {
        std::vector<double> occ, rho1, rho2;
        occ_of(R["dmdiag"].data(), occ, rho1, rho2);
        R["occ"] = occ; R["rho1"] = rho1; R["rho2"] = rho2;
        std::vector<double> ob((size_t)B * nf * nf), o1, r1, r2;
        for (int b = 0; b < B; ++b) { occ_of(&R["dmdiag_bins"][(size_t)b * ns], o1, r1, r2); std::copy(o1.begin(), o1.end(), ob.begin() + (size_t)b * nf * nf); }
        R["occ_bins"] = ob;
        std::vector<double> full((size_t)ns * ns, 0.0);
        for (int i = 0; i < ns; ++i) full[(size_t)i * ns + i] = R["dmdiag"][i];
        R["densitymatrix"] = full;
    } else {
        // contractions of the block density matrix with the operators in the eigenbasis (CTQMC.F90:2842-3060)
        auto occ_gen = [&](const double *blocks, std::vector<double> &occ) {
            occ.assign((size_t)nf * nf, 0.0);
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) { int fl[4] = {b, b, a, a}, ca[4] = {2, 1, 2, 1}; occ[(size_t)a * nf + b] = lp.expect(blocks, s->dmoff, 4, fl, ca); }
        };
        const double *blk = R["dmblocks"].data();
        std::vector<double> occ, rho1((size_t)nf * nf, 0.0), rho2((size_t)nf * nf * nf * nf, 0.0);
        occ_gen(blk, occ);
        for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) { int fl[2] = {a, b}, ca[2] = {2, 1}; rho1[(size_t)b * nf + a] = lp.expect(blk, s->dmoff, 2, fl, ca); }
        for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) for (int f3 = 0; f3 < nf; ++f3) for (int f4 = 0; f4 < nf; ++f4) {
            int fl[4] = {f1, f2, f3, f4}, ca[4] = {2, 2, 1, 1};
            rho2[(((size_t)f4 * nf + f3) * nf + f2) * nf + f1] = lp.expect(blk, s->dmoff, 4, fl, ca);
        }
        R["occ"] = occ; R["rho1"] = rho1; R["rho2"] = rho2;
        std::vector<double> ob((size_t)B * nf * nf), o1;
        for (int b = 0; b < B; ++b) { occ_gen(&R["dmblocks_bins"][(size_t)b * dms], o1); std::copy(o1.begin(), o1.end(), ob.begin() + (size_t)b * nf * nf); }
        R["occ_bins"] = ob;
        std::vector<double> full((size_t)ns * ns, 0.0);
        for (int b = 0; b < no; ++b) {
            const int n = lp.sst_n[b], off = lp.sst_off[b]; const double *Rb = blk + s->dmoff[b]; const std::vector<double> &V = lp.evec[b];
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
                double a = 0.0;
                for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) a += V[i + (size_t)n * k] * Rb[k * n + l] * V[j + (size_t)n * l];
                full[(size_t)(off + i) * ns + off + j] = a;
            }
        }
        R["densitymatrix"] = full;
    }
    if (s->meas_ssq) {   // MeasureS2_from_densitymatrix, create_ssquare_full (CTQMC.F90:2769-2840)
        std::vector<double> ssq(2 * nd + 1, 0.0), sop((size_t)ns * ns, 0.0);
        const double *blk = R["dmblocks"].data();
        for (int b = 0; b < no; ++b) {
            const int n = lp.sst_n[b], off = lp.sst_off[b]; const std::vector<double> &V = lp.evec[b], &S = lp.s2_occ[b];
            const int nel = __builtin_popcount((unsigned)lp.states[off]);
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) sop[(size_t)(off + i) * ns + off + j] = S[i + (size_t)n * j];
            std::vector<double> T((size_t)n * n, 0.0);              // T = S V, then S_eig = V^T T
            for (int i = 0; i < n; ++i) for (int l = 0; l < n; ++l) { double a = 0.0; for (int j = 0; j < n; ++j) a += S[i + (size_t)n * j] * V[j + (size_t)n * l]; T[i + (size_t)n * l] = a; }
            const double *Rb = blk + s->dmoff[b];
            for (int k = 0; k < n; ++k) for (int l = 0; l < n; ++l) {
                double se = 0.0; for (int i = 0; i < n; ++i) se += V[i + (size_t)n * k] * T[i + (size_t)n * l];
                ssq[nel] += Rb[k * n + l] * se;                       // tr(rho S^2): S^2 is symmetric
            }
        }
        R["ssquare"] = ssq; R["ssquare_operator"] = sop;
    }
    s->res_isector = s->cur_isector;
    if (s->ac_dat) {     // gtau_blocks (CTQMC.F90:8733-8739, unscaled): per accumulator bin and averaged over the bins' chains
        const size_t per = (size_t)nf * s->ac_lev * ntau;
        launch_gtau_blocks(s->C, s->stream, s->ac_dat, s->ac_lev, s->ac_n, s->ac_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s->stream));
        std::vector<double> bins((size_t)B * per), tot(per, 0.0);
        HIPCHK(hipMemcpy(bins.data(), s->ac_out, bins.size() * sizeof(double), hipMemcpyDeviceToHost));
        double wtot = 0.0;
        for (int b = 0; b < B; ++b) {
            const double nch = (double)(std::min<long long>((long long)(b + 1) * C.walkers_per_bin, W) - (long long)b * C.walkers_per_bin);
            wtot += nch;
            for (size_t i = 0; i < per; ++i) tot[i] += bins[(size_t)b * per + i] * nch;
        }
        for (size_t i = 0; i < per; ++i) tot[i] /= wtot;
        R["gtau_blocks"] = tot; R["gtau_blocks_bins"] = bins;
    }
    if (s->ser) {        // CTQMC.F90:8741-8742
        const size_t nm = (size_t)s->nmeas;
        std::vector<double> all((size_t)(2 * nf + 1) * nm);
        HIPCHK(hipMemcpy(all.data(), s->ser, all.size() * sizeof(double), hipMemcpyDeviceToHost));
        const double b2 = s->beta * s->beta;
        std::vector<double> gm(all.begin(), all.begin() + nf * nm), gi(all.begin() + nf * nm, all.begin() + 2 * nf * nm), ss(all.begin() + 2 * nf * nm, all.end());
        for (auto &v : gm) v /= b2;
        for (auto &v : gi) v /= b2;
        if (s->step_series & 1) R["gtau_mean_step"] = gm;
        if (s->step_series & 2) R["gtau_mid_step"] = gi;
        if (s->step_series & 4) R["sign_step"] = ss;
    }
    {   // acceptance ratios (CTQMC.F90:8680-8695); 16 sectors, Z = index 0
        std::vector<double> aa(16, 0.0), ar(16, 0.0);
        aa[0] = s->counters[C_ACCADD] / s->counters[C_TRYADD]; ar[0] = s->counters[C_ACCREM] / s->counters[C_TRYREM];
        std::vector<double> wa(15, 0.0), wr(15, 0.0), wp(15, 0.0);      // AccWormAdd/Rem/Rep(2:NWormSectors), CTQMC.F90:8685-8691
        if (s->cur_isector != 1) {
            const int is = s->cur_isector;       // AccAdd(Sector) is indexed from 1, the worm arrays from sector 2
            aa[is - 1] = s->counters[C_W_ACCADD] / s->counters[C_W_TRYADD]; ar[is - 1] = s->counters[C_W_ACCREM] / s->counters[C_W_TRYREM];
            wa[is - 2] = s->counters[C_ACCWADD] / s->counters[C_TRYWADD]; wr[is - 2] = s->counters[C_ACCWREM] / s->counters[C_TRYWREM];
            wp[is - 2] = s->counters[C_ACCWREP] / s->counters[C_TRYWREP];
        }
        R["accwormadd"] = wa; R["accwormrem"] = wr; R["accwormrep"] = wp;
        R["accadd"] = aa; R["accrem"] = ar;
        R["accglob"] = {s->counters[C_ACCGLOB] / s->counters[C_TRYGLOB]};
        R["accadd4"] = {s->counters[C_ACCADD4] / s->counters[C_TRYADD4]};      // CTQMC.F90:8677-8678 (0/0 when the moves are off)
        R["accrem4"] = {s->counters[C_ACCREM4] / s->counters[C_TRYREM4]};
        R["accflavc"] = {s->counters[C_ACCFLAVC] / s->counters[C_TRYFLAVC]};   // CTQMC.F90:8681
        R["accshift"] = {s->counters[C_ACCSHIFT] / s->counters[C_TRYSHIFT]};
    }
    if (s->acc_susz && true) {      // CTQMC.F90:8744-8746: ntau_n0 * zScaling, point Ntau = point 0
        const size_t per = (size_t)nf * nf * ntau, pero = (size_t)nf * nf * (ntau + 1);
        std::vector<double> raw = fetch(s->acc_susz, (size_t)s->nbs * per), tot(pero, 0.0), bins((size_t)s->nbs * pero, 0.0);
        if (fetch_err != hipSuccess) return fail(s, W2DYN_EHIP, "measure_finish: copying the accumulators to the host: %s", hipGetErrorString(fetch_err));
        for (int b = 0; b < s->nbs; ++b) {
            double nb = 0.0;
            for (int bin = b; bin < B; bin += s->nbs) nb += binw(bin);
            const double zb = -1.0 / (nb * ms);
            for (int ff = 0; ff < nf * nf; ++ff) {
                for (int t = 0; t < ntau; ++t) {
                    const double v = raw[(size_t)b * per + (size_t)ff * ntau + t];
                    tot[(size_t)ff * (ntau + 1) + t] += v; bins[(size_t)b * pero + (size_t)ff * (ntau + 1) + t] = v * zb;
                }
                bins[(size_t)b * pero + (size_t)ff * (ntau + 1) + ntau] = bins[(size_t)b * pero + (size_t)ff * (ntau + 1)];
            }
        }
        const double zs = -1.0 / (nmeas_tot * ms);
        for (int ff = 0; ff < nf * nf; ++ff) {
            for (int t = 0; t < ntau; ++t) tot[(size_t)ff * (ntau + 1) + t] *= zs;
            tot[(size_t)ff * (ntau + 1) + ntau] = tot[(size_t)ff * (ntau + 1)];
        }
        R["ntau_n0"] = tot; R["ntau_n0_bins"] = bins;
    }
    if (s->fourpnt & 3) {     // CTQMC.F90:8772-8786: zScaling (N4tau/beta)^3 / beta; zScaling sqrt(2l+1) sqrt(2l'+1) (-1)^(l'+1) / beta
        const int nt = s->n4tau, nl = s->n4leg, nwm = 2 * s->n4iwb + 1;
        const double zs = -1.0 / (nmeas_tot * ms);
        if (s->acc_g4tau) {
            const size_t per = (size_t)nf * nf * nt * nt * nt;
            std::vector<double> raw((size_t)s->nbl * per), tot(per, 0.0);
            (void)hipMemcpy(raw.data(), s->acc_g4tau, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            const double fac = zs * ((double)nt / beta) * ((double)nt / beta) * ((double)nt / beta) / beta;
            for (int b = 0; b < s->nbl; ++b) for (size_t i = 0; i < per; ++i) tot[i] += raw[(size_t)b * per + i];
            for (size_t i = 0; i < per; ++i) tot[i] *= fac;
            R["g4tau"] = tot;
        }
        if (s->acc_g4leg) {
            const size_t per = (size_t)nf * nf * nl * nl * nwm * 2;
            std::vector<double> raw((size_t)s->nbl * per), tot((size_t)nf * nf * nt * nt * nwm * 2, 0.0);     // result shape uses N4tau (CTQMC.F90:549)
            (void)hipMemcpy(raw.data(), s->acc_g4leg, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int ff = 0; ff < nf * nf; ++ff) for (int l = 0; l < nl; ++l) for (int lp = 0; lp < nl; ++lp) {
                const double fac = zs * std::sqrt(2.0 * l + 1.0) * std::sqrt(2.0 * lp + 1.0) * (((lp + 1) & 1) ? -1.0 : 1.0) / beta;
                for (int m = 0; m < 2 * nwm; ++m) {
                    double a = 0.0;
                    for (int b = 0; b < s->nbl; ++b) a += raw[(size_t)b * per + (((size_t)ff * nl + l) * nl + lp) * nwm * 2 + m];
                    tot[(((size_t)ff * nt + l) * nt + lp) * nwm * 2 + m] = a * fac;
                }
            }
            R["g4leg"] = tot;
        }
    }
    if (s->fourpnt & 4) {     // CTQMC.F90:8790-8800: G4iw * zScaling / beta^2, G2iw * zScaling / beta, zScaling = -1/(N mean_sign)
        const size_t n2 = 2 * (size_t)(s->n4iwf + s->n4iwb), n4 = (size_t)nf * nf * 4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1), ng2 = (size_t)nf * n2 * n2;
        auto reduce4 = [&](const double2 *dev, size_t per, const char *name, double extra, bool wb_first) {
            std::vector<double> raw((size_t)s->nb4 * per * 2), bins((size_t)s->nb4 * per * 2), tot(per * 2, 0.0);
            (void)hipMemcpy(raw.data(), dev, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            PackDesc d{(const double *)dev, s->nb4, per * 2, std::vector<double>(s->nb4), 0};
            for (int b = 0; b < s->nb4; ++b) {
                double nb = 0.0;
                for (int bin = b; bin < B; bin += s->nb4) nb += binw(bin);
                d.scale[b] = (-1.0 / (nb * ms)) * extra;
                for (size_t i = 0; i < per * 2; ++i) { double v = raw[(size_t)b * per * 2 + i]; tot[i] += v; bins[(size_t)b * per * 2 + i] = v * (-1.0 / (nb * ms)) * extra; }
            }
            for (size_t i = 0; i < per * 2; ++i) tot[i] *= (-1.0 / (nmeas_tot * ms)) * extra;
            if (wb_first) {      // device order [f1][f2][wb][w1][w2] -> the reference's G4iw(.., w1, w2, wb) (the packed moments stay in device order)
                const size_t nw = 2 * (size_t)s->n4iwf, nbq = 2 * (size_t)s->n4iwb + 1, ff = (size_t)nf * nf;
                auto turn = [&](std::vector<double> &v, size_t lead) {
                    std::vector<double> o(v.size());
                    for (size_t a = 0; a < lead * ff; ++a) for (size_t q = 0; q < nbq; ++q) for (size_t i = 0; i < nw * nw; ++i) {
                        const size_t src = ((a * nbq + q) * nw * nw + i) * 2, dst = ((a * nw * nw + i) * nbq + q) * 2;
                        o[dst] = v[src]; o[dst + 1] = v[src + 1];
                    }
                    v.swap(o);
                };
                turn(tot, 1); turn(bins, (size_t)s->nb4);
            }
            R[name] = tot; R[std::string(name) + "_bins"] = bins; s->pack[name] = d;
        };
        reduce4(s->acc4, n4, "g4iw", 1.0 / (beta * beta), true);
        if (s->acc4pp) reduce4(s->acc4pp, n4, "g4iw_pp", 1.0 / (beta * beta), true);
        if (s->acc2) reduce4(s->acc2, ng2, "g2iw", 1.0 / beta, false);
    }
    if (s->cur_isector == 4 && (C.acc_g4w || C.acc_g4wpp)) {
        // wormScaling(SectorG4) = CntSampling(G4) / (CntMeas(G4) CntSampling(Z)) * 16/beta^2 / mean_sign (CTQMC.F90:8825-8849,
        // 8883-8915); counts are pooled over the chains that share an accumulator bin
        const size_t n = (size_t)4 * s->n4iwf * s->n4iwf * (2 * s->n4iwb + 1) * 2;
        const int nbw = s->nbw;
        std::vector<double> cw(nbw, 0.0), cz(nbw, 0.0), mw(nbw, 0.0);
        double cwt = 0.0, czt = 0.0, mwt = 0.0;
        for (int w = 0; w < W; ++w) {
            const int b = (w / C.walkers_per_bin) % nbw;
            cw[b] += (double)ws[w].cnt[C_W_SAMPLING]; cz[b] += (double)ws[w].cnt[C_SAMPLING];
        }
        for (int b = 0; b < B; ++b) mw[b % nbw] += nwm[b];
        for (int b = 0; b < nbw; ++b) { cwt += cw[b]; czt += cz[b]; mwt += mw[b]; }
        auto scal = [&](double c_w, double c_z, double m_w) { return m_w != 0.0 ? c_w / (m_w * c_z) * 16.0 / (beta * beta) / ms : 0.0; };
        auto reducew = [&](const double *dev, const char *name) {
            std::vector<double> raw((size_t)nbw * n), tot(n, 0.0);
            (void)hipMemcpy(raw.data(), dev, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            PackDesc d{dev, nbw, n, std::vector<double>(nbw), 0};
            for (int b = 0; b < nbw; ++b) d.scale[b] = scal(cw[b], cz[b], mw[b]);
            s->pack[name] = d;
            for (int b = 0; b < nbw; ++b) {
                const double f = scal(cw[b], cz[b], mw[b]);
                for (size_t i = 0; i < n; ++i) { tot[i] += raw[(size_t)b * n + i]; raw[(size_t)b * n + i] *= f; }
            }
            const double f = scal(cwt, czt, mwt);
            for (size_t i = 0; i < n; ++i) tot[i] *= f;
            R[name] = tot; R[std::string(name) + "_bins"] = raw;
        };
        if (C.acc_g4w) reducew(C.acc_g4w, "g4iw_worm");
        if (C.acc_g4wpp) reducew(C.acc_g4wpp, "g4iwpp_worm");
        R["cntmeas_worm"] = {mwt};
    }
    if (s->cur_isector == 2 && (C.acc_giww || C.acc_gtauw)) {
        // wormScaling(SectorG) = CntSampling(G) / (CntMeas(G) CntSampling(Z)) * 4/beta / mean_sign; giw_worm = -sum * scaling
        // / (2Nd)^2, gtau_worm = sum * scaling * Ntau/beta / (2Nd)^2 (CTQMC.F90:8825, 8843-8876); counts pooled per bin
        std::vector<double> cw(B, 0.0), cz(B, 0.0);
        double cwt = 0.0, czt = 0.0, mwt = 0.0;
        for (int w = 0; w < W; ++w) { const int b = w / C.walkers_per_bin; cw[b] += (double)ws[w].cnt[C_W_SAMPLING]; cz[b] += (double)ws[w].cnt[C_SAMPLING]; }
        for (int b = 0; b < B; ++b) { cwt += cw[b]; czt += cz[b]; mwt += nwm[b]; }
        auto scal = [&](double c_w, double c_z, double m_w) { return m_w != 0.0 ? c_w / (m_w * c_z) * 4.0 / beta / ms / ((double)nf * nf) : 0.0; };
        auto reduceg = [&](const double *dev, size_t n, const char *name, double extra) {
            std::vector<double> raw((size_t)B * n), tot(n, 0.0);
            (void)hipMemcpy(raw.data(), dev, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            PackDesc d{dev, B, n, std::vector<double>(B), 0};
            for (int b = 0; b < B; ++b) d.scale[b] = scal(cw[b], cz[b], nwm[b]) * extra;
            s->pack[name] = d;
            for (int b = 0; b < B; ++b) {
                const double f = scal(cw[b], cz[b], nwm[b]) * extra;
                for (size_t i = 0; i < n; ++i) { tot[i] += raw[(size_t)b * n + i]; raw[(size_t)b * n + i] *= f; }
            }
            const double f = scal(cwt, czt, mwt) * extra;
            for (size_t i = 0; i < n; ++i) tot[i] *= f;
            R[name] = tot; R[std::string(name) + "_bins"] = raw;
        };
        if (C.acc_giww) reduceg(C.acc_giww, (size_t)C.niw2 * 2, "giw_worm", -1.0);
        if (C.acc_gtauw) reduceg(C.acc_gtauw, (size_t)ntau, "gtau_worm", (double)ntau / beta);
        R["cntmeas_worm"] = {mwt};
    }
    if (s->cur_isector == 3 && s->worm_meas_gsigma) {
        // GSigmaiw_worm = sum * wormScaling(GSigma) / (-4 (2 Nd)^2), wormNorm = 32 / beta (CTQMC.F90:8826, 8843-8849, 8987-8998); counts pooled per bin
        std::vector<double> cw(B, 0.0), cz(B, 0.0);
        double cwt = 0.0, czt = 0.0, mwt = 0.0;
        for (int w = 0; w < W; ++w) { const int b = w / C.walkers_per_bin; cw[b] += (double)ws[w].cnt[C_W_SAMPLING]; cz[b] += (double)ws[w].cnt[C_SAMPLING]; }
        for (int b = 0; b < B; ++b) { cwt += cw[b]; czt += cz[b]; mwt += nwm[b]; }
        auto scal = [&](double c_w, double c_z, double m_w) { return m_w != 0.0 ? c_w / (m_w * c_z) * 32.0 / beta / ms / (-4.0 * (double)nf * nf) : 0.0; };
        const size_t n = (size_t)C.niw2 * 2;
        std::vector<double> raw((size_t)B * n), tot(n, 0.0);
        (void)hipMemcpy(raw.data(), C.acc_giww, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
        PackDesc d{C.acc_giww, B, n, std::vector<double>(B), 0};
        for (int b = 0; b < B; ++b) d.scale[b] = scal(cw[b], cz[b], nwm[b]);
        s->pack["gsigmaiw_worm"] = d;
        for (int b = 0; b < B; ++b) { const double f = scal(cw[b], cz[b], nwm[b]); for (size_t i = 0; i < n; ++i) { tot[i] += raw[(size_t)b * n + i]; raw[(size_t)b * n + i] *= f; } }
        const double f = scal(cwt, czt, mwt);
        for (size_t i = 0; i < n; ++i) tot[i] *= f;
        R["gsigmaiw_worm"] = tot; R["gsigmaiw_worm_bins"] = raw; R["cntmeas_worm"] = {mwt};
    }
    if (is_wie_sector(s->cur_isector) && s->nbie > 0) {
        // wormScaling = CntSampling(worm) / (CntMeas(worm) CntSampling(Z)) * wormNorm / mean_sign with wormNorm = 128/beta^2 (5), 256/beta (10), 65536/beta^2 (11),
        // 4096/beta^2 (12), 8192/beta^2 (13), 256/beta^2 (14 - 16) (CTQMC.F90:8829-8849), times u_sparsity (all but sector 5; 9013-9343), times
        // 1/(2Nd)^2 for qq / qqtau, times Ntau/beta for the tau histograms.  Sectors 5, 10 - 13 count a measurement only when their array is kept
        // (8600-8629); counts pooled over the chains that share an accumulator bin
        const int is = s->cur_isector, nbe = s->nbie;
        const WieSector e = wie_sector(s, is);
        static const double num[17] = {0, 0, 0, 0, 0, 128.0, 0, 0, 0, 0, 256.0, 65536.0, 4096.0, 8192.0, 256.0, 256.0, 256.0};
        const bool counted = !e.needs_iw || s->worm_meas_more[e.q_iw];
        std::vector<double> cw(nbe, 0.0), cz(nbe, 0.0), mw(nbe, 0.0);
        double cwt = 0.0, czt = 0.0, mwt = 0.0;
        for (int w = 0; w < W; ++w) { const int b = (w / C.walkers_per_bin) % nbe; cw[b] += (double)ws[w].cnt[C_W_SAMPLING]; cz[b] += (double)ws[w].cnt[C_SAMPLING]; }
        if (counted) for (int b = 0; b < B; ++b) mw[b % nbe] += nwm[b];
        for (int b = 0; b < nbe; ++b) { cwt += cw[b]; czt += cz[b]; mwt += mw[b]; }
        const double norm = (is == 10 ? num[is] / beta : num[is] / (beta * beta)) * (is == 5 ? 1.0 : s->wie_sparsity) / (is == 10 ? (double)nf * nf : 1.0);
        auto scal = [&](double c_w, double c_z, double m_w) { return m_w != 0.0 ? c_w / (m_w * c_z) * norm / ms : 0.0; };
        auto reducei = [&](const double *dev, size_t n, size_t stride, const char *name, double extra) {
            std::vector<double> all((size_t)nbe * stride), raw((size_t)nbe * n), tot(n, 0.0);
            (void)hipMemcpy(all.data(), dev, all.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int b = 0; b < nbe; ++b) for (size_t i = 0; i < n; ++i) raw[(size_t)b * n + i] = all[(size_t)b * stride + i];
            if (stride == n) {              // the packed cross-rank reduction reads the device array as [bins][n]
                PackDesc d{dev, nbe, n, std::vector<double>(nbe), 0};
                for (int b = 0; b < nbe; ++b) d.scale[b] = scal(cw[b], cz[b], mw[b]) * extra;
                s->pack[name] = d;
            }
            for (int b = 0; b < nbe; ++b) {
                const double f = scal(cw[b], cz[b], mw[b]) * extra;
                for (size_t i = 0; i < n; ++i) { tot[i] += raw[(size_t)b * n + i]; raw[(size_t)b * n + i] *= f; }
            }
            const double f = scal(cwt, czt, mwt) * extra;
            for (size_t i = 0; i < n; ++i) tot[i] *= f;
            R[name] = tot; R[std::string(name) + "_bins"] = raw;
        };
        const size_t niw_e = (size_t)e.n1 * e.n2 * e.n3 * 2;
        if (s->worm_meas_more[e.q_iw]) reducei(s->acc_wie_iw, niw_e, niw_e, kWieNames[e.q_iw], 1.0);
        if (e.q_tau >= 0 && s->worm_meas_more[e.q_tau] && counted) reducei(s->acc_wie_tau, (size_t)ntau, (size_t)ntau, kWieNames[e.q_tau], (double)ntau / beta);
        R["cntmeas_worm"] = {mwt};
    }
    if (s->cur_isector >= 6 && s->cur_isector <= 9 && s->nbe > 0) {
        // wormScaling = CntSampling(worm) / (CntMeas(worm) CntSampling(Z)) * wormNorm / mean_sign, wormNorm = 16/beta (P2, P2pp) or
        // 16/beta^2 (P3, P3pp); the tau histograms times Ntau/beta (CTQMC.F90:8830-8849, 9116-9130, 9165-9179, 9215-9224, 9229-9313);
        // counts pooled over the chains that share an accumulator bin.  The division by (2 Nd)^4 is the caller's (impurity.py:817-839)
        const int is = s->cur_isector, nbe = s->nbe;
        std::vector<double> cw(nbe, 0.0), cz(nbe, 0.0), mw(nbe, 0.0);
        double cwt = 0.0, czt = 0.0, mwt = 0.0;
        for (int w = 0; w < W; ++w) { const int b = (w / C.walkers_per_bin) % nbe; cw[b] += (double)ws[w].cnt[C_W_SAMPLING]; cz[b] += (double)ws[w].cnt[C_SAMPLING]; }
        for (int b = 0; b < B; ++b) mw[b % nbe] += nwm[b];
        for (int b = 0; b < nbe; ++b) { cwt += cw[b]; czt += cz[b]; mwt += mw[b]; }
        const double norm = is <= 7 ? 16.0 / beta : 16.0 / (beta * beta);
        auto scal = [&](double c_w, double c_z, double m_w) { return m_w != 0.0 ? c_w / (m_w * c_z) * norm / ms : 0.0; };
        auto reducee = [&](const double *dev, size_t n, const char *name, double extra) {
            std::vector<double> raw((size_t)nbe * n), tot(n, 0.0);
            (void)hipMemcpy(raw.data(), dev, raw.size() * sizeof(double), hipMemcpyDeviceToHost);
            PackDesc d{dev, nbe, n, std::vector<double>(nbe), 0};
            for (int b = 0; b < nbe; ++b) d.scale[b] = scal(cw[b], cz[b], mw[b]) * extra;
            s->pack[name] = d;
            for (int b = 0; b < nbe; ++b) {
                const double f = scal(cw[b], cz[b], mw[b]) * extra;
                for (size_t i = 0; i < n; ++i) { tot[i] += raw[(size_t)b * n + i]; raw[(size_t)b * n + i] *= f; }
            }
            const double f = scal(cwt, czt, mwt) * extra;
            for (size_t i = 0; i < n; ++i) tot[i] *= f;
            R[name] = tot; R[std::string(name) + "_bins"] = raw;
        };
        if (is == 6 && s->worm_meas_eq[0]) reducee(s->acc_weq_iw, (size_t)2 * (2 * s->n2iwb + 1), "p2iw_worm", 1.0);
        if (is == 7 && s->worm_meas_eq[1]) reducee(s->acc_weq_iw, (size_t)2 * (2 * s->n2iwb + 1), "p2iwpp_worm", 1.0);
        if (is == 6 && s->worm_meas_eq[2]) reducee(s->acc_weq_tau, (size_t)ntau, "p2tau_worm", (double)ntau / beta);
        if (is == 7 && s->worm_meas_eq[3]) reducee(s->acc_weq_tau, (size_t)ntau, "p2taupp_worm", (double)ntau / beta);
        if (is == 8 && s->worm_meas_eq[4]) reducee(s->acc_weq_p3, (size_t)4 * s->n3iwf * (2 * s->n3iwb + 1), "p3iw_worm", 1.0);
        if (is == 9 && s->worm_meas_eq[5]) reducee(s->acc_weq_p3, (size_t)4 * s->n3iwf * (2 * s->n3iwb + 1), "p3iwpp_worm", 1.0);
        R["cntmeas_worm"] = {mwt};
    }
    {   // per-chain worm counters: steps in Z, steps in worm space, worm insertions tried / accepted, removals, replacements
        std::vector<double> wc((size_t)W * 8);
        for (int w = 0; w < W; ++w) {
            const unsigned long long *c = ws[w].cnt;
            const int idx[8] = {C_SAMPLING, C_W_SAMPLING, C_TRYWADD, C_ACCWADD, C_TRYWREM, C_ACCWREM, C_TRYWREP, C_ACCWREP};
            for (int i = 0; i < 8; ++i) wc[(size_t)w * 8 + i] = (double)c[idx[i]];
        }
        R["worm_counters"] = wc;
    }
    R["eigenstatesenergies"] = s->energies;
    // lhisto / rhisto: the reference never fills DTrace%lhisto / rhisto (only zeroed, Trace.F90:596-597), so its results
    // are 0/0 = NaN and 0/(sum of acceptance ratios) = 0 (CTQMC.F90:8698-8699)
    R["lhisto"] = std::vector<double>(100, std::nan("")); R["rhisto"] = std::vector<double>(100, 0.0);
    if (s->offdiag) {      // CTQMC.F90:8752-8753, 8768, 8771
        std::vector<double> glf = fetch(s->acc_gleg_full, (size_t)B * nf * nf * nleg), gtf = fetch(s->acc_gtau_full, (size_t)B * nf * nf * ntau);
        if (fetch_err != hipSuccess) return fail(s, W2DYN_EHIP, "measure_finish: copying the accumulators to the host: %s", hipGetErrorString(fetch_err));
        reduce_bins(glf, (size_t)nf * nf * nleg, "gleg_full", [&](double v, double n, size_t i) { int l = (int)(i % nleg); double zs = -1.0 / (n * ms); return -(v * std::sqrt((double)(2 * l) + 1.0) / beta) * zs; }, s->acc_gleg_full);
        reduce_bins(gtf, (size_t)nf * nf * ntau, "gtau_full", [&](double v, double n, size_t) { double zs = -1.0 / (n * ms); return -v * zs * (double)ntau / (beta * beta); }, s->acc_gtau_full);
    } else {   // gtau_full / gleg_full are only filled by the off-diagonal estimators (CTQMC.F90:8752, 8768): zero here
        R["gtau_full"] = std::vector<double>((size_t)nf * nf * ntau, 0.0); R["gleg_full"] = std::vector<double>((size_t)nf * nf * nleg, 0.0);
    }
    return 0;
}

// sector bookkeeping shared by ctqmc_measure and count_steps: every chain starts in Z space (CTQMC.F90:6328, 8400)
static int enter_sector(w2dyn_solver *s, int isector, int icomponent)
{
    if (isector != 1 && s->offdiag)
        return fail(s, W2DYN_EUNSUPPORTED, "worm sampling with offdiag = 1 is not supported");
    if (isector != 1 && s->C.xmoves && (!s->mat || mat_variant(s) > 4 || isector == 5 || isector >= 10))
        return fail(s, W2DYN_EUNSUPPORTED, "worm sampling with Percentage4OperatorMove or PercentageOuterMove > 0 runs on the matrix trace path (which init_solver takes when "
                    "PercentageWormInsert > 0 as well) with blocks of up to four states, in the sectors 2, 3, 4 and 6 - 9; not supported here (sector %d, largest block %d)",
                    isector, s->lp.nsmax);
    if (isector < 1 || isector > 16)
        return fail(s, W2DYN_EUNSUPPORTED, "worm sector %d does not exist (1 = Z, 2 - 16: Parameters.F90:28-43)", isector);
    if (isector != 1) {
        const int nf = s->nflav, nf4 = nf * nf * nf * nf;
        const bool two = isector == 2 || isector == 3 || isector == 10;          // components of two flavour indices
        const bool ufac = isector == 3 || isector == 5 || isector >= 10;
        if (icomponent < (ufac ? 1 : 0) || icomponent > (two ? nf * nf : nf4))
            return fail(s, W2DYN_EINVAL, "sector %d: component %d out of range (%d .. %d)", isector, icomponent, ufac ? 1 : 0, two ? nf * nf : nf4);
        s->wie_sparsity = 1.0;
        if (ufac) {
            // the insertion draws flavours until the U-matrix factor of the proposal does not vanish (CTQMC.F90:5047-5054): the reference loops for
            // ever when no choice has one; here that is an error of the call.  The factor is a product over disjoint sets of operators, so it can be
            // non-zero iff every antisymmetrised slice it reads has a non-vanishing element; the same counts make u_sparsity (9013-9343).
            int u[5] = {0, 0, 0, 0, 0}, rest = icomponent - 1;
            for (int i = two ? 2 : 4; i >= 1; --i) { u[i] = rest % nf; rest /= nf; }
            const std::vector<double> &U = s->umat_h;
            auto UM = [&](int a, int b, int c, int d) { return U[(((size_t)a * nf + b) * nf + c) * nf + d]; };
            auto UA1 = [&](int a, int b, int c, int d) { return 0.5 * (UM(a, b, c, d) - UM(b, a, c, d)); };
            auto UA2 = [&](int a, int b, int c, int d) { return 0.5 * (UM(a, b, c, d) - UM(a, b, d, c)); };
            auto UAB = [&](int a, int b, int c, int d) { return 0.5 * (UA1(a, b, c, d) - UA1(a, b, d, c)); };
            // numbers of non-vanishing elements: fixed indices >= 0, free ones -1
            auto nnz = [&](int kind, int a, int b, int c, int d) {
                double cnt = 0.0;
                for (int i = 0; i < nf4; ++i) {
                    const int x[4] = {i / (nf * nf * nf), (i / (nf * nf)) % nf, (i / nf) % nf, i % nf}, fix[4] = {a, b, c, d};
                    bool first = true;                                     // count every combination of the free indices once
                    for (int q = 0; q < 4; ++q) if (fix[q] >= 0 && x[q] != 0) first = false;
                    if (!first) continue;
                    const int y[4] = {a >= 0 ? a : x[0], b >= 0 ? b : x[1], c >= 0 ? c : x[2], d >= 0 ? d : x[3]};
                    const double v = kind == 1 ? UA1(y[0], y[1], y[2], y[3]) : (kind == 2 ? UA2(y[0], y[1], y[2], y[3]) : UAB(y[0], y[1], y[2], y[3]));
                    if (v != 0.0) cnt += 1.0;
                }
                return cnt;
            };
            const double n2 = (double)nf * nf, n4 = n2 * n2;
            double can = 1.0, sp = 1.0;          // can: product of the counts of the proposal's own slices (0: the worm never gets in)
            switch (isector) {
            case 3: case 5: can = nnz(1, u[1], -1, -1, -1); break;
            case 10: can = sp = nnz(1, u[1], -1, -1, -1) * nnz(2, -1, -1, u[2], -1); sp /= n4 * n2; break;
            case 11: can = sp = nnz(1, u[1], -1, -1, -1) * nnz(2, -1, -1, u[2], -1) * nnz(1, u[3], -1, -1, -1) * nnz(2, -1, -1, u[4], -1); sp /= n4 * n4 * n4; break;
            case 12: can = sp = nnz(1, u[1], -1, -1, -1) * nnz(2, -1, -1, u[2], -1) * nnz(3, u[3], -1, u[4], -1); sp /= n4 * n4; break;
            case 13: can = nnz(1, u[1], -1, -1, -1) * nnz(1, u[3], -1, -1, -1) * nnz(2, -1, -1, u[2], u[4]);
                     sp = nnz(1, u[1], -1, -1, -1) * nnz(2, -1, u[3], -1, -1) * nnz(3, -1, -1, u[2], u[4]) / (n4 * n4); break;
            case 14: can = sp = nnz(3, u[1], -1, u[2], -1) * nnz(3, u[3], -1, u[4], -1); sp /= n4; break;
            case 15: can = sp = nnz(2, -1, -1, u[2], u[4]) * nnz(1, u[1], u[3], -1, -1); sp /= n4; break;
            default: can = sp = nnz(1, u[1], -1, -1, -1) * nnz(3, u[3], -1, u[2], u[4]); sp /= n4; break;
            }
            if (can == 0.0) return fail(s, W2DYN_EINVAL, "sector %d, component %d: the U matrix has no element that lets the worm in", isector, icomponent);
            s->wie_sparsity = sp;
        }
    }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (isector == 1) s->p_wins = 0.0;                         // CTQMC.F90:8380-8385: a pure Z run switches worm insertion off for good
    s->cur_isector = isector;
    s->C.isector = isector; s->C.icomponent = icomponent; s->C.p_wins = isector == 1 ? 0.0 : s->p_wins; s->C.p_wrep = s->p_wrep;
    s->C.worm_eta = s->worm_eta;
    std::vector<WalkerScalars> ws(s->nwalkers);
    HIPCHK(hipMemcpy(ws.data(), s->C.ws, ws.size() * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    bool dirty = false;
    for (auto &w : ws) if (w.sector > 1) { w.sector = 1; dirty = true; }
    if (dirty) HIPCHK(hipMemcpy(s->C.ws, ws.data(), ws.size() * sizeof(WalkerScalars), hipMemcpyHostToDevice));
    return 0;
}

int w2dyn_ctqmc_measure(w2dyn_solver *s, int isector, int icomponent)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "ctqmc_measure before init_solver");
    // whatever way this call ends - result, soft abort, error - the context is back in Z space and gives up its reference
    // on the signal handlers (CTQMC.F90:9352-9355)
    struct Leave {
        w2dyn_solver *s;
        ~Leave() { s->cur_isector = 1; s->C.isector = 1; s->C.p_wins = 0.0; release_traps(s); }
    } leave{s};
    int rc;
    if ((rc = enter_sector(s, isector, icomponent))) return rc;
    if ((rc = w2dyn_measure_begin(s))) return rc;
    double t0 = now_s(), tsteps = 0.0;
    s->no_sampling = isector != 1 && s->worm_eta == 0.0;       // CTQMC.F90:8409-8418: nothing is sampled with eta = 0
    // Launches are asynchronous; the host waits for the stream every MEAS_SYNC measurements, so that it never runs more than
    // that far ahead of the GPU and a signal (SIGTERM / SIGINT / SIGUSR1, CTQMC.F90:8430) ends the run within that many
    // measurements: the measurements done so far are normalised and `aborted` says so (CTQMC.F90:8648-8658)
    const int64_t MEAS_SYNC = 32;
    bool interrupted = false;
    for (int64_t m = 0; m < s->nmeas && !s->no_sampling; ++m) {
        if (g_signal_fired) { interrupted = true; break; }
        double a = now_s();
        if ((rc = sweep_impl(s, s->ncorr, 1))) return rc;
        tsteps += now_s() - a;
        if ((rc = measure_once_impl(s, s->G2))) return rc;
        if ((m % MEAS_SYNC) == MEAS_SYNC - 1) { if ((rc = grow_if_close(s))) return rc; }
    }
    if ((rc = w2dyn_sync(s))) return rc;
    s->time_sim = now_s() - t0; s->time_sim_steps = tsteps;
    rc = w2dyn_measure_finish(s);
    if (interrupted && rc == 0) s->aborted = s->meas_done == 0 ? 2 : 1;
    return rc;
}

int w2dyn_set_sector(w2dyn_solver *s, int isector, int icomponent)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "set_sector before init_solver");
    return enter_sector(s, isector, icomponent);
}

int w2dyn_set_wormeta(w2dyn_solver *s, double eta) { if (!s) return W2DYN_EINVAL; s->worm_eta = eta; s->C.worm_eta = eta; return 0; }
int w2dyn_get_wormeta(w2dyn_solver *s, double *eta) { if (!s || !eta) return W2DYN_EINVAL; *eta = s->worm_eta; return 0; }

int w2dyn_ctqmc_worm_warmup(w2dyn_solver *s, int64_t nwarmups2, int isector, int icomponent)
{
    (void)nwarmups2; (void)icomponent;
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "ctqmc_worm_warmup before init_solver");
    // the reference allocates the buffers of the batched transform here and samples nothing (CTQMC.F90:8266-8348); this
    // library transforms every measurement on the spot, so only the arguments are checked
    if (isector < 1 || isector > 16) return fail(s, W2DYN_EUNSUPPORTED, "worm sector %d does not exist (1 = Z, 2 - 16: Parameters.F90:28-43)", isector);
    return 0;
}

int w2dyn_count_steps(w2dyn_solver *s, int isector, int icomponent, int64_t nfix, int64_t *steps_worm, int64_t *steps_z)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "count_steps before init_solver");
    if (isector < 2 || isector > 16) return fail(s, W2DYN_EUNSUPPORTED, "count_steps: worm sector %d does not exist (2 - 16: Parameters.F90:29-43)", isector);
    if (nfix < 0) return fail(s, W2DYN_EINVAL, "count_steps: nfix = %lld", (long long)nfix);
    int rc;
    if ((rc = enter_sector(s, isector, icomponent))) return rc;
    const DevConst &C = s->C; const size_t W = s->nwalkers, nf = s->nflav, kmax = C.kmax;
    // starting configuration of every chain (get_mc_config, CTQMC.F90:6322-6325)
    std::vector<WalkerScalars> ws0(W), ws1(W);
    std::vector<int> K0(W * nf); std::vector<double> TC0(W * nf * kmax), TA0(W * nf * kmax);
    HIPCHK(hipMemcpy(ws0.data(), C.ws, W * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(K0.data(), C.K, K0.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(TC0.data(), C.TC, TC0.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(TA0.data(), C.TA, TA0.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int kmax0 = C.kmax;
    if ((rc = sweep_impl(s, nfix, 1))) return rc;      // a signal is left pending for ctqmc_measure, as in the reference
    if ((rc = finish_paused(s, 2))) return rc;
    if ((rc = check_walker_errors(s))) return rc;
    HIPCHK(hipMemcpy(ws1.data(), C.ws, W * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    if (C.kmax != kmax0) {      // the capacity was enlarged during the search: the saved tau arrays go to the new stride
        std::vector<double> tc((size_t)W * nf * C.kmax, 0.0), ta((size_t)W * nf * C.kmax, 0.0);
        for (size_t wf = 0; wf < W * nf; ++wf) for (int e = 0; e < K0[wf]; ++e) { tc[wf * C.kmax + e] = TC0[wf * kmax0 + e]; ta[wf * C.kmax + e] = TA0[wf * kmax0 + e]; }
        TC0.swap(tc); TA0.swap(ta);
    }
    s->steps_worm_chains.assign(W, 0);
    double sw = 0.0;
    for (size_t w = 0; w < W; ++w) { s->steps_worm_chains[w] = (long long)ws1[w].cnt[C_W_SAMPLING]; sw += (double)ws1[w].cnt[C_W_SAMPLING]; }
    const int64_t mean_w = (int64_t)std::llround(sw / (double)W);
    if (steps_worm) *steps_worm = mean_w;
    if (steps_z) *steps_z = nfix - mean_w;
    // init_counters, clear_Trace, set_mc_config (CTQMC.F90:6434-6438): counters zero, old configuration, window at the
    // start, M and det rebuilt from scratch; the random streams are not rewound
    for (size_t w = 0; w < W; ++w) {
        WalkerScalars &x = ws1[w];
        memset(x.cnt, 0, sizeof x.cnt);
        x.outer = ws0[w].outer; x.wpos = 0; x.npairs = -1; x.sector = 1;
    }
    HIPCHK(hipMemcpy(C.ws, ws1.data(), W * sizeof(WalkerScalars), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.K, K0.data(), K0.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.TC, TC0.data(), TC0.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.TA, TA0.data(), TA0.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(C.accpair, 0, sizeof(unsigned long long) * s->nflav * C.naccpair));
    s->cur_isector = 1; s->C.isector = 1; s->C.p_wins = 0.0;
    LaunchCfg l = walker_launch(s);
    launch_rebuild(l, s->C);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_walker_errors(s);
}

int w2dyn_dest_ctqmc(w2dyn_solver *s)
{
    if (!s) return W2DYN_EINVAL;
    (void)hipSetDevice(s->device);
    free_device(s);
    release_traps(s);
    return 0;
}

// ---- configuration access ----------------------------------------------------------------------------------
static int fetch_walker(w2dyn_solver *s, int w, WalkerScalars &ws, std::vector<int> &K, std::vector<double> &TC, std::vector<double> &TA)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "no solver");
    if (w < 0 || w >= s->nwalkers) return fail(s, W2DYN_EINVAL, "walker %d out of range", w);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    const DevConst &C = s->C; const int nf = s->nflav, kmax = C.kmax;
    K.resize(nf); TC.resize((size_t)nf * kmax); TA.resize((size_t)nf * kmax);
    HIPCHK(hipMemcpy(&ws, C.ws + w, sizeof ws, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(K.data(), C.K + (size_t)w * nf, sizeof(int) * nf, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(TC.data(), C.TC + (size_t)w * nf * kmax, sizeof(double) * nf * kmax, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(TA.data(), C.TA + (size_t)w * nf * kmax, sizeof(double) * nf * kmax, hipMemcpyDeviceToHost));
    return 0;
}

int w2dyn_get_mc_config_scalars(w2dyn_solver *s, int walker, int *n, int *outer_sst, int *outer_state)
{
    WalkerScalars ws; std::vector<int> K; std::vector<double> TC, TA;
    int rc = fetch_walker(s, walker, ws, K, TC, TA); if (rc) return rc;
    int tot = 0; for (int k : K) tot += 2 * k;
    if (ws.sector > 1) tot += worm_nops_of(ws.sector);
    if (n) *n = tot; if (outer_sst) *outer_sst = ws.outer; if (outer_state) *outer_state = 1;
    return 0;
}

int w2dyn_get_mc_config(w2dyn_solver *s, int walker, int n, double *taus, int *orbs, int *spins, int *cas, int *hashybs,
                        int *outer_sst, int *outer_state)
{
    WalkerScalars ws; std::vector<int> K; std::vector<double> TC, TA;
    int rc = fetch_walker(s, walker, ws, K, TC, TA); if (rc) return rc;
    struct Op { double t; int f, ca, hyb; };
    std::vector<Op> ops;
    const int kmax = s->C.kmax;
    if (s->offdiag) {      // all operators sit in the slots of flavour 0, their flavours in the tag arrays
        std::vector<int> fc(K[0]), fa(K[0]);
        if (K[0]) { HIPCHK(hipMemcpy(fc.data(), s->C.FC + (size_t)walker * kmax, sizeof(int) * K[0], hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(fa.data(), s->C.FA + (size_t)walker * kmax, sizeof(int) * K[0], hipMemcpyDeviceToHost)); }
        for (int e = 0; e < K[0]; ++e) { ops.push_back({TC[e], fc[e], 1, -1}); ops.push_back({TA[e], fa[e], 2, -1}); }
    } else
    for (int f = 0; f < s->nflav; ++f) for (int e = 0; e < K[f]; ++e) { ops.push_back({TC[(size_t)f * kmax + e], f, 1, -1}); ops.push_back({TA[(size_t)f * kmax + e], f, 2, -1}); }
    if (ws.sector > 1) for (int i = 0; i < worm_nops_of(ws.sector); ++i) ops.push_back({ws.wtau[i], ws.wflav[i], (i & 1) ? 1 : 2, 0});     // worm operators: hashyb 0 (Trace.F90:6861-6865)
    if ((int)ops.size() != n) return fail(s, W2DYN_EINVAL, "get_mc_config: n=%d but the chain holds %zu operators", n, ops.size());
    std::sort(ops.begin(), ops.end(), [](const Op &a, const Op &b) { return a.t < b.t; });
    for (int i = 0; i < n; ++i) { taus[i] = ops[i].t; orbs[i] = (ops[i].f >> 1) + 1; spins[i] = (ops[i].f & 1) + 1; cas[i] = ops[i].ca; hashybs[i] = ops[i].hyb; }
    if (outer_sst) *outer_sst = ws.outer; if (outer_state) *outer_state = 1;
    return 0;
}

int w2dyn_set_mc_config(w2dyn_solver *s, int walker, int n, const double *taus, const int *orbs, const int *spins, const int *cas,
                        const int *hashybs, int outer_sst, int outer_state)
{
    (void)outer_state;
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "no solver");
    if (walker < -1 || walker >= s->nwalkers) return fail(s, W2DYN_EINVAL, "walker %d out of range", walker);
    if (s->offdiag) return fail(s, W2DYN_EUNSUPPORTED, "set_mc_config with offdiag = 1 is not supported");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    const DevConst &C = s->C; const int nf = s->nflav, kmax = C.kmax, nd = s->nd;
    std::vector<int> K(nf, 0); std::vector<double> TC((size_t)nf * kmax, 0.0), TA((size_t)nf * kmax, 0.0);
    (void)nd;
    if (outer_sst < 0 || outer_sst >= C.nouter) return fail(s, W2DYN_EINVAL, "set_mc_config: outer superstate %d out of range", outer_sst);
    int state = outer_sst;
    for (int i = 0; i < n; ++i) {
        if (hashybs && hashybs[i] == 0) return fail(s, W2DYN_EUNSUPPORTED, "worm operators in set_mc_config are not supported");
        int f = 2 * (orbs[i] - 1) + (spins[i] - 1);
        if (f < 0 || f >= nf || (cas[i] != 1 && cas[i] != 2)) return fail(s, W2DYN_EINVAL, "bad operator %d", i);
        if (i > 0 && !(taus[i] > taus[i - 1])) return fail(s, W2DYN_EINVAL, "set_mc_config: taus must be strictly increasing");
        state = s->lp.connect(state, f, cas[i]);
        if (state < 0) return fail(s, W2DYN_EINVAL, "set_mc_config: operator %d violates the quantum numbers", i);
    }
    if (state != outer_sst) return fail(s, W2DYN_EINVAL, "set_mc_config: sequence does not return to the outer superstate");
    std::vector<int> nc(nf, 0), na(nf, 0);
    for (int i = 0; i < n; ++i) {
        int f = 2 * (orbs[i] - 1) + (spins[i] - 1);
        int &cnt = (cas[i] == 1) ? nc[f] : na[f];
        if (cnt >= kmax) return fail(s, W2DYN_EOVERFLOW, "set_mc_config: flavour %d exceeds kmax=%d", f, kmax);
        ((cas[i] == 1) ? TC : TA)[(size_t)f * kmax + cnt] = taus[i];
        ++cnt;
    }
    for (int f = 0; f < nf; ++f) K[f] = nc[f];
    int w0 = walker < 0 ? 0 : walker, w1 = walker < 0 ? s->nwalkers : walker + 1;   // walker = -1: every chain
    for (int w = w0; w < w1; ++w) {
        WalkerScalars ws;
        HIPCHK(hipMemcpy(&ws, C.ws + w, sizeof ws, hipMemcpyDeviceToHost));
        ws.outer = outer_sst; ws.wpos = 0; ws.npairs = -1; ws.sector = 1;          // Trace.F90:6917-6919, 6973-6975
        HIPCHK(hipMemcpy(C.ws + w, &ws, sizeof ws, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(C.K + (size_t)w * nf, K.data(), sizeof(int) * nf, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(C.TC + (size_t)w * nf * kmax, TC.data(), sizeof(double) * nf * kmax, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(C.TA + (size_t)w * nf * kmax, TA.data(), sizeof(double) * nf * kmax, hipMemcpyHostToDevice));
    }
    // M, det and NPairs of every chain rebuilt on the device (Trace.F90:6990-6999)
    LaunchCfg l = walker_launch(s);
    launch_rebuild(l, s->C);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_walker_errors(s);
}

int w2dyn_save_mc_configs(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "save_mc_configs before init_solver");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    const DevConst &C = s->C; const size_t W = s->nwalkers, nf = s->nflav, kmax = C.kmax;
    s->snap_ws.resize(W); s->snap_K.resize(W * nf); s->snap_TC.resize(W * nf * kmax); s->snap_TA.resize(W * nf * kmax);
    HIPCHK(hipMemcpy(s->snap_ws.data(), C.ws, W * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s->snap_K.data(), C.K, W * nf * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s->snap_TC.data(), C.TC, W * nf * kmax * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s->snap_TA.data(), C.TA, W * nf * kmax * sizeof(double), hipMemcpyDeviceToHost));
    for (auto &w : s->snap_ws) if (w.sector > 1) return fail(s, W2DYN_EINVAL, "save_mc_configs: a chain is in worm space");
    s->snap_nf = (int)nf; s->snap_kmax = (int)kmax;
    return 0;
}

int w2dyn_restore_mc_configs(w2dyn_solver *s)
{
    if (!s || !s->have_solver) return fail(s, W2DYN_EINVAL, "restore_mc_configs before init_solver");
    const DevConst &C = s->C; const size_t W = s->nwalkers, nf = s->nflav, kmax = C.kmax;
    if (s->snap_ws.size() != W || s->snap_nf != (int)nf)
        return fail(s, W2DYN_EINVAL, "restore_mc_configs: no saved configurations of this shape (chains %zu/%zu, flavours %d/%zu)",
                    s->snap_ws.size(), W, s->snap_nf, nf);
    if (s->snap_kmax != (int)kmax) {      // saved at another capacity (it was enlarged in between, or set differently): re-stride
        int need = 0; for (int k : s->snap_K) need = std::max(need, k);
        if (need > (int)kmax) { int rc = grow_kmax(s, need + 8); if (rc) return rc; return w2dyn_restore_mc_configs(s); }
        std::vector<double> tc(W * nf * kmax, 0.0), ta(W * nf * kmax, 0.0);
        for (size_t wf = 0; wf < W * nf; ++wf) for (int e = 0; e < s->snap_K[wf]; ++e) { tc[wf * kmax + e] = s->snap_TC[wf * s->snap_kmax + e]; ta[wf * kmax + e] = s->snap_TA[wf * s->snap_kmax + e]; }
        s->snap_TC.swap(tc); s->snap_TA.swap(ta); s->snap_kmax = (int)kmax;
    }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<WalkerScalars> ws(W);
    HIPCHK(hipMemcpy(ws.data(), C.ws, W * sizeof(WalkerScalars), hipMemcpyDeviceToHost));
    for (size_t w = 0; w < W; ++w) {       // the configuration only: counters and window start afresh, the random streams move on
        ws[w].outer = s->snap_ws[w].outer;          // (CtHybConfig carries no generator state, impurity.py:45-107)
        ws[w].wpos = 0; ws[w].npairs = -1; ws[w].sector = 1;                   // Trace.F90:6917-6919, 6973-6975
    }
    HIPCHK(hipMemcpy(C.ws, ws.data(), W * sizeof(WalkerScalars), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.K, s->snap_K.data(), W * nf * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.TC, s->snap_TC.data(), W * nf * kmax * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(C.TA, s->snap_TA.data(), W * nf * kmax * sizeof(double), hipMemcpyHostToDevice));
    LaunchCfg l = walker_launch(s);
    launch_rebuild(l, s->C);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_walker_errors(s);
}

int w2dyn_get_minv(w2dyn_solver *s, int walker, int orb, int spin, int *kout, double *out, size_t out_doubles)
{
    WalkerScalars ws; std::vector<int> K; std::vector<double> TC, TA;
    int rc = fetch_walker(s, walker, ws, K, TC, TA); if (rc) return rc;
    if (s->offdiag) {      // orb = spin = 0: the one matrix over all flavours, rows and columns by flavour, then time (MInv_full)
        if (orb != 0 || spin != 0) return fail(s, W2DYN_EINVAL, "get_minv with offdiag = 1: orb = spin = 0 asks for the full matrix");
        const int kmax = s->C.kmax, k = K[0];
        if (kout) *kout = k;
        if (!out || k == 0) return 0;
        if (out_doubles < (size_t)k * k) return fail(s, W2DYN_EINVAL, "get_minv: buffer too small");
        std::vector<double> M((size_t)kmax * kmax); std::vector<int> fc(k), fa(k);
        HIPCHK(hipMemcpy(M.data(), s->C.M + (size_t)walker * s->nflav * kmax * kmax, sizeof(double) * kmax * kmax, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(fc.data(), s->C.FC + (size_t)walker * kmax, sizeof(int) * k, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(fa.data(), s->C.FA + (size_t)walker * kmax, sizeof(int) * k, hipMemcpyDeviceToHost));
        std::vector<int> oc(k), oa(k);
        for (int i = 0; i < k; ++i) oc[i] = oa[i] = i;
        std::sort(oc.begin(), oc.end(), [&](int a, int b) { return fc[a] != fc[b] ? fc[a] < fc[b] : TC[a] < TC[b]; });
        std::sort(oa.begin(), oa.end(), [&](int a, int b) { return fa[a] != fa[b] ? fa[a] < fa[b] : TA[a] < TA[b]; });
        for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) out[(size_t)i * k + j] = M[(size_t)oc[i] * kmax + oa[j]];
        return 0;
    }
    int f = 2 * (orb - 1) + (spin - 1);
    if (f < 0 || f >= s->nflav) return fail(s, W2DYN_EINVAL, "bad flavour");
    const int kmax = s->C.kmax, k = K[f];
    if (kout) *kout = k;
    if (!out || k == 0) return 0;
    if (out_doubles < (size_t)k * k) return fail(s, W2DYN_EINVAL, "get_minv: buffer too small");
    std::vector<double> M((size_t)kmax * kmax);
    HIPCHK(hipMemcpy(M.data(), s->C.M + ((size_t)walker * s->nflav + f) * kmax * kmax, sizeof(double) * kmax * kmax, hipMemcpyDeviceToHost));
    std::vector<int> oc(k), oa(k);
    for (int i = 0; i < k; ++i) oc[i] = oa[i] = i;
    std::sort(oc.begin(), oc.end(), [&](int a, int b) { return TC[(size_t)f * kmax + a] < TC[(size_t)f * kmax + b]; });
    std::sort(oa.begin(), oa.end(), [&](int a, int b) { return TA[(size_t)f * kmax + a] < TA[(size_t)f * kmax + b]; });
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) out[(size_t)i * k + j] = M[(size_t)oc[i] * kmax + oa[j]];
    return 0;
}

int w2dyn_get_walker_state(w2dyn_solver *s, int walker, double *out6)
{
    WalkerScalars ws; std::vector<int> K; std::vector<double> TC, TA;
    int rc = fetch_walker(s, walker, ws, K, TC, TA); if (rc) return rc;
    double cs = 0.0;
    HIPCHK(hipMemcpy(&cs, s->C.cfgsign + walker, sizeof cs, hipMemcpyDeviceToHost));
    // NPairs recounted on the host from the definition (Trace.F90:1770-1815)
    const DevConst &C = s->C; const int kmax = C.kmax;
    double wmin = ((double)ws.wpos / (double)(C.num_windows + 1)) * C.beta, wmax = ((double)(ws.wpos + 2) / (double)(C.num_windows + 1)) * C.beta;
    int np = 0;
    for (int f = 0; f < s->nflav; ++f) for (int i = 0; i < K[f]; ++i) for (int j = 0; j < K[f]; ++j) {
        double a = std::min(TC[(size_t)f * kmax + i], TA[(size_t)f * kmax + j]), b = std::max(TC[(size_t)f * kmax + i], TA[(size_t)f * kmax + j]);
        if (a >= wmin && b <= wmax && !(b > a + C.taudiff_max)) ++np;
    }
    out6[0] = ws.det_sign; out6[1] = ws.det_log; out6[2] = ws.wpos; out6[3] = np; out6[4] = (double)ws.ndraw; out6[5] = cs;
    return 0;
}

// ---- results ---------------------------------------------------------------------------------------------
int w2dyn_get_result(w2dyn_solver *s, const char *name, void *out, size_t bytes, size_t *needed)
{
    if (!s || !name) return W2DYN_EINVAL;
    std::string n(name);
    const void *src = nullptr; size_t size = 0;
    std::vector<int> itmp; std::vector<long long> ltmp; std::vector<double> dtmp;
    if (n == "accpair") { src = s->res_accpair.data(); size = s->res_accpair.size() * sizeof(long long); }
    else if (n == "statessuperstates") {   // block of every eigenstate, block order (CTQMC.F90:8053-8055)
        itmp.resize(s->nstates);
        for (int b = 0; b < s->lp.nsst; ++b) for (int i = 0; i < s->lp.sst_n[b]; ++i) itmp[s->lp.sst_off[b] + i] = b;
        src = itmp.data(); size = itmp.size() * sizeof(int);
    }
    else if (n == "occbasismapping") {   // (NStates, Nd, 2), CTQMC.F90:8059-8065
        itmp.resize((size_t)s->nstates * s->nd * 2);
        for (int i = 0; i < s->nstates; ++i) { int st = s->lp.states[i]; for (int b = 0; b < s->nd; ++b) for (int sp = 0; sp < 2; ++sp) itmp[((size_t)i * s->nd + b) * 2 + sp] = (st >> (b + sp * s->nd)) & 1; }
        src = itmp.data(); size = itmp.size() * sizeof(int);
    }
    else if (n == "accpairtau") { src = s->accpairtau.data(); size = s->accpairtau.size() * sizeof(double); }
    else if (n == "accpairtau_chains") { src = s->accpairtau_chains.data(); size = s->accpairtau_chains.size() * sizeof(double); }
    else if (n == "steps_worm_chains") { src = s->steps_worm_chains.data(); size = s->steps_worm_chains.size() * sizeof(long long); }
    else if (n == "cntsampling") { ltmp.assign(16, 0); if (!s->counters.empty()) { ltmp[0] = (long long)s->counters[C_SAMPLING]; if (s->res_isector > 1) ltmp[s->res_isector - 1] = (long long)s->counters[C_W_SAMPLING]; } src = ltmp.data(); size = ltmp.size() * sizeof(long long); }
    else if (n == "counters") { src = s->counters.data(); size = s->counters.size() * sizeof(double); }
    else {
        auto it = s->res.find(n);
        if (it == s->res.end()) return fail(s, W2DYN_EINVAL, "unknown or not yet available result '%s'", name);
        src = it->second.data(); size = it->second.size() * sizeof(double);
    }
    if (needed) *needed = size;
    if (out && bytes) memcpy(out, src, std::min(bytes, size));
    return 0;
}

int w2dyn_reduce_pack(w2dyn_solver *s, const char *names, void **dev_ptr, size_t *ndoubles, size_t *offsets, size_t *counts, int max_names)
{
    if (!s || !names || !dev_ptr || !ndoubles || !offsets || !counts) return fail(s, W2DYN_EINVAL, "reduce_pack: null argument");
    if (s->pack.empty()) return fail(s, W2DYN_EINVAL, "reduce_pack needs the accumulators of a finished measurement (after ctqmc_measure, before dest_ctqmc / init_solver)");
    HIPCHK(hipSetDevice(s->device));
    std::vector<const PackDesc *> list;
    size_t total = 0, nscale = 0;
    for (const char *p = names; *p;) {
        const char *q = strchr(p, ',');
        const std::string nm = q ? std::string(p, q) : std::string(p);
        p = q ? q + 1 : p + nm.size();
        if (nm.empty()) continue;
        auto it = s->pack.find(nm);
        if (it == s->pack.end()) return fail(s, W2DYN_EUNSUPPORTED, "reduce_pack: no per-bin accumulator named '%s' in this run", nm.c_str());
        if ((int)list.size() >= max_names) return fail(s, W2DYN_EINVAL, "reduce_pack: more than %d names", max_names);
        offsets[list.size()] = total; counts[list.size()] = it->second.per;
        total += 1 + 3 * it->second.per; nscale += (size_t)it->second.B;
        list.push_back(&it->second);
    }
    if (list.empty()) return fail(s, W2DYN_EINVAL, "reduce_pack: empty list of names");
    if (total > s->packbuf_n) { if (s->packbuf) (void)hipFree(s->packbuf); s->packbuf = nullptr; s->packbuf_n = 0; HIPCHK(hipMalloc(&s->packbuf, total * sizeof(double))); s->packbuf_n = total; }
    if (nscale > s->packscale_n) { if (s->packscale) (void)hipFree(s->packscale); s->packscale = nullptr; s->packscale_n = 0; HIPCHK(hipMalloc(&s->packscale, nscale * sizeof(double))); s->packscale_n = nscale; }
    std::vector<double> sc; sc.reserve(nscale);
    for (const PackDesc *d : list) sc.insert(sc.end(), d->scale.begin(), d->scale.end());
    HIPCHK(hipMemcpyAsync(s->packscale, sc.data(), nscale * sizeof(double), hipMemcpyHostToDevice, s->stream));
    size_t so = 0;
    for (size_t i = 0; i < list.size(); ++i) {
        launch_pack_moments(s->stream, list[i]->dev, list[i]->B, list[i]->per, s->packscale + so, list[i]->nleg, s->packbuf + offsets[i]);
        so += (size_t)list[i]->B;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));      // the caller hands the buffer to another stream (torch.distributed)
    *dev_ptr = s->packbuf; *ndoubles = total;
    return 0;
}

int w2dyn_get_scalar(w2dyn_solver *s, const char *name, double *out)
{
    if (!s || !name || !out) return W2DYN_EINVAL;
    std::string n(name);
    auto first = [&](const char *k) { auto it = s->res.find(k); return it == s->res.end() || it->second.empty() ? NAN : it->second[0]; };
    if (n == "mean_sign") *out = s->mean_sign;
    else if (n == "aborted") *out = s->aborted;
    else if (n == "time_warmup") *out = s->time_warmup;
    else if (n == "time_sim") *out = s->time_sim;
    else if (n == "time_sim_steps") *out = s->time_sim_steps;
    else if (n == "accglob") *out = first("accglob");
    else if (n == "accshift") *out = first("accshift");
    else if (n == "accadd4" || n == "accrem4" || n == "accflavc") *out = first(name);
    else if (n == "nwalkers") *out = s->nwalkers;
    else if (n == "nbins") *out = s->C.nbins;
    else if (n == "kmax") *out = s->C.kmax;
    else if (n == "steps_done") *out = (double)s->steps_done;
    else if (n == "meas_done") *out = (double)s->meas_done;
    else if (n == "num_windows") *out = s->C.num_windows;
    else if (n == "nb4") *out = s->nb4;
    else if (n == "nbw") *out = s->nbw;
    else if (n == "nbe") *out = s->nbe;
    else if (n == "nbie") *out = s->nbie;
    else if (n == "trace_path") *out = mat_variant(s);
    else if (n == "n3iwb") *out = s->n3iwb;
    else if (n == "nbl") *out = s->nbl;
    else if (n == "nbs") *out = s->nbs;
    else if (n == "device") *out = s->device;
    else if (n == "wormeta") *out = s->worm_eta;
    else if (n == "ntau") *out = s->ntau;
    else if (n == "n4tau") *out = s->n4tau;
    else if (n == "n4iwf") *out = s->n4iwf;
    else if (n == "n4iwb") *out = s->n4iwb;
    else if (n == "nsst") *out = s->lp.nsst;
    else if (n == "nsmax") *out = s->lp.nsmax;
    else if (n == "matrix_trace") *out = s->mat ? 1 : 0;
    else if (n == "nslide") *out = s->C.nslide;
    else if (n == "scratch_bytes") *out = (double)W2DYN_SCRATCH_BYTES;
    else return fail(s, W2DYN_EINVAL, "unknown scalar '%s'", name);
    return 0;
}

int w2dyn_kernel_time(w2dyn_solver *s, const char *kernel, double *avg_ms, int64_t *launches, int reset)
{
    if (!s || !kernel) return W2DYN_EINVAL;
    { int rc = drain_timers(s); if (rc) return rc; }
    auto &t = s->timers[kernel];
    if (avg_ms) *avg_ms = t.launches ? t.total_ms / (double)t.launches : 0.0;
    if (launches) *launches = t.launches;
    if (reset) t = KTimer{};
    return 0;
}

}  // extern "C"
