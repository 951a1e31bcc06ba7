"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the reference's golden vectors.

Chain of evidence: reference == oracle(std::mt19937) [test_oracle_golden.py, fixtures made by the reference];
oracle(philox, walker w) == HIP chain w [here], same seeded inputs, every move type.  Tolerance 1e-10 relative
(north_star: "within 1e-10 relative in fp64"); configurations 1e-11 absolute.
"""
import numpy as np
import pytest

import pyoracle
import util
from w2dynamics_amd.mctqmc import MCTQMC, W2DynError

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def close_elementwise(a, b):
    """|a - b| <= 1e-10 |b| + 1e-14 max|b| for every element: next to rel(), whose max-norm pins small entries (Legendre
    coefficients of high order are 1e-4 ... 1e-6 of the first one) to 1e-10 of the scale only."""
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all(np.abs(a - b) <= 1e-10 * np.abs(b) + 1e-14 * np.abs(b).max()))


def rel_nan(a, b):
    """rel() for arrays with 0/0 = NaN entries (outer blocks never visited, CTQMC.F90:8757-8758): same NaN pattern required."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    return rel(a[m], b[m]) if m.any() else 0.0


def make_gpu(p, ps, seed, nw, wpb=1, kmax=0):
    q = MCTQMC(nwalkers=nw, walkers_per_bin=wpb, kmax=kmax)
    q.init_rnd(seed); q.init_paras(ps)
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    return q


BASE = dict(Ntau=100, NLegMax=12, Niw=8, MeasGiw=1)      # NLookup_nfft keeps its default (10000): G(iw) is the exact sum whatever it says
TRAJ = [
    ("C1", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=20, TaudiffMax=2.0, PercentageGlobalMove=0.0, PercentageTauShiftMove=0.0), 1234),
    ("C1", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=20, TaudiffMax=-1.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 99),
    ("C3", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 77),
    ("C2", 2001, dict(BASE, Nwarmups=20000, Nmeas=10, NCorr=200, TaudiffMax=10.0, NLegMax=100, Ntau=1000), 5),
    ("C5Z", 501, dict(BASE, Nwarmups=4000, Nmeas=10, NCorr=100, TaudiffMax=10.0, NLegMax=40), 31),
    # two windows of 2/3 beta: ~140 operators inside the window, i.e. the window list lives in LDS (more than 64 entries)
    ("C5Z", 501, dict(BASE, Nwarmups=4000, Nmeas=10, NCorr=100, TaudiffMax=30.0, NLegMax=40, PercentageGlobalMove=0.02, PercentageTauShiftMove=0.02), 32),
    # matrix local-trace path: Kanamori (blocks of 1-3 states) and density-density U with coarse quantum numbers
    ("C4S", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 4242),
    ("C3Q", 501, dict(BASE, Nwarmups=3000, Nmeas=20, NCorr=30, TaudiffMax=5.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 99),
    ("C4", 1001, dict(BASE, Nwarmups=6000, Nmeas=10, NCorr=100, TaudiffMax=10.0, NLegMax=20, Ntau=200), 31),
    # QuantumNumbers = All (automatic block partitioning): Kanamori "Nt All", + inter-orbital hopping, "All" alone, 3 orbitals
    ("C4SA", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 4243),
    ("C4SX", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 4244),
    ("C4SQ", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 4245),
    ("C4A", 1001, dict(BASE, Nwarmups=4000, Nmeas=10, NCorr=100, TaudiffMax=10.0, NLegMax=20, Ntau=200), 32),
    # blocks of 9 and 10 states: outer states propagated in batches
    ("C4N", 501, dict(BASE, Nwarmups=3000, Nmeas=20, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1, MeasSsquare=1), 4251),
    ("C5K", 501, dict(BASE, Nwarmups=3000, Nmeas=10, NCorr=50, TaudiffMax=3.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, MeasSsquare=1), 4252),
    # the 5-orbital Kanamori bench workload thermalised: ~130 operators per chain, i.e. more than 64 outside the window (the
    # evaluator's second group of operators), ~50 inside it (checkpoints), default move mix
    ("C5KB", 5001, dict(BASE, Nwarmups=20000, Nmeas=4, NCorr=100, TaudiffMax=10.0, NLegMax=40, Ntau=200), 4253),
    # user symmetry moves in the global move (NSymMove = 2: an orbital exchange and a cyclic move with a different inverse); the
    # oracle's branch is pinned on reference runs (traj_C3sym / traj_C4Ssym)
    ("C3", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.3, PercentageTauShiftMove=0.05,
                     SymMoves=((1, 0, 3, 2), (1, 2, 3, 0))), 78),
    ("C4S", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.3, PercentageTauShiftMove=0.05,
                      SymMoves=((1, 0, 3, 2), (1, 2, 3, 0))), 4246),
]


@pytest.mark.parametrize("cfg,nftau,over,seed", TRAJ, ids=["C1-addrem", "C1-1window-allmoves", "C3-allmoves", "C2-default", "C5Z-default", "C5Z-wide-window-lds-list",
                              "C4S-kanamori-allmoves", "C3Q-coarse-blocks-allmoves", "C4-kanamori-default",
                              "C4SA-auto-blocks", "C4SX-auto-blocks-hopping", "C4SQ-all-alone", "C4A-auto-blocks-3orb",
                              "C4N-9-state-blocks", "C5K-5orb-kanamori-10-state-blocks", "C5KB-bench-workload-thermalised",
                              "C3-symmetry-moves", "C4S-symmetry-moves"])
def test_chains_retrace_oracle(cfg, nftau, over, seed):
    """Every chain: same configuration, M, sign, window and NPairs after warm-up; same estimators after sampling."""
    _retrace(cfg, nftau, over, seed)


# moves outside the default mix: two pairs at once (Percentage4OperatorMove, StepAdd4 / StepRem4) and pair moves that change the
# outer block (PercentageOuterMove); the oracle's branches are pinned on reference runs (traj_C3four ... traj_C4mix).  The
# outer cases use one window over beta, where such moves are accepted often; C5KB: sequential trace (10-state blocks)
XM_TRAJ = [
    ("C1", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=20, TaudiffMax=2.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, Percentage4OperatorMove=0.3), 61),
    ("C3", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, Percentage4OperatorMove=0.3), 62),
    ("C4S", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, Percentage4OperatorMove=0.3), 63),
    ("C1", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=20, TaudiffMax=10.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, PercentageOuterMove=0.3), 64),
    ("C3", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=25.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, PercentageOuterMove=0.3), 65),
    ("C4S", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=10.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, PercentageOuterMove=0.3), 66),
    ("C4", 1001, dict(BASE, Nwarmups=4000, Nmeas=10, NCorr=100, TaudiffMax=10.0, NLegMax=20, Ntau=200, Percentage4OperatorMove=0.1, PercentageOuterMove=0.1), 67),
    ("C5K", 501, dict(BASE, Nwarmups=2000, Nmeas=6, NCorr=50, TaudiffMax=3.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, Percentage4OperatorMove=0.2, PercentageOuterMove=0.2), 68),
]


@pytest.mark.parametrize("cfg,nftau,over,seed", XM_TRAJ, ids=["C1-two-pair-moves", "C3-two-pair-moves", "C4S-two-pair-moves", "C1-outer-moves", "C3-outer-moves",
                                                             "C4S-outer-moves", "C4-both", "C5K-both-sequential-trace"])
def test_chains_retrace_oracle_with_two_pair_and_outer_moves(cfg, nftau, over, seed):
    _retrace(cfg, nftau, over, seed)


# off-diagonal hybridisation (offdiag = 1): one inverse matrix over all flavours, pairs of unequal flavours, the permutation sign,
# gleg_full / gtau_full; the oracle's branch is pinned on reference runs (traj_C4SO, traj_C3O, traj_C4O)
OD_TRAJ = [
    ("C4SO", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, offdiag=1, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05), 81),
    ("C3O", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, offdiag=1, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05), 82),
    ("C4O", 501, dict(BASE, Nwarmups=3000, Nmeas=20, NCorr=30, TaudiffMax=3.0, offdiag=1, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05), 83),
    ("C4SO", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=-1.0, offdiag=1, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 84),
    # flavourchange_moves = 1: StepFlavourchange_general in place of the two-pair moves (oracle pinned on traj_C4SOf / traj_C4Of)
    ("C4SO", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=3.0, offdiag=1, flavourchange_moves=1, Percentage4OperatorMove=0.3,
                       PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05), 85),
    ("C4O", 501, dict(BASE, Nwarmups=3000, Nmeas=20, NCorr=30, TaudiffMax=10.0, offdiag=1, flavourchange_moves=1, Percentage4OperatorMove=0.3,
                      PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05), 86),
]


@pytest.mark.parametrize("cfg,nftau,over,seed", OD_TRAJ, ids=["C4SO-kanamori-hopping", "C3O-density-density", "C4O-three-orbitals", "C4SO-one-window",
                                                             "C4SO-flavour-change", "C4O-flavour-change"])
def test_chains_retrace_oracle_offdiag(cfg, nftau, over, seed):
    """Every chain: same configuration, full inverse matrix, determinant, sign and NPairs after warm-up; same gleg_full / gtau_full,
    histograms, occupations and density matrix after sampling."""
    p, ps, pset = util.setup(cfg, nftau, over)
    nw = 6
    q = make_gpu(p, ps, seed, nw)
    tdm = pset.get_real("TaudiffMax")
    q.set_taudiffmax(tdm)
    q.ctqmc_warmup(over["Nwarmups"])
    oracles = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", seed, w))
        o.set_taudiffmax(tdm); o.warmup(over["Nwarmups"])
        c = o.get_config(); so = o.state()
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]), "chain %d" % w
        assert osst == c["outer_sst"]
        g = q.get_mc_config(n, w)
        if n:
            assert np.abs(g[0] - c["taus"]).max() < 1e-11
            assert np.array_equal(g[1], c["orbs"]) and np.array_equal(g[2], c["spins"]) and np.array_equal(g[3], c["cas"])
        st = q.get_walker_state(w)
        assert st["det_sign"] == so["det_sign"] and st["window_position"] == so["window_position"]
        assert st["det_log"] == pytest.approx(so["det_log"], abs=1e-8)
        a, b = q.get_minv(w, 0, 0), o.minv_full()
        assert a.shape == b.shape
        if a.size:
            assert rel(a, b) < TOL
        oracles.append(o)
    q.ctqmc_measure(1, 1)
    bins = {k: getattr(q, k + "_bins") for k in ("gleg_full", "gtau_full", "histo", "occ")}
    msign = 0.0; nonzero = 0.0; cm = np.zeros(10)
    for w, o in enumerate(oracles):
        o.measure(); r = o.results(); cm += r["counters_more"]
        for k in bins:      # these problems have a sign problem: a bin is divided by the mean sign of its context, a chain of the oracle by its own
            f_gpu, f_ora = (1.0, 1.0) if k == "histo" else (q.mean_sign, r["mean_sign"])
            assert rel(bins[k][w] * f_gpu, r[k] * f_ora) < TOL, (k, w)
        nonzero = max(nonzero, np.abs(r["gleg_full"][0, 0, 1, 0]).max())
        assert q.get_walker_state(w)["cfgsign"] == o.state()["cfgsign"]
        msign += r["mean_sign"] / nw
        o.close()
    assert nonzero > 1e-3                      # the off-diagonal elements are measured
    if over.get("flavourchange_moves"):
        assert cm[8] > 0 and cm[9] > 0 and q.accflavc == pytest.approx(cm[9] / cm[8], rel=1e-12)
    assert q.mean_sign == pytest.approx(msign, abs=1e-12)
    q.dest_ctqmc(); q.close()


def test_full_size_run_properties():
    """BASELINE.json's bench configuration at full size (C2, 131072 chains, NCorr = 200, Ntau = 1000, NLegMax = 100,
    Nftau = 5001) through size-independent properties: a chain's trajectory depends on (seed, chain number) only - the
    same chains come out of a 64-chain context at the right offset and of the CPU oracle; the pooled results are the
    means of the bins; histograms are normalised; half filling and the sign come out right."""
    over = dict(Nwarmups=2000, Nmeas=3, NCorr=200, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Niw=8, Nftau=5001)
    p, ps, pset = util.setup("C2", 5001, over)
    W = 131072

    def run(nw, first):
        q = MCTQMC(nwalkers=nw, first_walker=first, walkers_per_bin=8)
        q.init_rnd(977); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
        q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"]); q.ctqmc_measure(1, 1)
        return q

    big = run(W, 0)
    gl, glb, hi = big.gleg, big.gleg_bins, big.histo
    assert glb.shape == (W // 8, 1, 2, 100)
    assert rel(gl, glb.mean(axis=0)) < 1e-12 and rel(big.gtau, big.gtau_bins.mean(axis=0)) < 1e-12
    assert np.abs(hi.sum(axis=-1) - 1.0).max() < 1e-12 and big.mean_sign == -1.0
    occ = big.occ
    assert abs(occ[0, 0, 0, 0] - 0.5) < 0.01 and abs(occ[0, 1, 0, 1] - 0.5) < 0.01 and 0.0 < occ[0, 0, 0, 1] < 0.25
    assert np.all(big.gtau[..., 1:-1] > 0.0)
    for first in (0, W - 64):
        small = run(64, first)
        assert rel(small.gleg_bins, glb[first // 8: first // 8 + 8]) < 1e-12
        for w in (0, 63):
            n, osst, _ = small.get_mc_config_scalars(w)
            nb, osb, _ = big.get_mc_config_scalars(first + w)
            assert n == nb and osst == osb and n > 0
            assert np.array_equal(small.get_mc_config(n, w)[0], big.get_mc_config(n, first + w)[0])
        small.dest_ctqmc(); small.close()
    w = W - 1                                          # the last chain of the full-size run against the CPU oracle
    o = pyoracle.OracleWalker(p, pset, ("philox", 977, w))
    o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure()
    c = o.get_config()
    n, osst, _ = big.get_mc_config_scalars(w)
    assert n == len(c["taus"]) and osst == c["outer_sst"] and np.abs(big.get_mc_config(n, w)[0] - c["taus"]).max() < 1e-10
    o.close()
    big.dest_ctqmc(); big.close()


def test_full_size_run_properties_sequential_trace():
    """The 5-orbital Kanamori bench workload (C5K: blocks of up to 10 states, sequential trace with the cached outside product
    and the window checkpoints in global memory) at 16384 chains: a chain's trajectory depends on (seed, chain number) only,
    so the chains at both ends of the full-size context come out of a 64-chain context at the same offset - same
    configurations, same per-bin estimators - whatever else shares the GPU and wherever their checkpoint slots lie."""
    over = dict(Nwarmups=400, Nmeas=2, NCorr=100, TaudiffMax=10.0, Ntau=200, NLegMax=40, Niw=8, Nftau=5001)
    p, ps, pset = util.setup("C5KB", 5001, over)
    W = 16384

    def run(nw, first):
        q = MCTQMC(nwalkers=nw, first_walker=first, walkers_per_bin=8)
        q.init_rnd(4321); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
        q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"]); q.ctqmc_measure(1, 1)
        return q

    big = run(W, 0)
    glb, occb = big.gleg_bins, big.occ_bins
    assert np.abs(big.histo.sum(axis=-1) - 1.0).max() < 1e-12
    for first in (0, W - 64):
        small = run(64, first)
        for w in (0, 31, 63):
            n, osst, _ = small.get_mc_config_scalars(w)
            nb, osb, _ = big.get_mc_config_scalars(first + w)
            assert n == nb and osst == osb and n > 0
            assert np.array_equal(small.get_mc_config(n, w)[0], big.get_mc_config(n, first + w)[0])
        # per-bin results are divided by the mean sign of their own context, and this problem has a (mild) sign problem:
        # about one chain in 10^4 carries the opposite sign (checked against the oracle: tests/zero_sign_probe.py)
        assert rel(small.gleg_bins * small.mean_sign, glb[first // 8: first // 8 + 8] * big.mean_sign) < 1e-12
        assert rel(small.occ_bins * small.mean_sign, occb[first // 8: first // 8 + 8] * big.mean_sign) < 1e-12
        small.dest_ctqmc(); small.close()
    assert -1.0 <= big.mean_sign < -0.99
    big.dest_ctqmc(); big.close()


@pytest.mark.parametrize("in_lds", ["1", "0"])
def test_matrix_path_lds_layouts_retrace_oracle(monkeypatch, in_lds):
    """The matrix path has two LDS layouts (operator matrices staged in LDS at one workgroup per CU, or left in global
    memory at two workgroups per CU; the host picks by size).  Both must retrace the oracle."""
    monkeypatch.setenv("W2DYN_MAT_OPS_IN_LDS", in_lds)
    cfg, nftau, over, seed = [t for t in TRAJ if t[0] == "C4S"][0]
    _retrace(cfg, nftau, over, seed + 1)


def test_chains_retrace_oracle_spin_dependent_bath():
    """Global moves with a bath that depends on the spin: the spin flip maps a flavour onto one with another F(tau), so its inverse matrices are
    built from scratch, while the relabellings between flavours of identical F(tau) (the other test cases: every spin flip) carry them along
    (step_glob, DevConst::fsame); M, det_sign and the estimators of every chain agree with the oracle either way."""
    _retrace("C3", 501, dict(BASE, Nwarmups=3000, Nmeas=30, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.3, PercentageTauShiftMove=0.05), 79,
             patch=lambda p: p.ftau.__setitem__((slice(None), 1, slice(None), 1), 0.7 * p.ftau[:, 1, :, 1]))


def _retrace(cfg, nftau, over, seed, patch=None):
    p, ps, pset = util.setup(cfg, nftau, over)
    if patch is not None:
        p.ftau = np.array(p.ftau, copy=True); patch(p)
    nw = 6
    q = make_gpu(p, ps, seed, nw)
    tdm = pset.get_real("TaudiffMax")
    q.set_taudiffmax(tdm)
    q.ctqmc_warmup(over["Nwarmups"])
    oracles = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", seed, w))
        o.set_taudiffmax(tdm); o.warmup(over["Nwarmups"])
        c = o.get_config(); so = o.state()
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]), "chain %d" % w
        assert osst == c["outer_sst"]
        g = q.get_mc_config(n, w)
        if n:
            assert np.abs(g[0] - c["taus"]).max() < 1e-11
            assert np.array_equal(g[1], c["orbs"]) and np.array_equal(g[2], c["spins"]) and np.array_equal(g[3], c["cas"])
        st = q.get_walker_state(w)
        assert st["det_sign"] == so["det_sign"] and st["window_position"] == so["window_position"] and st["npairs"] == so["npairs"]
        assert st["det_log"] == pytest.approx(so["det_log"], abs=1e-8)
        for orb in range(1, p.nd + 1):
            for sp in (1, 2):
                a, b = q.get_minv(w, orb, sp), o.minv(orb, sp)
                assert a.shape == b.shape
                if a.size:
                    assert rel(a, b) < TOL
        oracles.append(o)
    q.ctqmc_measure(1, 1)
    # a-18: the arrays of MeasOuterHistograms per chain (one chain per bin)
    HIST = ("outersuperstatehisto", "signhistosuperstates", "tracecontribsuperstates", "outerstatehisto", "signhistostates", "tracecontribstates")
    hbins = {k: getattr(q, k + "_bins") for k in HIST}
    bins = {k: getattr(q, k + "_bins") for k in ("gleg", "gtau", "giw", "histo", "occ")}
    tot = {k: 0.0 for k in bins}
    msign = 0.0; dm = 0.0; rho1 = 0.0; ssq = None; cm = np.zeros(10)
    for w, o in enumerate(oracles):
        o.measure(); r = o.results(); cm += r["counters_more"]
        for k in bins:
            assert rel(bins[k][w], r[k]) < TOL, (k, w)
            if k == "gleg":      # Chebyshev moments turned into Legendre coefficients once per run: every order on its own scale
                assert close_elementwise(bins[k][w], r[k]), (k, w)
            tot[k] = tot[k] + r[k] / nw
        msign += r["mean_sign"] / nw
        for k in HIST:      # trace contributions: exp of a log trace of magnitude ~ beta * U, hence 1e-9
            assert rel_nan(hbins[k][w], r[k]) < (1e-9 if k.startswith("tracecontrib") else 1e-12), (k, w)
        assert q.get_walker_state(w)["cfgsign"] == o.state()["cfgsign"]
        # every chain has the same number of measurements, so sign-weighted chain results average like this
        if "densitymatrix" in r:
            dm = dm + r["densitymatrix"] * r["mean_sign"]; rho1 = rho1 + r["rho1"] * r["mean_sign"]
        if "ssquare" in r:       # MeasSsquare: linear in the density matrix, pools like it
            ssq = (0.0 if ssq is None else ssq) + r["ssquare"] * r["mean_sign"]
            sop = r["ssquare_operator"]
        o.close()
    assert q.mean_sign == pytest.approx(msign, abs=1e-12)
    if over.get("Percentage4OperatorMove", 0) > 0:      # AccAdd4 / AccRem4 pooled over the chains (CTQMC.F90:8677-8678)
        assert cm[0] > 0 and cm[2] > 0 and cm[1] + cm[3] > 0
        assert q.accadd4 == pytest.approx(cm[1] / cm[0], rel=1e-12) and q.accrem4 == pytest.approx(cm[3] / cm[2], rel=1e-12)
    if msign == -1.0:
        for k in bins:      # the module-level arrays are the means over chains
            assert rel(getattr(q, k), tot[k]) < TOL, k
    assert rel(q.densitymatrix, dm / nw / msign) < TOL
    assert rel(q.rho1, rho1 / nw / msign) < TOL
    if ssq is not None:
        assert np.abs(ssq).max() > 0.1 and rel(q.ssquare, ssq / nw / msign) < TOL
        assert np.abs(q.ssquare_operator - sop).max() < 1e-12
    assert np.all(np.isnan(q.lhisto)) and np.all(q.rhisto == 0.0)       # what the reference returns (CTQMC.F90:8698-8699)
    assert np.abs(q.eigenstatesenergies - r["eigenstatesenergies"]).max() < 1e-12
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("name", ["C1", "C3", "C2", "C4S", "C3Q", "C4"])
def test_recorded_configuration_vs_reference_golden(name):
    """set_mc_config (full rebuild of M on the device) + one measurement == what the REFERENCE printed."""
    g = util.load_golden("meas_%s.npz" % name)
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, int(g["seed"]), 4, wpb=4)
    q.set_taudiffmax(over["TaudiffMax"])
    c = util.cfg_from(g, "cfg_")
    q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
    q.ctqmc_measure(1, 1)
    for k in ("gleg", "gtau", "giw", "occ", "histo"):
        assert rel(getattr(q, k), g["res_" + k]) < TOL, k
    assert close_elementwise(q.gleg, g["res_gleg"])
    assert q.mean_sign == float(g["res_mean_sign"])
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("cfg,niw,nwarm", [("C2L", 2100, 30000), ("C1", 2000, 3000), ("C5Z", 300, 4000)], ids=["C2L-two-frequency-blocks", "C1-default-Niw", "C5Z-10-flavours"])
def test_giw_production_frequency_counts_vs_oracle(cfg, niw, nwarm):
    """k_meas_giw at the reference's default Niw = 2000 and beyond one frequency block of the kernel (2048 positive
    frequencies per workgroup: Niw = 2100 gives a second, ragged block), with orders beyond one register tile of 24
    annihilators (C2L, beta = 200: k ~ 35) - chain by chain against the oracle's exact sums, element by
    element: |a - b| <= 1e-10 |b| + 1e-13 max|b| (the phases come from three table entries per frequency instead of one
    sincos per element; G(iw) falls off as 1/w, so a max-norm alone would not see the tail)."""
    over = dict(Ntau=100, NLegMax=12, Niw=niw, MeasGiw=1, Nwarmups=nwarm, Nmeas=2, NCorr=50, TaudiffMax=10.0)
    p, ps, pset = util.setup(cfg, 501, over)
    nw = 5
    q = make_gpu(p, ps, 808, nw, wpb=2)
    q.set_taudiffmax(10.0); q.ctqmc_warmup(nwarm); q.ctqmc_measure(1, 1)
    gb = q.giw_bins
    assert gb.shape == (3, p.nd, 2, 2 * niw)
    tot = 0.0; kmaxseen = 0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 808, w))
        o.set_taudiffmax(10.0); o.warmup(nwarm); o.measure()
        r = o.results(); tot = tot + r["giw"] / nw
        kmaxseen = max(kmaxseen, max(o.minv(orb, sp).shape[0] for orb in range(1, p.nd + 1) for sp in (1, 2)))
        o.close()
    a, b = q.giw, tot
    assert np.all(np.abs(a - b) <= 1e-10 * np.abs(b) + 1e-13 * np.abs(b).max())
    assert np.abs(a - np.conj(a[..., ::-1])).max() == 0.0          # G(-iw) = conj G(iw), exactly
    if cfg == "C2L":
        assert kmaxseen > 24
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("atomics", ["", "1"], ids=["factorised", "atomics"])
@pytest.mark.parametrize("name", ["C1", "C3", "C4S"])
def test_two_particle_legendre_and_tau_vs_reference_golden(monkeypatch, name, atomics):
    """MeasGtau4PntLeg (FourPnt = 3): g4tau and g4leg of a recorded configuration == what the REFERENCE printed (g4leg up
    to N4leg: the reference leaves uninitialised memory beyond, see tests/test_oracle_golden.py), and in a sampled run
    the mean over chains == the mean of the oracle chains with the same streams."""
    if atomics:      # g4leg has two kernels: the factorised one (default) and one atomic add per term (fallback for large tables)
        monkeypatch.setenv("W2DYN_G4LEG_ATOMICS", "1")
    g = util.load_golden("g4leg_%s.npz" % name)
    over = g["overrides"]; nl = over["N4leg"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=5, walkers_per_bin=2)
    q.fourpnt = 3
    q.init_rnd(int(g["seed"])); q.init_paras(ps)
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"])
    c = util.cfg_from(g, "cfg_")
    q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
    q.ctqmc_measure(1, 1)
    assert q.g4tau.shape == g["res_g4tau"].shape and rel(q.g4tau, g["res_g4tau"]) < TOL
    assert q.g4leg.shape == g["res_g4leg"].shape
    assert rel(q.g4leg[..., :nl, :nl, :], g["res_g4leg"][..., :nl, :nl, :]) < TOL
    assert np.all(q.g4leg[..., nl:, :, :] == 0.0) and np.all(q.g4leg[..., :, nl:, :] == 0.0)
    q.dest_ctqmc(); q.close()
    # sampled run, chain by chain against the oracle
    over2 = dict(over, Nwarmups=1500, Nmeas=5, NCorr=20)
    p, ps, pset = util.setup(g["config"], g["nftau"], over2)
    nw = 4
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1); q.fourpnt = 3
    q.init_rnd(33); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over2["Nwarmups"]); q.ctqmc_measure(1, 1)
    gt = 0.0; gl = 0.0; ms = 0.0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 33, w), fourpnt=3)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over2["Nwarmups"]); o.measure()
        r = o.results()
        gt = gt + r["g4tau"] * r["mean_sign"] / nw; gl = gl + r["g4leg"] * r["mean_sign"] / nw; ms += r["mean_sign"] / nw
        o.close()
    assert rel(q.g4tau, gt / ms) < TOL and rel(q.g4leg, gl / ms) < TOL
    q.dest_ctqmc(); q.close()


def test_two_particle_legendre_kernels_agree_at_benchmark_size(monkeypatch):
    """g4leg at N4leg = 20, N4iwb = 10 (400 (l, l') pairs: two tiles of the factorised kernel, the <12> instantiation):
    the factorised kernel == the per-term atomics kernel on the same sampled chains."""
    g = util.load_golden("g4leg_C1.npz")
    over = dict(g["overrides"], N4tau=20, N4leg=20, N4iwb=10, Nwarmups=3000, Nmeas=3, NCorr=30)
    res = []
    for atomics in ("", "1"):
        if atomics:
            monkeypatch.setenv("W2DYN_G4LEG_ATOMICS", "1")
        p, ps, pset = util.setup(g["config"], g["nftau"], over)
        q = MCTQMC(nwalkers=6, walkers_per_bin=2); q.fourpnt = 2
        q.init_rnd(71); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
        q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"]); q.ctqmc_measure(1, 1)
        res.append(np.array(q.g4leg)); q.dest_ctqmc(); q.close()
    assert res[0].shape[-3:] == (20, 20, 21) and np.abs(res[0]).max() > 0.0
    assert rel(res[0], res[1]) < TOL


@pytest.mark.parametrize("name", ["C1", "C3", "C4S"])
def test_two_particle_matsubara_vs_reference_golden(name):
    """Z-space g2iw / g4iw / g4iw_pp (FourPnt = 4) of a recorded configuration == what the REFERENCE printed."""
    g = util.load_golden("g4_%s.npz" % name)
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=5, walkers_per_bin=2)
    q.fourpnt = 4
    q.init_rnd(int(g["seed"])); q.init_paras(ps)
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"])
    c = util.cfg_from(g, "cfg_")
    q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
    q.ctqmc_measure(1, 1)
    for k in ("g2iw", "g4iw", "g4iw_pp"):
        if "res_" + k in g:
            assert rel(getattr(q, k), g["res_" + k]) < TOL, k
            assert rel(getattr(q, k + "_bins").mean(axis=0), g["res_" + k]) < TOL, k      # every chain holds the same configuration
    assert rel(q.gleg, g["res_gleg"]) < TOL
    q.dest_ctqmc(); q.close()


def test_two_particle_matsubara_chains_vs_oracle():
    """Sampling run with FourPnt = 4: the mean over chains of g4iw equals the mean of the oracle chains."""
    over = dict(BASE, Nwarmups=2000, Nmeas=5, NCorr=20, TaudiffMax=4.0, N4iwf=3, N4iwb=2, MeasG2iw=1, MeasG4iwPP=1,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C3", 501, over)
    nw = 6
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.fourpnt = 4
    q.init_rnd(11); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(4.0); q.ctqmc_warmup(2000); q.ctqmc_measure(1, 1)
    tot = {"g4iw": 0.0, "g4iw_pp": 0.0, "g2iw": 0.0}
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 11, w), fourpnt=4)
        o.set_taudiffmax(4.0); o.warmup(2000); o.measure(); r = o.results()
        for k in tot:
            tot[k] = tot[k] + r[k] / nw
            assert rel(getattr(q, k + "_bins")[w], r[k]) < TOL, (k, w)
        o.close()
    for k in tot:
        assert rel(getattr(q, k), tot[k]) < TOL, k
    q.dest_ctqmc(); q.close()


def test_two_particle_matsubara_production_box_and_folded_bins(monkeypatch):
    """a-14 at the frequency box of BASELINE.json's two-particle configuration (N4iwf = 10, N4iwb = 5) with the accumulator
    budget cut so that the 6 sample bins fold onto 3 accumulator bins (chains b and b + 3 share one): every accumulator
    bin equals the mean of the oracle chains it holds, the pooled result the mean of all."""
    monkeypatch.setenv("W2DYN_ACC_BYTES", str(3 * (2 * 2 * 20 * 20 * 11 * 2 + 2 * 30 * 30) * 16 + 1000))
    over = dict(BASE, Nwarmups=2000, Nmeas=4, NCorr=20, TaudiffMax=2.0, N4iwf=10, N4iwb=5, MeasG2iw=1, MeasG4iwPP=1,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C1", 501, over)
    nw = 6
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.fourpnt = 4
    q.init_rnd(13); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    assert int(q.scalar("nb4")) == 3 and int(q.scalar("nbins")) == 6
    q.set_taudiffmax(2.0); q.ctqmc_warmup(2000); q.ctqmc_measure(1, 1)
    per = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 13, w), fourpnt=4)
        o.set_taudiffmax(2.0); o.warmup(2000); o.measure(); per.append(o.results()); o.close()
    for k in ("g4iw", "g4iw_pp", "g2iw"):
        b = getattr(q, k + "_bins")
        assert b.shape[0] == 3 and b.shape[-1] == (30 if k == "g2iw" else 11)
        for a in range(3):
            assert rel(b[a], 0.5 * (per[a][k] + per[a + 3][k])) < TOL, (k, a)
        assert rel(getattr(q, k), sum(r[k] for r in per) / nw) < TOL, k
    q.dest_ctqmc(); q.close()


def test_two_particle_matsubara_other_convention_ragged_box():
    """a-14 with ZPHConvention = 1 on a box that is ragged for every kernel of the assembly (N4iwf = 9: 18 fermionic frequencies =
    one full and one partial 16 x 16 tile; N4iwb = 4: 9 bosonic frequencies = two chunks of the exchange kernel), two orbitals
    (equal- and different-flavour pairs), chain by chain against the oracle."""
    over = dict(BASE, Nwarmups=2000, Nmeas=3, NCorr=20, TaudiffMax=4.0, N4iwf=9, N4iwb=4, MeasG2iw=1, ZPHConvention=1,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C3", 501, over)
    nw = 4
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.fourpnt = 4
    q.init_rnd(17); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(4.0); q.ctqmc_warmup(2000); q.ctqmc_measure(1, 1)
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 17, w), fourpnt=4)
        o.set_taudiffmax(4.0); o.warmup(2000); o.measure(); r = o.results(); o.close()
        for k in ("g4iw", "g2iw"):
            assert getattr(q, k + "_bins")[w].shape == r[k].shape and rel(getattr(q, k + "_bins")[w], r[k]) < TOL, (k, w)
    q.dest_ctqmc(); q.close()


def test_worm_estimator_folded_bins(monkeypatch):
    """a-16 with folded accumulator bins (nbw < nbins) at N4iwf = 10, N4iwb = 5 (three passes of k_meas_g4w over the
    frequency box): the pooled g4iw_worm follows from the oracle chains as in the unfolded case."""
    g = util.load_golden("worm_C5a.npz")
    monkeypatch.setenv("W2DYN_ACC_BYTES", str(2 * 20 * 20 * 11 * 16 + 100))
    comp = int(g["component"]); nw = 4
    q, p, pset, over = _worm_gpu(g, nw, 1, 22)
    assert int(q.scalar("nbw")) == 2
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_worm_warmup(0, 4, comp)
    q.ctqmc_measure(4, comp)
    pooled = 0.0; cw = cz = mw = 0.0; binsum = [0.0, 0.0]; bcw = [0.0, 0.0]; bcz = [0.0, 0.0]; bmw = [0.0, 0.0]
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 22, w), fourpnt=8)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure_sector(4, comp)
        r = o.results(); oc = o.worm_counters()
        raw = r["g4iw_worm"] * (oc["cnt_meas_w"] * r["counters"][11] / max(oc["cnt_sampling_w"], 1.0))     # undo the chain's own scaling
        pooled = pooled + raw; binsum[w % 2] = binsum[w % 2] + raw
        cw += oc["cnt_sampling_w"]; cz += r["counters"][11]; mw += oc["cnt_meas_w"]
        bcw[w % 2] += oc["cnt_sampling_w"]; bcz[w % 2] += r["counters"][11]; bmw[w % 2] += oc["cnt_meas_w"]
        o.close()
    assert mw > 0
    assert rel(q.g4iw_worm, pooled * cw / (mw * cz)) < TOL
    bins = q.g4iw_worm_bins
    assert bins.shape[0] == 2
    for b in range(2):
        assert rel(bins[b], binsum[b] * bcw[b] / (bmw[b] * bcz[b])) < TOL, b
    q.dest_ctqmc(); q.close()


def test_random_streams_continue_across_solves():
    """The reference seeds its generator once (impurity.py:387) and the state survives init_solver / dest_ctqmc: a second
    solve of the same context goes on in every chain's stream instead of replaying it (iteration 2 of a DMFT loop must
    not repeat iteration 1).  Checked against oracle chains whose second solve shares the first one's generator."""
    over = dict(BASE, Nwarmups=1500, Nmeas=10, NCorr=20, TaudiffMax=2.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C1", 501, over)
    nw = 4
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.init_rnd(41); q.init_paras(ps)
    first, second = [], []
    for it in range(2):
        q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, it + 1)
        q.set_taudiffmax(2.0); q.ctqmc_warmup(1500); q.ctqmc_measure(1, 1)
        (first if it == 0 else second).append((np.array(q.gleg_bins), [q.get_walker_state(w)["ndraw"] for w in range(nw)]))
        q.dest_ctqmc()
    assert not np.array_equal(first[0][0], second[0][0])
    assert all(b > a for a, b in zip(first[0][1], second[0][1]))
    for w in range(nw):
        o1 = pyoracle.OracleWalker(p, pset, ("philox", 41, w))
        o1.set_taudiffmax(2.0); o1.warmup(1500); o1.measure()
        assert rel(first[0][0][w], o1.results()["gleg"]) < TOL
        o2 = pyoracle.OracleWalker(p, pset, ("shared", o1))
        o2.set_taudiffmax(2.0); o2.warmup(1500); o2.measure()
        assert rel(second[0][0][w], o2.results()["gleg"]) < TOL, w
        o2.close(); o1.close()
    q.init_rnd(41)                                   # a new seed call starts the streams afresh
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(2.0); q.ctqmc_warmup(1500); q.ctqmc_measure(1, 1)
    assert np.array_equal(np.array(q.gleg_bins), first[0][0])
    q.dest_ctqmc(); q.close()


def test_soft_abort_by_signal():
    """SIGUSR1 during ctqmc_measure (trapped from init_solver on, CTQMC.F90:8076-8078): the run stops between two launches,
    what was measured is normalised, aborted = 1 (CTQMC.F90:8648-8658); the handlers are gone afterwards."""
    import os, signal, threading, time
    prev = signal.signal(signal.SIGUSR1, lambda *a: None)       # a stray late signal must not kill the test process
    try:
        over = dict(BASE, Nwarmups=1000, Nmeas=2000000, NCorr=50, TaudiffMax=2.0)
        p, ps, pset = util.setup("C1", 501, over)
        q = MCTQMC(nwalkers=256, walkers_per_bin=8)
        q.init_rnd(3); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
        q.set_taudiffmax(2.0); q.ctqmc_warmup(1000)
        t = threading.Timer(0.5, lambda: os.kill(os.getpid(), signal.SIGUSR1))
        t0 = time.time(); t.start()
        q.ctqmc_measure(1, 1)
        t.join()
        assert time.time() - t0 < 30.0
        done = int(q.scalar("meas_done"))
        assert q.aborted == 1 and 0 < done < over["Nmeas"]
        assert q.mean_sign == -1.0 and np.abs(q.histo.sum(axis=-1) - 1.0).max() < 1e-12 and np.all(np.isfinite(q.gleg))
        assert abs(q.occ[0, 0, 0, 0] - 0.5) < 0.05
        q.dest_ctqmc()
        # a complete run afterwards reports aborted = 0
        _, ps2, _ = util.setup("C1", 501, dict(over, Nmeas=5))
        q.init_paras(ps2); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 2)
        q.set_taudiffmax(2.0); q.ctqmc_warmup(100); q.ctqmc_measure(1, 1)
        assert q.aborted == 0 and int(q.scalar("meas_done")) == 5
        q.dest_ctqmc(); q.close()
    finally:
        signal.signal(signal.SIGUSR1, prev)


def test_ragged_and_empty_chains():
    """Empty configurations (k=0 everywhere), chains with very different orders in one launch, walker count not a
    multiple of the block size."""
    p, ps, pset = util.setup("C1", 501, dict(BASE, Nwarmups=0, Nmeas=1, NCorr=0, TaudiffMax=2.0))
    q = make_gpu(p, ps, 1, 7, wpb=1)
    q.set_taudiffmax(2.0)
    q.ctqmc_measure(1, 1)                       # all empty
    assert np.all(q.gleg == 0.0) and np.all(q.histo[:, :, 0] == 1.0)
    occ_e = q.occ_bins
    for w in range(7):
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == 0
        up, dn = osst & 1, (osst >> 1) & 1
        assert occ_e[w, 0, 0, 0, 0] == up and occ_e[w, 0, 1, 0, 1] == dn
    # one chain gets a long configuration, the others stay empty
    k = 20
    taus = np.sort(np.random.default_rng(0).uniform(0.01, p.beta - 0.01, 2 * k))
    cas = np.array([1, 2] * k, np.int32)
    q.set_mc_config(2 * k, taus, np.ones(2 * k, np.int32), np.ones(2 * k, np.int32), cas, -np.ones(2 * k, np.int32), 0, 1, walker=3)
    o = pyoracle.OracleWalker(p, pset, ("philox", 1, 3))
    o.set_taudiffmax(2.0)
    o.set_config(dict(taus=taus, orbs=np.ones(2 * k, np.int32), spins=np.ones(2 * k, np.int32), cas=cas, outer_sst=0))
    assert rel(q.get_minv(3, 1, 1), o.minv(1, 1)) < TOL
    q.ctqmc_measure(1, 1)
    o.measure()
    assert rel(q.gleg_bins[3], o.results()["gleg"]) < TOL
    assert np.all(q.gleg_bins[2] == 0.0)
    o.close(); q.dest_ctqmc(); q.close()


def test_capacity_grows_instead_of_overflowing():
    """The reference's matrices grow with the expansion order (MatrixUpdate.F90:410-442 reallocates); here a chain that
    fills its kmax slots stops, the host enlarges the arrays and the chain goes on: with kmax = 2 (C1 needs ~8) warm-up,
    calibration-free run and estimators retrace the oracle exactly as with the default capacity."""
    over = dict(BASE, Nwarmups=3000, Nmeas=20, NCorr=20, TaudiffMax=-1.0, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C1", 501, over)
    nw = 8
    q = make_gpu(p, ps, 3, nw, wpb=1, kmax=2)
    assert int(q.scalar("kmax")) == 2
    q.set_taudiffmax(-1.0)
    q.ctqmc_warmup(3000)
    assert int(q.scalar("kmax")) > 2
    q.ctqmc_measure(1, 1)
    kmax_seen = 0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 3, w))
        o.set_taudiffmax(-1.0); o.warmup(3000); o.measure()
        r = o.results(); c = o.get_config()
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]) and osst == c["outer_sst"]
        if n:
            assert np.abs(q.get_mc_config(n, w)[0] - c["taus"]).max() < 1e-11
        for k in ("gleg", "gtau", "occ", "histo"):
            assert rel(getattr(q, k + "_bins")[w], r[k]) < TOL, (k, w)
        for sp in (1, 2):
            a, b = q.get_minv(w, 1, sp), o.minv(1, sp)
            assert a.shape == b.shape and (a.size == 0 or rel(a, b) < TOL)
        kmax_seen = max(kmax_seen, q.get_minv(w, 1, 1).shape[0], q.get_minv(w, 1, 2).shape[0])
        o.close()
    assert kmax_seen > 2
    q.dest_ctqmc(); q.close()


def test_capacity_overflow_inside_measurement_run_is_reported():
    """Chains that stop at capacity in the middle of the asynchronous measurement loop cannot be repaired afterwards (their
    measurements would be taken at the wrong steps): the run fails loudly with W2DYN_EOVERFLOW."""
    p, ps, pset = util.setup("C1", 501, dict(BASE, Nwarmups=0, Nmeas=64, NCorr=100, TaudiffMax=-1.0))
    q = make_gpu(p, ps, 3, 8, kmax=2)
    q.set_taudiffmax(-1.0)
    with pytest.raises(W2DynError) as e:
        q.ctqmc_measure(1, 1)          # no warm-up: the chains grow from the empty configuration inside the run
    assert e.value.code == -4
    q.close()


def test_unsupported_modes_are_refused():
    p, _, _ = util.setup("C1", 501, {})
    for over in (dict(segment=1), dict(offdiag=1, FourPnt=0, MeasSusz=1), dict(statesampling=1), dict(flavourchange_moves=1), dict(offdiag=1, Percentage4OperatorMove=0.1),
                 dict(EnableOuterTruncation="YES")):
        _, ps, _ = util.setup("C1", 501, over)
        q = MCTQMC(nwalkers=4)
        q.init_rnd(1); q.init_paras(ps)
        with pytest.raises(W2DynError) as e:
            q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
        assert e.value.code == -2, over
        q.close()
    # the window calibration with the moves outside the default mix is refused (worm sampling with them runs on the tree-product trace path)
    _, ps, _ = util.setup("C1", 501, dict(Percentage4OperatorMove=0.1, WormMeasGiw=1, PercentageWormInsert=0.2, PercentageWormReplace=0.2))
    q = MCTQMC(nwalkers=4); q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    with pytest.raises(W2DynError) as e:
        q.ctqmc_calibrate(True, 10, 10, 10, 100)
    assert e.value.code == -2
    q.close()
    # a component whose U-matrix slices vanish would never let the worm in
    _, ps, _ = util.setup("C1", 501, dict(PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormMeasUcaca=1))
    q = MCTQMC(nwalkers=4); q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    for sec, comp in ((15, 2), (14, 0)):      # (15, 2): u1 = u3, the antisymmetrised U(u1, u3, ., .) vanishes; (14, 0): these sectors sample one component
        with pytest.raises(W2DynError) as e:
            q.ctqmc_measure(sec, comp)
        assert e.value.code == -1, (sec, comp)
    q.close()
    # worm sampling together with the two-pair / outer-block moves: tree-product trace path only (blocks of up to four states)
    p9, ps9, _ = util.setup("C4N", 501, dict(Percentage4OperatorMove=0.1, WormMeasGiw=1, PercentageWormInsert=0.2, PercentageWormReplace=0.2))
    q = MCTQMC(nwalkers=4); q.init_rnd(1); q.init_paras(ps9); q.init_solver(p9.umatrix, p9.ftau, p9.muimp, p9.screening, 1)
    with pytest.raises(W2DynError) as e:
        q.ctqmc_measure(2, 1)
    assert e.value.code == -2
    q.close()
    # blocks of more than 8 states are refused: 3-orbital Kanamori with the particle number as the only quantum number
    from w2dynamics_amd import params, synth
    p2 = synth.config("C4", 501)
    ps = params.make_pstring({}, beta=p2.beta, Nd=p2.nd, Hamiltonian="Kanamori", QuantumNumbers="Nt")
    q = MCTQMC(nwalkers=4); q.init_rnd(1); q.init_paras(ps)
    with pytest.raises(W2DynError) as e:
        q.init_solver(p2.umatrix, p2.ftau, p2.muimp, p2.screening, 1)
    assert e.value.code == -2
    q.close()


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4", "C5Z"])
def test_full_run_statistics_vs_reference(name):
    """Full-run observables agree with long REFERENCE runs (8 seeds, tests/golden/stat_*.npz) within error bars.

    chi^2-style check: |HIP - ref| < 5 sigma_combined element-wise for occ and the leading Legendre coefficients,
    and the reduced chi^2 over those elements < 2."""
    g = util.load_golden("stat_%s.npz" % name)
    over = dict(g["overrides"])
    nwalk = 512
    over.update(Nmeas=max(50, int(over["Nmeas"]) * 8 // nwalk * 2))
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, 2024, nwalk, wpb=16)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"] // 4)
    q.ctqmc_measure(1, 1)
    chi = []
    for key in ("occ", "gleg"):
        b = getattr(q, key + "_bins"); nb = b.shape[0]
        mean = b.mean(axis=0); err = b.std(axis=0, ddof=1) / np.sqrt(nb)
        ref, referr = g[key + "_mean"], g[key + "_err"]
        if key == "gleg":
            mean, err, ref, referr = mean[..., :10:2], err[..., :10:2], ref[..., :10:2], referr[..., :10:2]
        sig = np.sqrt(err ** 2 + referr ** 2)
        mask = sig > 0
        z = (mean[mask] - ref[mask]) / sig[mask]
        assert np.abs(z).max() < 5.0, (key, np.abs(z).max())
        chi.append((z ** 2).mean())
    assert np.mean(chi) < 2.0, chi
    assert abs(q.accadd[0] - float(g["accadd_mean"][0])) < 5 * float(g["accadd_err"][0]) + 0.003
    q.dest_ctqmc(); q.close()


def test_offdiag_statistics_vs_reference():
    """offdiag = 1: gleg_full (diagonal and off-diagonal flavour pairs), occupations and the mean sign of a 512-chain run agree with
    eight seeded REFERENCE runs (tests/golden/stat_C4SO.npz) within error bars; this problem has a sign problem (mean sign -0.68)."""
    g = util.load_golden("stat_C4SO.npz")
    over = dict(g["overrides"])
    nwalk = 512
    over.update(Nmeas=int(over["Nmeas"]) * 8 // nwalk * 2)
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, 2025, nwalk, wpb=16)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"] // 4)
    q.ctqmc_measure(1, 1)
    assert abs(q.mean_sign - float(g["mean_sign_mean"])) < 5 * float(g["mean_sign_err"]) + 0.01
    chi = []
    for key in ("occ", "gleg_full"):
        b = getattr(q, key + "_bins"); nb = b.shape[0]
        mean = b.mean(axis=0); err = b.std(axis=0, ddof=1) / np.sqrt(nb)
        ref, referr = g[key + "_mean"], g[key + "_err"]
        if key == "gleg_full":
            mean, err, ref, referr = mean[..., :8], err[..., :8], ref[..., :8], referr[..., :8]
            assert np.abs(ref[0, 0, 1, 0]).max() > 0.1            # the off-diagonal Green's function is there
        sig = np.sqrt(err ** 2 + referr ** 2)
        mask = sig > 0
        z = (mean[mask] - ref[mask]) / sig[mask]
        assert np.abs(z).max() < 5.0, (key, np.abs(z).max())
        chi.append((z ** 2).mean())
    assert np.mean(chi) < 2.0, chi
    assert abs(q.accadd[0] - float(g["accadd_mean"][0])) < 5 * float(g["accadd_err"][0]) + 0.003
    q.dest_ctqmc(); q.close()


# C4Sa: matrix trace path (2-orbital Kanamori); C5a: BASELINE.json configs[4] (5 orbitals, beta = 100, N4iwf = 10, N4iwb = 5,
# component 12 = flavours 0011: lane-split level integrals over 10 flavours, window lists of more than 64 operators in LDS)
WORM_CASES = ["C1a", "C1b", "C3a", "C4Sa", "C5a"]
WORM2_CASES = ["C1a", "C1b", "C3a", "C4Sa"]       # sector 2 (one-particle worm)


def _worm_gpu(g, nw, wpb, seed):
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=nw, walkers_per_bin=wpb)
    q.fourpnt = 8
    q.init_rnd(seed); q.init_paras(ps)
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"])
    return q, p, pset, over


@pytest.mark.parametrize("name", WORM_CASES)
def test_worm_sampling_chains_vs_oracle(name):
    """Worm sampling in sector 4 (a-15/a-16): every chain retraces the oracle chain with the same stream through the
    eta-search step count (count_steps restores the configuration, not the stream), the mixed Z / worm loop and the
    G4iw worm estimator; the pooled result follows from the per-chain ones."""
    g = util.load_golden("worm_%s.npz" % name)
    comp = int(g["component"]); nw = 5
    q, p, pset, over = _worm_gpu(g, nw, 1, 21)
    assert q.wormeta[2] == over["WormEta"]
    q.ctqmc_warmup(over["Nwarmups"])
    sw, sz = q.count_steps(4, comp, 3000)
    assert sw + sz == 3000
    chains_w = q.steps_worm_chains
    q.ctqmc_worm_warmup(0, 4, comp)
    q.ctqmc_measure(4, comp)
    wc = q.worm_counters; bins = q.g4iw_worm_bins
    keys = ["g4iw_worm"] + (["g4iwpp_worm"] if over.get("G4pp") else [])
    pooled = {k: 0.0 for k in keys}; cw = cz = mw = 0.0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 21, w), fourpnt=8)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
        assert o.count_steps(4, comp, 3000)[0] == chains_w[w], w
        o.measure_sector(4, comp)
        r = o.results(); oc = o.worm_counters(); c = o.get_config()
        got = [r["counters"][11], oc["cnt_sampling_w"], oc["try_wadd"], oc["acc_wadd"], oc["try_wrem"], oc["acc_wrem"], oc["try_wrep"], oc["acc_wrep"]]
        assert np.array_equal(wc[w], got), (w, wc[w], got)
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]) and osst == c["outer_sst"]
        taus, orbs, spins, cas, hyb, _, _ = q.get_mc_config(n, w)
        if n:
            assert np.abs(taus - c["taus"]).max() < 1e-12 and np.array_equal(hyb, c["hashybs"]) and np.array_equal(cas, c["cas"])
            assert np.array_equal(orbs, c["orbs"]) and np.array_equal(spins, c["spins"])
        for k in keys:
            assert rel(getattr(q, k + "_bins")[w], r[k]) < TOL, (k, w)
            pooled[k] = pooled[k] + r[k] * (oc["cnt_meas_w"] * r["counters"][11] / max(oc["cnt_sampling_w"], 1.0))
        cw += oc["cnt_sampling_w"]; cz += r["counters"][11]; mw += oc["cnt_meas_w"]
        assert rel(q.histo_bins[w], r["histo"]) < TOL and rel(q.occ_bins[w], r["occ"]) < TOL
        o.close()
    assert mw > 0 and bins.shape[0] == nw
    for k in keys:
        assert rel(getattr(q, k), pooled[k] * cw / (mw * cz)) < TOL, k
    assert q.cntsampling[0] == cz and q.cntsampling[3] == cw
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("name", WORM_CASES)
def test_worm_sampling_vs_reference_statistics(name):
    """The REFERENCE's g4iw_worm of one short chain (worm_*.npz, mt19937) is one draw from the distribution of the
    single-chain bins of a GPU run with the same parameters: elementwise within 6 sigma of the GPU bin scatter around
    the GPU mean (the streams differ, so this is a statistical, not a bit-wise, comparison)."""
    g = util.load_golden("worm_%s.npz" % name)
    comp = int(g["component"]); nw = 512
    q, p, pset, over = _worm_gpu(g, nw, 1, 5)
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(4, comp)
    for k in ["g4iw_worm"] + (["g4iwpp_worm"] if over.get("G4pp") else []):
        b = getattr(q, k + "_bins"); m = b.mean(axis=0)
        sig = np.sqrt(b.real.var(axis=0) + b.imag.var(axis=0))
        ref = g["res_" + k]
        assert np.all(np.abs(ref - m) < 6.0 * sig + 1e-12), (k, np.abs(ref - m).max(), sig.max())
        # the pooled estimate is close to the plain mean of the bins
        assert np.abs(getattr(q, k) - m).max() < 0.3 * sig.max()
    assert abs(q.accwormadd[2] - float(g["res_accwormadd"][2])) < 0.05
    assert abs(q.accwormrep[2] - float(g["res_accwormrep"][2])) < 0.1
    q.dest_ctqmc(); q.close()


def test_worm_sampling_vs_reference_seeds():
    """BASELINE.json configs[4] (5 orbitals, beta = 100, sector 4, component 0011, N4iwf = 10, N4iwb = 5): the worm estimator
    of a GPU run against mean +- standard error over 16 seeded REFERENCE chains (tests/golden/wormstat_C5a.npz, 2000
    measurements each): element by element within 5 sigma of the combined error, reduced chi^2 below 2 over the elements
    with a signal; the Z-space occupations sampled along with it likewise."""
    g = util.load_golden("wormstat_C5a.npz")
    # chains of the reference's own length (2000 measurements after the same warm-up): a chain starts in Z space and moves
    # between the two spaces every few thousand steps, so the ratio estimator N_worm / N_Z of a much shorter chain would carry a
    # start-up bias the reference's chains do not have
    comp = int(g["component"]); nw = 128
    over = dict(g["overrides"])
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=nw, walkers_per_bin=4)
    q.fourpnt = 8
    q.init_rnd(6); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(4, comp)
    b = q.g4iw_worm_bins; nb = b.shape[0]
    mean = q.g4iw_worm
    err = np.sqrt(b.real.var(axis=0, ddof=1) + b.imag.var(axis=0, ddof=1)) / np.sqrt(nb)
    ref, referr = g["g4iw_worm_mean"], g["g4iw_worm_err"]
    sig = np.sqrt(err ** 2 + referr ** 2)
    z = np.abs(mean - ref) / sig
    assert z.max() < 5.0, z.max()
    strong = np.abs(ref) > 5 * referr                   # elements where the reference sees a signal at all
    assert strong.sum() > 20
    assert (z[strong] ** 2).mean() / 2.0 < 2.0, (z[strong] ** 2).mean()       # complex: two degrees of freedom per element
    assert np.abs(mean[strong] - ref[strong]).max() < 0.5 * np.abs(ref).max()
    ob = q.occ_bins
    zo = np.abs(ob.mean(axis=0) - g["occ_mean"]) / np.sqrt(ob.var(axis=0, ddof=1) / nb + g["occ_err"] ** 2 + 1e-30)
    assert zo.max() < 5.0, zo.max()
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("name", WORM2_CASES)
def test_g_worm_sampling_chains_vs_oracle(name):
    """Worm sampling in sector 2 (one-particle Green's function, two worm operators): every chain retraces the oracle
    chain with the same stream through count_steps, the mixed Z / worm loop and the giw_worm / gtau_worm estimators;
    the pooled result follows from the per-chain ones."""
    g = util.load_golden("worm2_%s.npz" % name)
    comp = int(g["component"]); nw = 5
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, 23, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    assert q.wormeta[0] == over["WormEta"]
    q.ctqmc_warmup(over["Nwarmups"])
    sw, sz = q.count_steps(2, comp, 3000)
    assert sw + sz == 3000
    chains_w = q.steps_worm_chains
    q.ctqmc_worm_warmup(0, 2, comp)
    q.ctqmc_measure(2, comp)
    wc = q.worm_counters
    keys = ["giw_worm"] + (["gtau_worm"] if over.get("WormMeasGtau") else [])
    pooled = {k: 0.0 for k in keys}; cw = cz = mw = 0.0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 23, w))
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
        assert o.count_steps(2, comp, 3000)[0] == chains_w[w], w
        o.measure_sector(2, comp)
        r = o.results(); oc = o.worm_counters(); c = o.get_config()
        got = [r["counters"][11], oc["cnt_sampling_w"], oc["try_wadd"], oc["acc_wadd"], oc["try_wrem"], oc["acc_wrem"], oc["try_wrep"], oc["acc_wrep"]]
        assert np.array_equal(wc[w], got), (w, wc[w], got)
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]) and osst == c["outer_sst"]
        taus, orbs, spins, cas, hyb, _, _ = q.get_mc_config(n, w)
        if n:
            assert np.abs(taus - c["taus"]).max() < 1e-12 and np.array_equal(hyb, c["hashybs"]) and np.array_equal(cas, c["cas"])
            assert np.array_equal(orbs, c["orbs"]) and np.array_equal(spins, c["spins"])
        for k in keys:
            assert rel(getattr(q, k + "_bins")[w], r[k]) < TOL, (k, w)
            pooled[k] = pooled[k] + r[k] * (oc["cnt_meas_w"] * r["counters"][11] / max(oc["cnt_sampling_w"], 1.0))
        cw += oc["cnt_sampling_w"]; cz += r["counters"][11]; mw += oc["cnt_meas_w"]
        assert rel(q.histo_bins[w], r["histo"]) < TOL and rel(q.occ_bins[w], r["occ"]) < TOL
        o.close()
    assert mw > 0
    for k in keys:
        assert rel(getattr(q, k), pooled[k] * cw / (mw * cz)) < TOL, k
    assert q.cntsampling[0] == cz and q.cntsampling[1] == cw
    q.dest_ctqmc(); q.close()


WORMEQ_CASES = ["P2_C1a", "P2_C1b", "P2_C3a", "P2pp_C1a", "P2pp_C3a", "P3_C1a", "P3_C3a", "P3pp_C1a", "P3pp_C3a",
                "P2_C4Sa", "P3_C4Sa", "P2pp_C4Sa", "P3pp_C4Sa", "P2_C4Na", "P3pp_C4Na",
                "GS_C1a", "GS_C3a", "GS_C4Sa", "GS_C4Na"]      # GS: sector 3 (GSigma): three-operator object, random flavours, U-matrix factor
# sectors 5 and 10 - 16: worms of four, six, eight and twelve operators with U-matrix factors (those with more than four: matrix trace path, also for C1 / C3)
WORMEQ_CASES += ["S%d_%s" % (sec, cfg) for cfg in ("C1", "C3", "C4S") for sec in (5, 10, 11, 12, 13, 14, 15, 16) if (sec, cfg) != (11, "C3")]
WORMEQ_CASES += ["S%d_C4N" % sec for sec in (5, 10, 14, 15, 16)]
# worm sampling together with Percentage4OperatorMove / PercentageOuterMove (matrix trace path, also for C1 / C3; sectors 2, 3, 4, 6)
WORMEQ_CASES += ["X4_C4S", "X2_C4S", "X3_C4S", "X4_C3", "X2_C1"]        # blocks of more than four states (X6_C4N, oracle only): refused


@pytest.mark.parametrize("name", WORMEQ_CASES)
def test_equal_time_worm_sampling_chains_vs_oracle(name):
    """Worm sampling with equal-time pairs, sectors 6 - 9 (P2 / P2pp / P3 / P3pp), scalar (C1, C3) and matrix trace path (C4S):
    every chain retraces the oracle chain with the same stream through count_steps, the mixed Z / worm loop (insertion with
    the pair offsets, replacement that moves the partner along) and the p2iw / p2tau / p3iw estimators; the pooled result
    follows from the per-chain ones.  The oracle itself retraces the reference's runs (tests/test_oracle_golden.py)."""
    g = util.load_golden("wormeq_%s.npz" % name)
    comp = int(g["component"]); sec = int(g["sector"]); nw = 5
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    fourpnt = 8 if sec in (4, 5, 11) else 0
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1); q.fourpnt = fourpnt
    q.init_rnd(29); q.init_paras(ps)
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    # one-state blocks without the extra moves: closed-form local trace, for worms of up to twelve operators; otherwise the operator list
    closed = str(g["config"]) in ("C1", "C3") and not name.startswith("X")
    assert (int(q.scalar("trace_path")) == 0) == closed, (name, q.scalar("trace_path"))
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"])
    sw, sz = q.count_steps(sec, comp, 3000)
    assert sw + sz == 3000
    chains_w = q.steps_worm_chains
    q.ctqmc_worm_warmup(0, sec, comp)
    q.ctqmc_measure(sec, comp)
    wc = q.worm_counters
    keys = [k for k in pyoracle.WORM_EQ_NAMES + pyoracle.WORM_MORE_NAMES + ("gsigmaiw_worm", "g4iw_worm", "giw_worm", "gtau_worm") if "res_" + k in g and g["res_" + k].size > 1]
    assert keys
    pooled = {k: 0.0 for k in keys}; cw = cz = mw = 0.0; nrep = 0.0; n4 = 0.0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 29, w), fourpnt=fourpnt)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
        assert o.count_steps(sec, comp, 3000)[0] == chains_w[w], w
        o.measure_sector(sec, comp)
        r = o.results(); oc = o.worm_counters(); c = o.get_config()
        got = [r["counters"][11], oc["cnt_sampling_w"], oc["try_wadd"], oc["acc_wadd"], oc["try_wrem"], oc["acc_wrem"], oc["try_wrep"], oc["acc_wrep"]]
        assert np.array_equal(wc[w], got), (w, wc[w], got)
        nrep += oc["acc_wrep"]; n4 += float(r["counters_more"][[1, 3, 5, 7]].sum())
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]) and osst == c["outer_sst"]
        taus, orbs, spins, cas, hyb, _, _ = q.get_mc_config(n, w)
        if n:
            assert np.abs(taus - c["taus"]).max() < 1e-12 and np.array_equal(hyb, c["hashybs"]) and np.array_equal(cas, c["cas"])
            assert np.array_equal(orbs, c["orbs"]) and np.array_equal(spins, c["spins"])
        for k in keys:
            # every bin is divided by the mean sign of the whole run, the oracle chain by its own (3-orbital Kanamori: not always 1)
            assert rel(getattr(q, k + "_bins")[w] * q.mean_sign, r[k] * r["mean_sign"]) < TOL, (k, w)
            pooled[k] = pooled[k] + r[k] * r["mean_sign"] * (oc["cnt_meas_w"] * r["counters"][11] / max(oc["cnt_sampling_w"], 1.0))
        cw += oc["cnt_sampling_w"]; cz += r["counters"][11]; mw += oc["cnt_meas_w"]
        assert rel(q.histo_bins[w], r["histo"]) < TOL and rel(q.occ_bins[w] * q.mean_sign, r["occ"] * r["mean_sign"]) < TOL
        o.close()
    assert mw > 0
    if float(g["res_accwormrep"][sec - 2]) > 0.01:
        assert nrep > 0                                 # the replacement with a moved partner is in play
    if name.startswith("X"):
        assert n4 > 0                                   # accepted two-pair / outer-block moves
    for k in keys:
        assert rel(getattr(q, k) * q.mean_sign, pooled[k] * cw / (mw * cz)) < TOL, k
    assert q.cntsampling[0] == cz and q.cntsampling[sec - 1] == cw
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("name", WORM2_CASES)
def test_g_worm_sampling_vs_reference_statistics(name):
    """The REFERENCE's giw_worm / gtau_worm of one short chain (worm2_*.npz, mt19937) is one draw from the distribution
    of the single-chain bins of a GPU run with the same parameters (6 sigma of the bin scatter, elementwise)."""
    g = util.load_golden("worm2_%s.npz" % name)
    comp = int(g["component"]); nw = 512
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, 6, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(2, comp)
    for k in ["giw_worm"] + (["gtau_worm"] if over.get("WormMeasGtau") else []):
        b = getattr(q, k + "_bins"); m = b.mean(axis=0)
        sig = np.sqrt(b.real.var(axis=0) + b.imag.var(axis=0))
        ref = g["res_" + k]
        assert np.all(np.abs(ref - m) < 6.0 * sig + 1e-12), (k, np.abs(ref - m).max(), sig.max())
        assert np.abs(getattr(q, k) - m).max() < 0.3 * sig.max()
    assert abs(q.accwormadd[0] - float(g["res_accwormadd"][0])) < 0.05
    assert abs(q.accwormrep[0] - float(g["res_accwormrep"][0])) < 0.1
    q.dest_ctqmc(); q.close()


def test_solver_worm_giw_agrees_with_z_space_estimate():
    """HipCtHybSolver.solve_worm (impurity.py:906-1010): G(iw) assembled from the sector-2 worm components agrees with
    the Z-space Legendre estimate of the same problem within the error bars of the two runs."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Nwarmups=3000, Nwarmups2Plus=3000, Nmeas=300, NCorr=20, TaudiffMax=6.0, Nftau=501, Niw=10, NLegMax=30, NLegOrder=30,
                PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=0.01, WormMeasGiw=1, WormSearchEta=1)
    s = HipCtHybSolver(base, seed=11, nwalkers=512, walkers_per_bin=16)
    s.set_problem(synth.config("C3", nftau=501))
    rw = s.solve_worm()
    gw = np.asarray(rw["giw"].mean()); ew = np.asarray(rw["giw"].stderr())
    s.set_problem(synth.config("C3", nftau=501))
    rz = s.solve()
    gz = np.asarray(rz["giw"].mean()); ez = np.asarray(rz["giw"].stderr())
    assert gw.shape == gz.shape == (20, 2, 2, 2, 2)
    for b in range(2):
        for sp in range(2):
            d = np.abs(gw[:, b, sp, b, sp] - gz[:, b, sp, b, sp]); e = np.sqrt(ew[:, b, sp, b, sp] ** 2 + ez[:, b, sp, b, sp] ** 2)
            assert np.all(d < 5.0 * e + 2e-3), (b, sp, d.max(), e.max())      # 2e-3: Legendre truncation at 30 coefficients
    assert np.abs(gw[:, 0, 0, 0, 0].imag).max() > 0.1 and np.all(gw[10:, 0, 0, 0, 0].imag < 0)
    assert np.abs(gw[:, 0, 0, 1, 0]).max() == 0.0
    occ_w = np.asarray(rw["occ"].mean()); occ_z = np.asarray(rz["occ"].mean())
    assert np.abs(occ_w - occ_z).max() < 0.02


@pytest.mark.parametrize("name", ["C1", "C3", "C1off"])
def test_gtau_accumulator_modes_chains_vs_oracle(name):
    """a-19: AccumGtau = 2 keeps one autocorrelation accumulator per chain (as the reference keeps one per rank); with
    one chain per bin gtau_blocks and G(tau) of every chain agree with the oracle chain of the same stream, the pooled
    gtau_blocks is the chain average.  AccumGtau = 0 measures neither G(tau) nor G(l) (CTQMC.F90:2267)."""
    g = util.load_golden("accum_%s.npz" % name)
    over = dict(g["overrides"]); over.setdefault("MeasGiw", 1); over.setdefault("NLookup_nfft", 0)
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    nw = 6
    q = make_gpu(p, ps, 29, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(1, 1)
    blocks = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 29, w))
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure()
        r = o.results()
        if over["AccumGtau"] == 2:
            assert rel(q.gtau_bins[w], r["gtau"]) < TOL and rel(q.gleg_bins[w], r["gleg"]) < TOL
            assert q.gtau_blocks_bins[w].shape == r["gtau_blocks"].shape == g["res_gtau_blocks"].shape
            assert rel(q.gtau_blocks_bins[w], r["gtau_blocks"]) < TOL, w
            blocks.append(r["gtau_blocks"])
        else:
            assert np.all(q.gtau_bins[w] == 0.0) and np.all(q.gleg_bins[w] == 0.0)
            assert rel(q.giw_bins[w], r["giw"]) < TOL
        assert rel(q.occ_bins[w], r["occ"]) < TOL
        o.close()
    if blocks:
        assert rel(q.gtau_blocks, np.mean(blocks, axis=0)) < TOL
        # the reference's single chain is one draw: its level-0 variance lies inside the chain scatter (loosely)
        ref0 = g["res_gtau_blocks"][..., 0]; mine = np.array(blocks)[..., 0]
        assert np.all(ref0 < 6.0 * mine.max(axis=0) + 1e-12)
    q.dest_ctqmc(); q.close()


def test_solver_worm_giw_kanamori_agrees_with_z_space_estimate():
    """solve_worm on the matrix trace path (2-orbital Kanamori): G(iw) from the sector-2 worm components agrees with the
    Z-space Legendre estimate within the error bars of the two runs."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Nwarmups=3000, Nwarmups2Plus=3000, Nmeas=300, NCorr=20, TaudiffMax=3.0, Nftau=501, Niw=10, NLegMax=30, NLegOrder=30,
                PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=0.01, WormMeasGiw=1, WormSearchEta=1)
    s = HipCtHybSolver(base, seed=12, nwalkers=512, walkers_per_bin=16)
    s.set_problem(synth.config("C4S", nftau=501))
    rw = s.solve_worm()
    gw = np.asarray(rw["giw"].mean()); ew = np.asarray(rw["giw"].stderr())
    s.set_problem(synth.config("C4S", nftau=501))
    rz = s.solve()
    gz = np.asarray(rz["giw"].mean()); ez = np.asarray(rz["giw"].stderr())
    for b in range(2):
        for sp in range(2):
            d = np.abs(gw[:, b, sp, b, sp] - gz[:, b, sp, b, sp]); e = np.sqrt(ew[:, b, sp, b, sp] ** 2 + ez[:, b, sp, b, sp] ** 2)
            assert np.all(d < 5.0 * e + 2e-3), (b, sp, d.max(), e.max())
    assert np.all(gw[10:, 0, 0, 0, 0].imag < 0)
    do = np.abs(np.asarray(rw["occ"].mean()) - np.asarray(rz["occ"].mean()))
    eo = np.sqrt(np.asarray(rw["occ"].stderr()) ** 2 + np.asarray(rz["occ"].stderr()) ** 2)
    print("occ: largest difference", do.max(), "in units of the combined error", (do / (eo + 1e-12)).max())
    assert np.all(do < 5.0 * eo + 0.01), (do.max(), eo.max())


def test_worm_modes_refused_or_ignored():
    """eta = 0 samples nothing (CTQMC.F90:8409-8418)."""
    g = util.load_golden("worm_C1a.npz")
    q, p, pset, over = _worm_gpu(g, 4, 2, 3)
    q.wormeta[:] = 0.0
    q.ctqmc_measure(4, 1)
    assert q.aborted == 0 and np.all(q.g4iw_worm == 0.0) and q.cntsampling[3] == 0
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("name", ["C1", "C3", "C4S"])
def test_window_calibration_chains_vs_oracle(name):
    """ctqmc_calibrate, both phases: every chain owns a list / a completion criterion like a rank of the reference does
    and retraces the oracle chain with the same stream (scalar and matrix trace path)."""
    g = util.load_golden("calib_%s.npz" % name)
    over = g["overrides"]; n1, grid, n2, nprog = (int(x) for x in g["phases"])
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    nw = 5
    q = make_gpu(p, ps, 17, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    q.init_counters()
    q.ctqmc_calibrate(True, n1, grid, n2, nprog)
    lists = q.accpairtau_chains
    assert rel(q.accpairtau, np.sort(lists, axis=1).mean(axis=0)) < 1e-14
    tdm = float(np.sort(q.accpairtau)[int(0.995 * n1)])
    q.set_taudiffmax(tdm)
    q.init_counters()
    q.ctqmc_calibrate(False, n1, grid, n2, nprog)
    acc = q.accpair
    tot = 0
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 17, w))
        o.set_taudiffmax(over["TaudiffMax"]); o.init_counters()
        o.calibrate(True, n1, grid, n2, nprog)
        assert np.abs(o.accpairtau(n1) - lists[w]).max() < 1e-12, w
        o.set_taudiffmax(tdm); o.init_counters()
        o.calibrate(False, n1, grid, n2, nprog)
        tot = tot + o.accpair()
        c = o.get_config()
        n, osst, _ = q.get_mc_config_scalars(w)
        assert n == len(c["taus"]) and osst == c["outer_sst"]
        if n:
            assert np.abs(q.get_mc_config(n, w)[0] - c["taus"]).max() < 1e-12
        o.close()
    assert np.array_equal(acc, tot)
    # the reference's single chain is one draw of the same procedure: its provisional window lies inside the chain scatter
    t_ch = np.sort(lists, axis=1)[:, int(0.995 * n1)]
    assert t_ch.min() * 0.5 < float(g["taudiffmax"]) < t_ch.max() * 2.0
    # a production run after the calibration uses the default histogram again
    q.ctqmc_warmup(200); q.ctqmc_measure(1, 1)
    assert q.accpair.shape[-1] == over["Ntau"]
    q.dest_ctqmc(); q.close()


def test_solver_calibrates_window_when_taudiffmax_not_positive():
    """HipCtHybSolver.solve with TaudiffMax <= 0 runs the two calibration phases (impurity.py:589-591, 1150-1195)."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    s = HipCtHybSolver(dict(BASE, Nwarmups=2000, Nmeas=20, NCorr=20, TaudiffMax=-1.0, Nftau=501), seed=5, nwalkers=64, walkers_per_bin=8)
    s.set_problem(synth.config("C3", nftau=501))
    q = s.ctqmc
    p = s.problem
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    tdm = s.calibrate_taudiff_max(phase1pairnum=300, phase2taugrid=40, phase2pairnum=200)
    assert 1.0 < tdm < p.beta
    q.dest_ctqmc()
    r = s.solve()
    assert np.allclose(np.abs(np.asarray(r["sign"].mean())), 1.0)
    assert np.all(np.isfinite(np.asarray(r["gleg"].mean())))


def test_solver_worm_component_with_eta_search():
    """HipCtHybSolver.solve_component (impurity.py:656-892): eta search by count_steps balances the two spaces, the
    worm estimator comes back as a DistributedSample normalised by nflavours^4."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    s = HipCtHybSolver(dict(BASE, Nwarmups=3000, Nwarmups2Plus=4000, Nmeas=200, NCorr=20, TaudiffMax=6.0, Nftau=501, N4iwf=2, N4iwb=1,
                            PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, WormSearchEta=1),
                       seed=9, nwalkers=256, walkers_per_bin=8)
    s.set_problem(synth.config("C3", nftau=501), compute_fourpoint=8)
    cfg = []
    r = s.solve_component(0, 4, 16, cfg)
    assert len(cfg) == 1 and len(cfg[0]["taus"]) > 0
    eta = r["worm-eta/00016"]
    steps = r["steps-worm-partition/00016"]
    assert eta > 0 and 0.5 < steps[3] / steps[0] < 2.0
    g = np.asarray(r["g4iw-worm/00016"].mean())
    assert g.shape == (4, 4, 3) and np.all(np.isfinite(g)) and np.abs(g).max() > 0
    # second component from the stored configuration, root-finding search
    s.qmc["WormSearchEta"] = 2
    s.set_problem(synth.config("C3", nftau=501), compute_fourpoint=8)
    r2 = s.solve_component(0, 4, 1, cfg)
    st = r2["steps-worm-partition/00001"]
    # brentq stops at xtol = 0.001 in eta (impurity.py:744-779), a sizeable relative error for eta of a few 1e-3: looser band
    assert 0.3 < st[3] / st[0] < 3.0


def test_solver_p2_worm_agrees_with_z_space_density_correlation():
    """HipCtHybSolver.solve_component in sector 6 (P2, equal-time pairs c c+ = 1 - n): p2tau-worm of the component (f f g g),
    times nflavours^4 / 16 (the 2^4 of wormNorm, CTQMC.F90:8830), is <(1 - n_f)(tau) (1 - n_g)(0)> - which the Z-space estimator
    MeasSusz gives independently as 1 - <n_f> - <n_g> + <n_f(tau) n_g(0)>.  Two estimators, two sampling spaces, one number:
    equal-flavour and different-flavour component within the error bars; the eta search balances the spaces."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Ntau=20, Nwarmups=20000, Nmeas=1000, NCorr=20, TaudiffMax=6.0, Nftau=501)
    p = synth.config("C1", nftau=501); nfl = 2 * p.nd
    s = HipCtHybSolver(dict(base, MeasSusz=1), seed=3, nwalkers=1024, walkers_per_bin=8)
    s.set_problem(p)
    r = s.solve(0)
    chi = np.asarray(r["ntau-n0"].mean()); occ = np.asarray(r["occ"].mean())
    for f, g in ((0, 1), (0, 0)):
        comp = 1 + ((f * nfl + f) * nfl + g) * nfl + g
        s2 = HipCtHybSolver(dict(base, Nwarmups2Plus=4000, N2iwb=3, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0,
                                 WormSearchEta=1, WormMeasP2tauPH=1, WormMeasP2iwPH=1), seed=5, nwalkers=1024, walkers_per_bin=8)
        s2.set_problem(p)
        r2 = s2.solve_component(0, 6, comp)
        st = r2["steps-worm-partition/%05d" % comp]
        assert 0.5 < st[5] / st[0] < 2.0
        pt = np.asarray(r2["p2tau-worm/%05d" % comp].mean()) * nfl ** 4 / 16.0
        pe = np.asarray(r2["p2tau-worm/%05d" % comp].stderr()) * nfl ** 4 / 16.0
        hh = 1.0 - occ[f // 2, f % 2, f // 2, f % 2] - occ[g // 2, g % 2, g // 2, g % 2] + chi[f // 2, f % 2, g // 2, g % 2]
        mid = 0.5 * (hh[:-1] + hh[1:])                  # the worm estimator averages over a tau bin, the Z-space one sits on the bin edges
        assert np.all(np.abs(pt - mid) < 5.0 * pe + 0.01 * np.abs(mid)), (f, g, np.abs(pt - mid).max(), pe.max())
        # the static component of p2iw is the integral of the same correlator over tau
        piw = np.asarray(r2["p2iw-worm/%05d" % comp].mean()) * nfl ** 4 / 16.0
        assert abs(piw[3].real - pt.mean() * p.beta) < 0.02 * abs(piw[3].real) and abs(piw[3].imag) < 1e-9


def test_solver_gsigma_worm_agrees_with_dyson_self_energy():
    """HipCtHybSolver.solve_component in sector 3 (GSigma: the improved estimator, a three-operator equal-time object with the
    U matrix in the proposal): gsigmaiw-worm of the diagonal component is Sigma(iw) G(iw) - which the Z-space run gives
    independently through G(l) -> G(iw) and the Dyson equation (postprocessing).  Hartree term included (Sigma ~ 1 - 0.4i at the
    first frequency for U = 2 at half filling)."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Ntau=100, NLegMax=30, Niw=20, Nwarmups=20000, Nmeas=1000, NCorr=20, TaudiffMax=6.0, Nftau=501,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p = synth.config("C1", nftau=501); nfl = 2 * p.nd
    s = HipCtHybSolver(base, seed=3, nwalkers=1024, walkers_per_bin=8)
    s.set_problem(p)
    r = s.postprocessing(s.solve(0))
    g = np.asarray(r["giw"].mean())[:, 0, 0, 0, 0]; sg = np.asarray(r["siw"].mean())[:, 0, 0, 0, 0] * g
    s2 = HipCtHybSolver(dict(base, Nwarmups2Plus=4000, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormSearchEta=1,
                             WormMeasGSigmaiw=1), seed=5, nwalkers=1024, walkers_per_bin=8)
    s2.set_problem(p)
    r2 = s2.solve_component(0, 3, 1)
    st = r2["steps-worm-partition/00001"]
    assert 0.5 < st[2] / st[0] < 2.0
    gs = np.asarray(r2["gsigmaiw-worm/00001"].mean()); ge = np.asarray(r2["gsigmaiw-worm/00001"].stderr())
    assert gs.shape == sg.shape
    n = gs.shape[0] // 2
    assert np.all(np.abs(gs[n:n + 12] - sg[n:n + 12]) < 5.0 * ge[n:n + 12] + 0.02 * np.abs(sg[n:n + 12])), np.abs(gs - sg)[n:n + 12]
    assert np.abs(gs[n:] - np.conj(gs[:n][::-1])).max() < 1e-12
    # solve_worm with the improved estimator (impurity.py:993-997): G = G0 + G0 GSigma from the components of both spins
    s3 = HipCtHybSolver(dict(base, Nwarmups2Plus=4000, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormSearchEta=1,
                             WormMeasGSigmaiw=1), seed=7, nwalkers=1024, walkers_per_bin=8)
    s3.set_problem(p)
    rw = s3.solve_worm()
    gw = np.asarray(rw["giw"].mean()); ew = np.asarray(rw["giw"].stderr())
    gz = np.asarray(r["giw"].mean()); ez = np.asarray(r["giw"].stderr())
    for sp in range(2):
        d = np.abs(gw[:, 0, sp, 0, sp] - gz[:, 0, sp, 0, sp]); e = np.sqrt(ew[:, 0, sp, 0, sp] ** 2 + ez[:, 0, sp, 0, sp] ** 2)
        assert np.all(d < 5.0 * e + 2e-3), (sp, d.max(), e.max())
    assert np.abs(gw[:, 0, 0, 0, 1]).max() == 0.0


def test_solver_qq_worm_symmetric_improved_estimator():
    """HipCtHybSolver.solve_worm with WormMeasQQ (sector 10: two three-operator equal-time objects q, q+ with the U matrix in the
    proposal, six worm operators on the matrix trace path): G = G0 + G0 (QQ + Sigma_0) G0, the symmetric improved estimator (Eq. (13)
    of PRB 100, 075119; impurity.py:998-1003, 1141-1148), agrees with G(iw) of a Z-space run (G(l) -> G(iw))."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Ntau=100, NLegMax=30, Niw=20, Nwarmups=20000, Nmeas=1000, NCorr=20, TaudiffMax=6.0, Nftau=501,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p = synth.config("C1", nftau=501)
    s = HipCtHybSolver(base, seed=3, nwalkers=1024, walkers_per_bin=8)
    s.set_problem(p)
    r = s.postprocessing(s.solve(0))
    s3 = HipCtHybSolver(dict(base, Nwarmups2Plus=4000, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormSearchEta=1,
                             WormMeasQQ=1), seed=11, nwalkers=1024, walkers_per_bin=8)
    s3.set_problem(p)
    rw = s3.solve_worm()
    st = rw["steps-worm-partition/00001"]
    assert 0.3 < st[9] / st[0] < 3.0
    gw = np.asarray(rw["giw"].mean()); ew = np.asarray(rw["giw"].stderr())
    gz = np.asarray(r["giw"].mean()); ez = np.asarray(r["giw"].stderr())
    for sp in range(2):
        d = np.abs(gw[:, 0, sp, 0, sp] - gz[:, 0, sp, 0, sp]); e = np.sqrt(ew[:, 0, sp, 0, sp] ** 2 + ez[:, 0, sp, 0, sp] ** 2)
        assert np.all(d < 5.0 * e + 2e-3), (sp, d.max(), e.max())
    assert np.abs(gw[:, 0, 0, 0, 1]).max() == 0.0


def test_solver_ucaca_worm_agrees_with_z_space_density_correlation_and_spin_symmetry():
    """HipCtHybSolver.solve_component in sector 14 (Ucaca: two equal-time pairs c+ c, each weighted with the doubly antisymmetrised U
    matrix, CTQMC.F90:5778-5782): on the one-orbital model the component (0 0 1 1) picks (U/2)^2 n_dn(tau) n_up(0), (0 0 0 0) picks
    (U/2)^2 n_dn(tau) n_dn(0) and (0 1 1 0) picks (-U/2)^2 S-(tau) S+(0).  ucacatau-worm times nflavours^4 / 16 times beta is the correlator
    (the module array carries the 2^4 of wormNorm and 1 / beta): the first two agree with MeasSusz of a Z-space run, the third with <n_up n_up> - <n_up n_dn>
    (spin rotation symmetry of the paramagnetic model), within the error bars."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Ntau=20, Nwarmups=20000, Nmeas=1000, NCorr=20, TaudiffMax=6.0, Nftau=501)
    p = synth.config("C1", nftau=501); nfl = 2 * p.nd
    uh2 = (0.5 * float(np.asarray(p.umatrix).reshape(nfl, nfl, nfl, nfl)[0, 1, 0, 1])) ** 2
    s = HipCtHybSolver(dict(base, MeasSusz=1), seed=3, nwalkers=1024, walkers_per_bin=8)
    s.set_problem(p)
    chi = np.asarray(s.solve(0)["ntau-n0"].mean())
    mid = lambda a: 0.5 * (a[:-1] + a[1:])            # the worm estimator averages over a tau bin, the Z-space one sits on the bin edges
    nn_same, nn_diff = mid(0.5 * (chi[0, 0, 0, 0] + chi[0, 1, 0, 1])), mid(0.5 * (chi[0, 1, 0, 0] + chi[0, 0, 0, 1]))
    got = {}
    for c4 in ((0, 0, 1, 1), (0, 0, 0, 0), (0, 1, 1, 0)):
        comp = 1 + ((c4[0] * nfl + c4[1]) * nfl + c4[2]) * nfl + c4[3]
        s2 = HipCtHybSolver(dict(base, Nwarmups2Plus=4000, N2iwb=3, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0,
                                 WormSearchEta=1, WormMeasUcaca=1, WormMeasUcacatau=1), seed=5, nwalkers=1024, walkers_per_bin=8)
        s2.set_problem(p)
        r2 = s2.solve_component(0, 14, comp)
        st = r2["steps-worm-partition/%05d" % comp]
        assert 0.5 < st[13] / st[0] < 2.0
        f = nfl ** 4 / 16.0 * p.beta / uh2             # 16: the 2^4 of wormNorm (CTQMC.F90:8830), as in sector 6
        got[c4] = (np.asarray(r2["ucacatau-worm/%05d" % comp].mean()) * f, np.asarray(r2["ucacatau-worm/%05d" % comp].stderr()) * f)
        # the static component of ucaca-worm is the integral of the same correlator over tau
        uiw = np.asarray(r2["ucaca-worm/%05d" % comp].mean()) * f
        assert abs(uiw[3].real - got[c4][0].mean() * p.beta) < 0.02 * abs(uiw[3].real) and abs(uiw[3].imag) < 1e-9
    for c4, ref in (((0, 0, 1, 1), nn_diff), ((0, 0, 0, 0), nn_same), ((0, 1, 1, 0), nn_same - nn_diff)):
        m, e = got[c4]
        assert np.all(np.abs(m - ref) < 5.0 * e + 0.02 * np.abs(ref)), (c4, np.abs(m - ref).max(), e.max())
    m = got[(0, 0, 0, 0)][0] - got[(0, 0, 1, 1)][0]; e = np.sqrt(got[(0, 0, 0, 0)][1] ** 2 + got[(0, 0, 1, 1)][1] ** 2 + got[(0, 1, 1, 0)][1] ** 2)
    assert np.all(np.abs(got[(0, 1, 1, 0)][0] - m) < 5.0 * e), np.abs(got[(0, 1, 1, 0)][0] - m).max()


def test_solver_components_of_the_u_matrix_worm_sectors():
    """HipCtHybSolver.solve_component in the sectors 5, 14 and 16: the eta search balances worm and partition-function space, the result set carries
    the reference's keys (impurity.py:813-871) with the shapes of the module arrays, divided by nflavours^4, finite, with error bars."""
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd import synth
    base = dict(BASE, Ntau=40, NLegMax=10, Niw=8, N2iwb=3, N4iwf=2, N4iwb=1, Nwarmups=5000, Nmeas=200, NCorr=20, TaudiffMax=6.0, Nftau=501,
                PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05, Nwarmups2Plus=2000, PercentageWormInsert=0.2, PercentageWormReplace=0.2,
                WormEta=1.0, WormSearchEta=1)
    p = synth.config("C1", nftau=501)
    comp = 1 + 0 * 8 + 0 * 4 + 1 * 2 + 1
    for sec, extra, keys, fourpnt in ((14, dict(WormMeasUcaca=1, WormMeasUcacatau=1), {"ucaca-worm": (7,), "ucacatau-worm": (40,)}, 0),
                                      (16, dict(WormMeasQUDdag=1), {"quddag-worm": (16,)}, 0),
                                      (5, dict(WormMeasH4iw=1), {"h4iw-worm": (4, 4, 3)}, 8)):
        s = HipCtHybSolver(dict(base, **extra), seed=21 + sec, nwalkers=256, walkers_per_bin=8)
        s.set_problem(p, fourpnt)
        r = s.solve_component(0, sec, comp)
        st = r["steps-worm-partition/%05d" % comp]
        assert 0.3 < st[sec - 1] / st[0] < 3.0, (sec, st[0], st[sec - 1])
        for key, shape in keys.items():
            m = np.asarray(r["%s/%05d" % (key, comp)].mean()); e = np.asarray(r["%s/%05d" % (key, comp)].stderr())
            assert m.shape == shape and np.all(np.isfinite(m)) and np.abs(m).max() > 0.0 and np.all(np.isfinite(e)), (sec, key, m.shape)


@pytest.mark.parametrize("name", ["C1", "C3", "C4"])
def test_susceptibility_recorded_configuration_vs_reference_golden(name):
    """f-3, MeasSusz = 1 (matrix solver, meas_ntau_n0, Trace.F90:6472-6835): <n(tau) n(0)> of a recorded configuration ==
    what the REFERENCE printed - one window (C1: reference time beta), several windows (C3), and blocks of several states
    (C4, 3-orbital Kanamori: densities inserted into the bra / ket vectors of the matrix trace path, k_meas_susz_mat)."""
    g = util.load_golden("susz_%s.npz" % name)
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, int(g["seed"]), 4, wpb=4)
    q.set_taudiffmax(over["TaudiffMax"])
    c = util.cfg_from(g, "cfg_")
    q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
    q.ctqmc_measure(1, 1)
    assert q.ntau_n0.shape == g["res_ntau_n0"].shape
    assert rel(q.ntau_n0, g["res_ntau_n0"]) < TOL
    q.dest_ctqmc(); q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["C1", "C3", "C4S"])
def test_step_series_are_the_chain_means_of_the_oracle_series(cfg):
    """Gtau_mean_step, Gtau_mid_step, sign_step (CTQMC.F90:2394-2402): entry m of the HIP series is the mean over the chains of
    what every chain's oracle twin (same stream) records at its m-th measurement; C1 has empty configurations (sign entry 0),
    C4S runs on the matrix trace path."""
    over = dict(BASE, Nwarmups=2000, Nmeas=12, NCorr=30, TaudiffMax=3.0 if cfg == "C4S" else 6.0, PercentageGlobalMove=0.1,
                PercentageTauShiftMove=0.1, Gtau_mean_step=1, Gtau_mid_step=1, sign_step=1)
    p, ps, pset = util.setup(cfg, 501, over)
    nw = 6
    q = make_gpu(p, ps, 57, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(1, 1)
    tot = {}
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 57, w))
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure()
        r = o.results()
        for k in ("gtau_mean_step", "gtau_mid_step", "sign_step"):
            tot[k] = tot.get(k, 0.0) + r[k] / nw
        o.close()
    for k in tot:
        a = getattr(q, k)
        assert a.shape == tot[k].shape and np.abs(tot[k]).max() > 0, k
        assert np.abs(a - tot[k]).max() < 1e-10 * np.abs(tot[k]).max(), k
    q.dest_ctqmc(); q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C4", "C4N", "C5K"])
def test_ssquare_recorded_configuration_vs_reference_golden(name):
    """f-3, MeasSsquare = 1 (CTQMC.F90:2769-2840): <S^2> per electron number and the S^2 operator of the occupation basis ==
    what the REFERENCE printed for the same recorded configuration (tree path: C4; sequential path: C4N, C5K)."""
    g = util.load_golden("ssq_%s.npz" % name)
    over = g["overrides"]
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = make_gpu(p, ps, int(g["seed"]), 4, wpb=4)
    q.set_taudiffmax(over["TaudiffMax"])
    c = util.cfg_from(g, "cfg_")
    q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
    q.ctqmc_measure(1, 1)
    assert q.ssquare.shape == g["res_ssquare"].shape and q.ssquare_operator.shape == g["res_ssquare_operator"].shape
    assert rel(q.ssquare, g["res_ssquare"]) < TOL
    assert np.abs(q.ssquare_operator - g["res_ssquare_operator"]).max() < 1e-12
    assert rel(q.densitymatrix, g["res_densitymatrix"]) < TOL
    q.dest_ctqmc(); q.close()


@pytest.mark.gpu
def test_ssquare_is_refused_when_the_blocks_do_not_conserve_the_spin():
    """One-state blocks (Nt Szt Azt): the spin-flip terms of S^2 leave the block, the reference adds them to the wrong
    element - refused, not imitated."""
    p, ps, pset = util.setup("C3", 501, dict(BASE, MeasSsquare=1))
    q = MCTQMC(nwalkers=4)
    q.init_rnd(1); q.init_paras(ps)
    with pytest.raises(W2DynError, match="MeasSsquare"):
        q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.close()


@pytest.mark.parametrize("budget", [None, 100000], ids=["bin-per-chain", "folded-bins"])
def test_susceptibility_chains_vs_oracle(monkeypatch, budget):
    """MeasSusz in a sampled run (window shifts, global and tau-shift moves): every chain's <n(tau) n(0)> agrees with the
    oracle chain of the same stream; with a small accumulator budget several sample bins share an accumulator bin
    (atomic adds) and the pooled result is still the mean over the chains.  Ntau = 300: two tau points per thread."""
    if budget:
        monkeypatch.setenv("W2DYN_ACC_BYTES", str(budget))
    over = dict(BASE, Nwarmups=3000, Nmeas=12, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1,
                MeasSusz=1, Ntau=300)
    p, ps, pset = util.setup("C3", 501, over)
    nw = 6
    q = make_gpu(p, ps, 41, nw, wpb=1)
    q.set_taudiffmax(over["TaudiffMax"])
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(1, 1)
    nbs = int(q.scalar("nbs"))
    assert nbs == (nw if not budget else 100000 // (16 * 300 * 8))
    res = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 41, w))
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure()
        res.append(o.results()["ntau_n0"]); o.close()
    res = np.array(res)
    if not budget:
        for w in range(nw):
            assert rel(q.ntau_n0_bins[w], res[w]) < TOL, w
    else:
        for b in range(nbs):
            assert rel(q.ntau_n0_bins[b], res[b::nbs].mean(axis=0)) < TOL, b
    assert rel(q.ntau_n0, res.mean(axis=0)) < TOL
    q.dest_ctqmc(); q.close()


@pytest.mark.parametrize("cfg,tdm", [("C4S", 3.0), ("C3Q", 5.0)], ids=["C4S-kanamori", "C3Q-coarse-blocks"])
def test_susceptibility_matrix_path_chains_vs_oracle(cfg, tdm):
    """MeasSusz on the matrix trace path in a sampled run (window shifts, global and tau-shift moves, several windows): every
    chain's <n(tau) n(0)> agrees with the oracle chain of the same stream, which takes every one of the nflav^2 Ntau traces
    from scratch with the four density operators in the list."""
    over = dict(BASE, Nwarmups=2000, Nmeas=6, NCorr=30, TaudiffMax=tdm, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1, MeasSusz=1, Ntau=70)
    p, ps, pset = util.setup(cfg, 501, over)
    nw = 5
    q = make_gpu(p, ps, 43, nw, wpb=1)
    q.set_taudiffmax(tdm)
    q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(1, 1)
    res = []
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", 43, w))
        o.set_taudiffmax(tdm); o.warmup(over["Nwarmups"]); o.measure()
        res.append(o.results()["ntau_n0"]); o.close()
    res = np.array(res)
    assert np.abs(res).max() > 0.1
    for w in range(nw):
        assert rel(q.ntau_n0_bins[w], res[w]) < TOL, w
    assert rel(q.ntau_n0, res.mean(axis=0)) < TOL
    q.dest_ctqmc(); q.close()


def test_susceptibility_is_refused_beyond_blocks_of_eight_states():
    p, ps, pset = util.setup("C4N", 501, dict(BASE, Nwarmups=0, Nmeas=1, NCorr=1, TaudiffMax=3.0, MeasSusz=1))      # blocks of up to 9 states
    q = MCTQMC(nwalkers=4, walkers_per_bin=1)
    q.init_rnd(1); q.init_paras(ps)
    with pytest.raises(W2DynError) as e:
        q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    assert e.value.code == -2 and "MeasSusz" in str(e.value)
    q.close()
