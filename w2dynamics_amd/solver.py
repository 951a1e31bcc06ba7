"""Host-side counterpart of the reference's solver wrapper for the Z-space path.

`HipCtHybSolver` reproduces the call order of `CtHybSolver.set_problem` / `solve` (`w2dyn/dmft/impurity.py:497-534,
571-653`) on top of the `mctqmc`-shaped HIP front end, one process per GPU:

  init_rnd(seed) -> init_paras(pstring) -> fourpnt -> init_solver(U, ftau, -muimp, screening, iter+1)
  -> set_taudiffmax -> ctqmc_warmup -> ctqmc_measure(1,1) -> result pickup -> DistributedSample wrapping
  -> dest_ctqmc / dest_paras

Chains are sharded over ranks without any collective in the sampling path: rank r runs chains
[r*nwalkers, (r+1)*nwalkers) of one global stream family (reference: seed + rank, impurity.py:381).  The only
exchange step is the final reduction (RCCL all-reduce): the one-particle results, whose bins the jackknife needs on the
host anyway, go through `statistics.DistributedSample`; the two-particle arrays (hundreds of MB per GPU at production
frequency boxes) are reduced to their moments on the GPU and all-reduced there (`MCTQMC.reduce_pack`,
`statistics.PackedMoments`), only mean and error bar ever reach the host.
"""
from __future__ import annotations

import time

import numpy as np

from . import params as _params
from . import postproc as _postproc
from .mctqmc import MCTQMC
from .statistics import DistributedSample, TorchComm

RESULT_KEYS_BINNED = ("gleg", "gtau", "occ", "histo")

# `CtHybSolver.get_resultset` (impurity.py:1198-1283): result key -> module variable, always present / present under a
# condition on the QMC configuration.  The keys are the HDF5 quantity names (`<iter>/ineq-NNN/<key>/{value,error}`).
RESULTSET_ALWAYS = {
    "gtau": "gtau", "gtau-full": "gtau_full", "rhist": "rhisto", "lhist": "lhisto", "hist": "histo",
    "ssts-states": "statessuperstates", "energies-eigenstates": "eigenstatesenergies", "occbasis-mapping": "occbasismapping",
    "hist-sst": "outersuperstatehisto", "hist-state": "outerstatehisto", "sign-sst": "signhistosuperstates",
    "sign-state": "signhistostates", "contrib-sst": "tracecontribsuperstates", "contrib-state": "tracecontribstates",
    "occ": "occ", "gleg": "gleg", "gleg-full": "gleg_full", "accept-ins": "accadd", "accept-ins4": "accadd4",
    "accept-rem": "accrem", "accept-rem4": "accrem4", "accept-pair-tau": "accpair", "accept-glob": "accglob",
    "accept-shift": "accshift", "accept-flavourchange": "accflavc", "sign": "mean_sign", "time-warmup": "time_warmup",
    "time-simulation": "time_sim", "time-sampling": "time_sim_steps",
}
RESULTSET_CONDITIONAL = (      # (key, module variable, condition on the QMC configuration and FourPnt)
    ("giw-meas", "giw", lambda c, fp: c["MeasGiw"] != 0 and c["offdiag"] == 0),
    ("g2iw", "g2iw", lambda c, fp: c["MeasG2iw"] != 0),
    ("densitymatrix", "densitymatrix", lambda c, fp: c["MeasDensityMatrix"] != 0 and c["segment"] == 0),
    ("rho2", "rho2", lambda c, fp: c["MeasDensityMatrix"] != 0 and c["segment"] == 0),
    ("rho1", "rho1", lambda c, fp: c["MeasDensityMatrix"] != 0 and c["segment"] == 0),
    ("g4tau", "g4tau", lambda c, fp: fp & 1), ("g4leg", "g4leg", lambda c, fp: fp & 2), ("g4iw", "g4iw", lambda c, fp: fp & 4),
    ("g4iw-pp", "g4iw_pp", lambda c, fp: fp & 4 and c["MeasG4iwPP"] != 0),
    ("ntau-n0", "ntau_n0", lambda c, fp: c["segment"] == 0 and c["MeasSusz"] != 0),
    ("ssquare-operator", "ssquare_operator", lambda c, fp: c["MeasSsquare"] != 0),
    ("gtau-mean-step", "gtau_mean_step", lambda c, fp: c["Gtau_mean_step"] != 0),
    ("gtau-mid-step", "gtau_mid_step", lambda c, fp: c["Gtau_mid_step"] != 0),
    ("sign-step", "sign_step", lambda c, fp: c["sign_step"] != 0),
    ("s2", "ssquare", lambda c, fp: c["MeasSsquare"] != 0 and (c["MeasDensityMatrix"] != 0 or c["Eigenbasis"] != 0)),
    ("gtau-blocks", "gtau_blocks", lambda c, fp: c["AccumGtau"] == 2),
)
# quantities with one observation per accumulator bin (their error bar comes from the bins): key -> "<variable>_bins"
RESULTSET_BINNED = {"gtau": "gtau_bins", "gleg": "gleg_bins", "occ": "occ_bins", "hist": "histo_bins", "giw-meas": "giw_bins",
                    "hist-sst": "outersuperstatehisto_bins", "hist-state": "outerstatehisto_bins", "sign-sst": "signhistosuperstates_bins",
                    "sign-state": "signhistostates_bins", "contrib-sst": "tracecontribsuperstates_bins",
                    "contrib-state": "tracecontribstates_bins", "gtau-blocks": "gtau_blocks_bins", "ntau-n0": "ntau_n0_bins"}


def hdf5_layout(result):
    """What `HdfOutput.write_impurity_result` (hdfout.py:369-397) writes below `ineq-NNN` for a result dictionary:
    `<quantity>/value` and `<quantity>/error` for samples, `<quantity>/value` for plain arrays and numbers."""
    out = {}
    for key, val in result.items():
        if isinstance(val, DistributedSample):
            out[key] = {"value": np.asarray(val.mean()), "error": np.asarray(val.stderr())}
        else:
            out[key] = {"value": np.asarray(val)}
    return out


class HipCtHybSolver:
    def __init__(self, qmc_overrides, seed, comm=None, device=0, nwalkers=1024, walkers_per_bin=8, kmax=0):
        self.comm = comm if comm is not None else TorchComm()
        self.rank = self.comm.Get_rank()
        self.qmc = dict(_params.QMC_DEFAULTS)
        self.qmc.update(qmc_overrides or {})
        self.seed = seed
        self.nwalkers = nwalkers
        self.ctqmc = MCTQMC(device=device, nwalkers=nwalkers, first_walker=self.rank * nwalkers,
                            kmax=kmax, walkers_per_bin=walkers_per_bin)
        self.ctqmc.init_rnd(seed)
        self.ctqmc.fancy_prog = False
        self.fttype = "legendre"         # [General] FTType of the reference's configuration (impurity.py:1054)
        self.last_local = {}             # this rank's own values of quantities that are averaged over the ranks (eta, window)

    def set_problem(self, problem, compute_fourpoint=0):
        """problem: w2dynamics_amd.synth.Problem (arrays already in init_solver layout, muimp negated)."""
        self.problem = problem
        pstr = _params.config_to_pstring(self.qmc) + _params.config_to_pstring({
            "Hamiltonian": problem.hamiltonian, "beta": problem.beta, "Nd": problem.nd, "ParaMag": 0,
            "QuantumNumbers": problem.quantum_numbers, "Phonon": 0, "g_phonon": "", "omega0": "", "Screening": 0,
            "Uw": 0}) + _params.symm_to_pstring([])
        self.ctqmc.init_paras(pstr)
        self.ctqmc.fourpnt = compute_fourpoint

    def solve(self, iter_no=0):
        q = self.ctqmc
        self.comm.Barrier()
        q.simid = self.rank
        t0 = time.perf_counter()
        p = self.problem
        q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, iter_no + 1)
        tdm = self.qmc["TaudiffMax"]
        q.set_taudiffmax(tdm if tdm > 0.0 else self.calibrate_taudiff_max())       # impurity.py:589-591
        q.ctqmc_warmup(self.qmc["Nwarmups"])
        q.ctqmc_measure(1, 1)
        if q.aborted == 2:
            raise KeyboardInterrupt("CTQMC was interrupted by signal")
        result = self.get_resultset()
        if q.fourpnt & 4:      # get_resultset, impurity.py:1253-1262: g4iw, g4iw-pp, g2iw; moments made on the GPU
            names = {"g4iw": "g4iw"}
            if self.qmc["MeasG4iwPP"]:
                names["g4iw-pp"] = "g4iw_pp"
            if self.qmc["MeasG2iw"]:
                names["g2iw"] = "g2iw"
            pack = q.reduce_pack(names.values(), self.comm).allreduce()
            for key, name in names.items():
                result[key] = pack.sample(name)
        # Green's function for the Dyson equation, (iw, band, spin, band, spin), one sample per accumulator bin
        # (impurity.py:611-619), and the self-energy moments from rho1 / rho2 (impurity.py:628-640)
        niw = 2 * int(self.qmc["Niw"])
        fttype = "none" if (self.fttype == "none" and self.qmc["MeasGiw"]) else "legendre"
        src = q.giw_bins if fttype == "none" else q.gleg_bins
        key = "giw-meas" if fttype == "none" else "gleg"
        result["giw"] = DistributedSample(np.array([_postproc.get_giw({key: b}, niw, fttype) for b in src]), self.comm)
        nfl = 2 * p.nd
        result["smom"] = DistributedSample(np.array([_postproc.lattice_convention(
            _postproc.get_siw_mom(np.asarray(p.umatrix).reshape((nfl,) * 4), q.rho1, q.rho2))]), self.comm)
        result["time-qmc"] = time.perf_counter() - t0         # a plain number, as in impurity.py:610
        q.dest_ctqmc()
        q.dest_paras()
        return result

    def get_resultset(self):
        """`CtHybSolver.get_resultset` (impurity.py:1198-1283) on the module variables of the HIP front end: the same keys
        under the same conditions, each a DistributedSample (impurity.py:641-642) - one observation per accumulator bin
        where the library keeps bins, else one per rank.  The two-particle arrays of a FourPnt = 4 run are replaced by
        their device-made moments in `solve`.  `rho1` / `rho2` in (band, spin, ...) shape as there."""
        q, c, fp = self.ctqmc, self.qmc, int(self.ctqmc.fourpnt)
        nd = self.problem.nd
        names = dict(RESULTSET_ALWAYS)
        names.update({key: var for key, var, cond in RESULTSET_CONDITIONAL if cond(c, fp)})
        result = {}
        for key, var in names.items():
            if key in ("g4iw", "g4iw-pp", "g2iw"):
                if key == "g2iw" and not fp & 4:        # allocated and zeroed, but only measured along with g4iw (CTQMC.F90:838-840, 2157)
                    n2 = 2 * (int(c["N4iwf"]) + int(c["N4iwb"]))
                    result[key] = DistributedSample(np.zeros((1, nd, 2, n2, n2), dtype=complex), self.comm)
                continue                                # device-made moments (solve)
            if key in RESULTSET_BINNED:
                local = np.asarray(getattr(q, RESULTSET_BINNED[key]))
            else:
                local = np.asarray(getattr(q, var))[None]
            if key == "rho1":
                local = local.reshape((1,) + (nd, 2) * 2)
            elif key == "rho2":
                local = local.reshape((1,) + (nd, 2) * 4)
            result[key] = DistributedSample(local, self.comm)
        return result

    def postprocessing(self, result, g0inviw=None):
        """`ImpurityResult.postprocessing(siw_method="dyson", smom_method="estimate")` (impurity.py:208-225): adds
        result["siw"] = G0^-1 - G^-1 as jackknife pseudovalues over the accumulator bins of all ranks."""
        niw = 2 * int(self.qmc["Niw"])
        if g0inviw is None:
            g0inviw = self.problem.g0inviw(niw)
        result["siw"] = _postproc.siw_dyson(result["giw"], g0inviw, diagonal_only=True)
        return result

    def solve_component(self, iter_no, isector, icomponent, mc_config_inout=None):
        """Worm estimator of one component in the call order of `CtHybSolver.solve_component` (`impurity.py:656-892`):
        init_solver -> (stored configuration | warm-up) -> eta search by count_steps -> ctqmc_worm_warmup ->
        ctqmc_measure(isector, icomponent) -> result normalised by nflavours^4.  Sectors 2 (one-particle Green's
        function: giw-worm, gtau-worm), 4 (two-particle Green's function) and 6 - 9 (equal-time pairs: p2iw-worm, p2tau-worm,
        p2iwpp-worm, p2taupp-worm, p3iw-worm, p3iwpp-worm) are the ones the HIP library samples."""
        import scipy.optimize as opt
        q = self.ctqmc
        self.comm.Barrier()
        q.simid = self.rank
        t0 = time.perf_counter()
        p = self.problem
        q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, iter_no + 1)
        if mc_config_inout:
            c = mc_config_inout[0]
            if c.get("all_chains_saved"):      # every chain goes on from its own configuration, as every rank of the reference does
                q.restore_mc_configs()
            else:
                q.set_mc_config(len(c["taus"]), c["taus"], c["orbs"], c["spins"], c["cas"], c["hashybs"], c["outer_sst"], c["outer_state"])
        tdm = self.qmc["TaudiffMax"]
        q.set_taudiffmax(tdm if tdm > 0.0 else self.calibrate_taudiff_max())
        if not mc_config_inout:
            q.ctqmc_warmup(self.qmc["Nwarmups"])
            if mc_config_inout is not None:
                n, osst, ost = q.get_mc_config_scalars()
                taus, orbs, spins, cas, hyb, osst, ost = q.get_mc_config(n)
                q.save_mc_configs()
                mc_config_inout.append(dict(taus=taus, orbs=orbs, spins=spins, cas=cas, hashybs=hyb, outer_sst=osst, outer_state=ost,
                                            all_chains_saved=True))
        worm_warmups = self.qmc["Nwarmups2Plus"]
        if worm_warmups < 0:
            worm_warmups = int(0.1 * self.qmc["Nwarmups"])

        def imbalance(eta):
            """(steps in worm space - steps in Z space) / all steps of a count_steps run at this eta."""
            q.wormeta[:] = eta
            sw, sz = q.count_steps(isector, icomponent, worm_warmups)
            if sw + sz != worm_warmups:
                raise ValueError("%d steps in worm, %d steps in Z, some steps missing!" % (sw, sz))
            return float(sw - sz) / float(sw + sz), sw, sz

        # what the reference's two searches do (impurity.py:717-779), as one table: mode 1 rescales eta by the ratio of the
        # step counts until the two spaces are balanced within 10 % (50 rounds, eta = 0 when worm space is never reached);
        # mode 2 looks for the root of the imbalance on [1e-4, 1e2], then on [1e-6, 1e4]; otherwise eta = WormEta
        mode = int(self.qmc["WormSearchEta"])
        eta = float(self.qmc["WormEta"])
        if mode == 1:
            for round_ in range(50):
                d, sw, sz = imbalance(eta)
                if abs(d) <= 0.1:
                    break
                eta = 0.0 if (sw == 0 or round_ == 49) else eta * float(sz) / float(sw)
                if eta == 0.0:
                    break
        elif mode == 2:
            eta = 0.0
            for lo, hi in ((1e-4, 1e2), (1e-6, 1e4)):
                try:
                    eta = opt.brentq(lambda e: imbalance(e)[0], lo, hi, xtol=0.001, rtol=0.001, maxiter=50, disp=True)
                    break
                except ValueError:      # no sign change on this bracket
                    continue
        self.last_local["eta"] = float(eta)               # this rank's own value; all ranks go on with the mean
        eta_stat = DistributedSample([np.array([eta])], self.comm)
        eta_mean = float(np.asarray(eta_stat.mean()).ravel()[0])
        q.wormeta[:] = eta_mean
        q.ctqmc_worm_warmup(worm_warmups, isector, icomponent)
        q.ctqmc_measure(isector, icomponent)
        if q.aborted == 2:
            raise KeyboardInterrupt("CTQMC was interrupted by signal")
        nfl = 2 * p.nd
        result = {}
        if isector == 4:       # moments made on the GPU and all-reduced there; normalised by nflavours^4 (impurity.py:1317-1323)
            names = {}
            if self.qmc["G4ph"] == 1:
                names["g4iw-worm/%05d" % icomponent] = "g4iw_worm"
            if self.qmc["G4pp"] == 1:
                names["g4iwpp-worm/%05d" % icomponent] = "g4iwpp_worm"
            if names:
                pack = q.reduce_pack(names.values(), self.comm).allreduce()
                for key, name in names.items():
                    result[key] = pack.sample(name, 1.0 / nfl ** 4)
        if isector == 2:          # get_resultset_worm, impurity.py:1312-1315 (already divided by nflavours^2, CTQMC.F90:8871)
            if self.qmc["WormMeasGiw"] != 0:
                result["giw-worm/%05d" % icomponent] = DistributedSample(q.giw_worm_bins, self.comm)
            if self.qmc["WormMeasGtau"] != 0:
                result["gtau-worm/%05d" % icomponent] = DistributedSample(q.gtau_worm_bins, self.comm)
        if isector == 3 and self.qmc["WormMeasGSigmaiw"] != 0:      # get_resultset_worm, impurity.py:1316-1317
            result["gsigmaiw-worm/%05d" % icomponent] = DistributedSample(q.gsigmaiw_worm_bins, self.comm)
        if 6 <= isector <= 9:     # equal-time sectors, impurity.py:817-839: the module array divided by nflavours^4
            for sec, key, name, flag in ((6, "p2iw-worm", "p2iw_worm", "WormMeasP2iwPH"), (6, "p2tau-worm", "p2tau_worm", "WormMeasP2tauPH"),
                                         (7, "p2iwpp-worm", "p2iwpp_worm", "WormMeasP2iwPP"), (7, "p2taupp-worm", "p2taupp_worm", "WormMeasP2tauPP"),
                                         (8, "p3iw-worm", "p3iw_worm", "WormMeasP3iwPH"), (9, "p3iwpp-worm", "p3iwpp_worm", "WormMeasP3iwPP")):
                if sec == isector and self.qmc[flag] == 1:
                    result["%s/%05d" % (key, icomponent)] = DistributedSample(getattr(q, name + "_bins") / nfl ** 4, self.comm)
        # sectors with U-matrix factors: the module array divided by nflavours^4 (impurity.py:813-815, 841-871); qq / qqtau as they are (1318-1321)
        for sec, key, name, flag in ((5, "h4iw-worm", "h4iw_worm", "WormMeasH4iw"), (11, "qqqq-worm", "qqqq_worm", "WormMeasQQQQ"),
                                     (12, "nqqdag-worm", "nqqdag_worm", "WormMeasNQQdag"), (13, "qqdd-worm", "qqdd_worm", "WormMeasQQdd"),
                                     (14, "ucaca-worm", "ucaca_worm", "WormMeasUcaca"), (14, "ucacatau-worm", "ucacatau_worm", "WormMeasUcacatau"),
                                     (15, "uccaa-worm", "uccaa_worm", "WormMeasUccaa"), (15, "uccaatau-worm", "uccaatau_worm", "WormMeasUccaatau"),
                                     (16, "quddag-worm", "quddag_worm", "WormMeasQUDdag")):
            if sec == isector and self.qmc[flag] != 0:
                result["%s/%05d" % (key, icomponent)] = DistributedSample(getattr(q, name + "_bins") / nfl ** 4, self.comm)
        if isector == 10:
            if self.qmc["WormMeasQQ"] != 0:
                result["qqiw-worm/%05d" % icomponent] = DistributedSample(q.qq_worm_bins, self.comm)
                if self.qmc["WormMeasQQtau"] != 0:
                    result["qqtau-worm/%05d" % icomponent] = DistributedSample(q.qqtau_worm_bins, self.comm)
        # Z-space quantities are sampled along with every component (impurity.py:966-977)
        result["occ"] = DistributedSample(q.occ_bins, self.comm)
        result["rho1"] = DistributedSample(np.array([q.rho1]), self.comm)
        result["rho2"] = DistributedSample(np.array([q.rho2]), self.comm)
        result["worm-eta/%05d" % icomponent] = eta_mean
        result["sign"] = DistributedSample(np.array([q.mean_sign]), self.comm)
        result["steps-worm-partition/%05d" % icomponent] = np.array(q.cntsampling, dtype=np.double)
        result["time-qmc"] = time.perf_counter() - t0
        q.dest_ctqmc()
        q.dest_paras()
        return result

    def solve_worm(self, iter_no=0, components=None, g0inviw=None):
        """`CtHybSolver.solve_worm` (`impurity.py:906-1010`) for `WormMeasGiw` (worm sector 2), `WormMeasGSigmaiw` (sector 3) or `WormMeasQQ` (sector 10):
        every diagonal component is sampled by `solve_component`, the component list is assembled into a matrix in the orbitals -
        G(iw) itself (`get_giw` with FTType `none_worm`, `impurity.py:1066-1072`), or GSigma(iw), from which G = G0 + G0 GSigma
        follows (`get_gsigmaiw_worm`, `giw_from_gsigmaiw_worm`, `impurity.py:1104-1138`: the improved estimator, Eq. (9) of PRB 100,
        075119) -, the Z-space quantities sampled along with every component are averaged and the self-energy moments follow from
        rho1 / rho2.  `g0inviw` (iw, band, spin, band, spin): the Weiss field for the improved estimator, the problem's by default."""
        meas_g, meas_gs, meas_qq = self.qmc["WormMeasGiw"] != 0, self.qmc["WormMeasGSigmaiw"] != 0, self.qmc["WormMeasQQ"] != 0
        if int(meas_g) + int(meas_gs) + int(meas_qq) != 1:
            raise ValueError("solve_worm: exactly one of WormMeasGiw, WormMeasGSigmaiw, WormMeasQQ must be set")
        sector, name = (2, "giw-worm") if meas_g else ((3, "gsigmaiw-worm") if meas_gs else (10, "qqiw-worm"))
        p = self.problem
        nd = p.nd; nfl = 2 * nd; niw = 2 * int(self.qmc["Niw"])
        if components is None:            # ci.GFComponent(bandspin=(f, f), n_ops=2).index, impurity.py:919-921
            components = [1 + f * nfl + f for f in range(nfl)]
        conf, per_comp = [], []
        for comp in components:
            self.set_problem(p, self.ctqmc.fourpnt)
            per_comp.append((comp, self.solve_component(iter_no, sector, comp, conf)))
        result = {}
        for comp, r in per_comp:
            result.update({k: v for k, v in r.items() if k.endswith("/%05d" % comp)})
        nb = min(np.asarray(r["%s/%05d" % (name, c)].local).shape[0] for c, r in per_comp)
        mat = np.zeros((nb, niw, nd, 2, nd, 2), dtype=complex)
        for comp, r in per_comp:
            f1, f2 = divmod(comp - 1, nfl)
            mat[:, :, f1 // 2, f1 % 2, f2 // 2, f2 % 2] = np.asarray(r["%s/%05d" % (name, comp)].local)[:nb]
        if meas_g:
            result["giw"] = DistributedSample(mat, self.comm)
        elif meas_gs:
            if g0inviw is None:
                g0inviw = p.g0inviw(niw)
            g0 = np.linalg.inv(np.asarray(g0inviw).reshape(niw, nfl, nfl))                     # orbspin.invert
            gs = mat.reshape(nb, niw, nfl, nfl)
            result["gsigmaiw"] = DistributedSample(mat, self.comm)
            result["giw"] = DistributedSample((g0[None] + np.einsum("wab,nwbc->nwac", g0, gs)).reshape(nb, niw, nd, 2, nd, 2), self.comm)
        for key in ("occ", "rho1", "rho2"):
            result[key] = DistributedSample(np.concatenate([np.asarray(r[key].local) for _, r in per_comp]), self.comm)
        smom = _postproc.lattice_convention(_postproc.get_siw_mom(
            np.asarray(p.umatrix).reshape((nfl,) * 4), np.asarray(result["rho1"].mean()), np.asarray(result["rho2"].mean())))
        result["smom"] = DistributedSample(np.array([smom]), self.comm)
        if meas_qq:
            # symmetric improved estimator, Eq. (13) of PRB 100, 075119: G = G0 + G0 (QQ + Sigma_0) G0 with the zeroth moment of the self-energy
            # from rho1 (`get_qqiw_worm`, `giw_from_qqiw_worm`, impurity.py:1113-1119, 1141-1148, 1002-1003)
            if g0inviw is None:
                g0inviw = p.g0inviw(niw)
            g0 = np.linalg.inv(np.asarray(g0inviw).reshape(niw, nfl, nfl))
            inner = mat.reshape(nb, niw, nfl, nfl) + np.asarray(smom)[0].reshape(nfl, nfl)[None, None]
            result["qqiw"] = DistributedSample(mat, self.comm)
            result["giw"] = DistributedSample((g0[None] + np.einsum("wab,nwbc,wcd->nwad", g0, inner, g0)).reshape(nb, niw, nd, 2, nd, 2), self.comm)
        return result

    def calibrate_taudiff_max(self, phase1pairnum=1000, phase2taugrid=100, phase2pairnum=1000, nprogress=100000):
        """Window calibration with the library's two calibration phases (`ctqmc_calibrate`), the protocol of
        `CtHybSolver.calibrate_taudiff_max` / `estimate_taudiff_max` (`impurity.py:1150-1195`): phase 1 collects accepted
        pair distances with one window, their 99.5 % quantile - averaged over the ranks - is the provisional window; phase 2
        histograms the accepted distances on it, and the distance at which an exponential fitted to a flavour's histogram
        has dropped to 1 % is that flavour's estimate; the largest one is returned."""
        import scipy.optimize as opt
        q = self.ctqmc

        def phase(first):
            q.init_counters()
            q.ctqmc_calibrate(first, phase1pairnum, phase2taugrid, phase2pairnum, nprogress)

        def over_ranks(x):
            return np.asarray(DistributedSample([np.asarray(x, dtype=np.float64)], self.comm).mean())

        phase(True)
        quantile = float(np.sort(q.accpairtau)[int(0.995 * phase1pairnum)])
        window = float(over_ranks([quantile]).ravel()[0])
        self.last_local.update(taudiffmax_phase1=quantile, taudiffmax_phase1_mean=window)
        q.set_taudiffmax(window)
        phase(False)
        histo = over_ranks(q.accpair)                                            # (band, spin, distance bin)
        centres = (np.arange(phase2taugrid) + 0.5) * window / phase2taugrid      # transform.tau_bins(..., "centre")

        def decay_length(acc):
            amp_rate = opt.leastsq(lambda p: acc - p[0] * np.exp(p[1] * centres), (acc[0], -1.0))[0]
            return np.log(1.0 - 0.99) / amp_rate[1]

        return float(max(decay_length(acc) for acc in histo.reshape(-1, phase2taugrid)))
