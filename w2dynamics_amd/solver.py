"""Host-side counterpart of the reference's solver wrapper for the Z-space path.

`HipCtHybSolver` reproduces the call order of `CtHybSolver.set_problem` / `solve` (`w2dyn/dmft/impurity.py:497-534,
571-653`) on top of the `mctqmc`-shaped HIP front end, one process per GPU:

  init_rnd(seed) -> init_paras(pstring) -> fourpnt -> init_solver(U, ftau, -muimp, screening, iter+1)
  -> set_taudiffmax -> ctqmc_warmup -> ctqmc_measure(1,1) -> result pickup -> DistributedSample wrapping
  -> dest_ctqmc / dest_paras

Chains are sharded over ranks without any collective in the sampling path: rank r runs chains
[r*nwalkers, (r+1)*nwalkers) of one global stream family (reference: seed + rank, impurity.py:381).  The only
exchange step is the final reduction (`statistics.DistributedSample`, RCCL all-reduce).
"""
from __future__ import annotations

import time

import numpy as np

from . import params as _params
from .mctqmc import MCTQMC
from .statistics import DistributedSample, TorchComm

RESULT_KEYS_BINNED = ("gleg", "gtau", "occ", "histo")


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
        result = {}
        for key in RESULT_KEYS_BINNED:
            result[key] = DistributedSample(getattr(q, key + "_bins"), self.comm)
        if self.qmc["MeasGiw"]:
            result["giw-meas"] = DistributedSample(q.giw_bins, self.comm)
        result["sign"] = DistributedSample(np.array([q.mean_sign]), self.comm)
        result["accept-ins"] = DistributedSample(np.array([q.accadd]), self.comm)
        result["accept-rem"] = DistributedSample(np.array([q.accrem]), self.comm)
        result["rho1"] = DistributedSample(np.array([q.rho1]), self.comm)
        result["rho2"] = DistributedSample(np.array([q.rho2]), self.comm)
        result["time-qmc"] = time.perf_counter() - t0
        q.dest_ctqmc()
        q.dest_paras()
        return result

    def calibrate_taudiff_max(self, phase1pairnum=1000, phase2taugrid=100, phase2pairnum=1000, nprogress=100000):
        """Two-phase window calibration in the call order of `CtHybSolver.calibrate_taudiff_max`
        (`impurity.py:1150-1172`): the 99.5th percentile of the accepted pair distances sets a provisional window,
        then the distance histogram on that window is fitted (`estimate_taudiff_max`)."""
        q = self.ctqmc
        q.init_counters()
        q.ctqmc_calibrate(True, phase1pairnum, phase2taugrid, phase2pairnum, nprogress)
        acctaulist = q.accpairtau
        acctaulist.sort()
        taudiffmax = DistributedSample([np.array((acctaulist[int(0.995 * phase1pairnum)],))], self.comm).mean()
        taudiffmax = float(np.asarray(taudiffmax).ravel()[0])
        q.set_taudiffmax(taudiffmax)
        q.init_counters()
        q.ctqmc_calibrate(False, phase1pairnum, phase2taugrid, phase2pairnum, nprogress)
        return self.estimate_taudiff_max(self.comm, q.accpair, taudiffmax, phase2taugrid)

    @staticmethod
    def estimate_taudiff_max(comm, accept_pair, taudiffmax, phase2taugrid):
        """Distance at which the fitted exponential decay of the pair acceptance has dropped to 1 %, largest over the
        flavours (`impurity.py:1174-1195`)."""
        import scipy.optimize as opt
        accept_pair = np.asarray(DistributedSample([np.asarray(accept_pair, dtype=np.float64)], comm).mean())
        taus = (np.arange(phase2taugrid) + 0.5) * taudiffmax / phase2taugrid          # transform.tau_bins(..., "centre")
        estimates = np.empty(accept_pair.shape[:2])
        for b in range(accept_pair.shape[0]):
            for s in range(accept_pair.shape[1]):
                fit = opt.leastsq(lambda p, t, acc: acc - p[0] * np.exp(p[1] * t), (accept_pair[b, s, 0], -1), (taus, accept_pair[b, s]))
                estimates[b, s] = np.log(1 - 0.99) / fit[0][1]
        return float(np.amax(estimates))
