"""TEST INFRASTRUCTURE ONLY - drives the reference's own solver built by oracle/build_ref.sh.

`oracle/_ref/CTQMC*.so` is the f2py module of /root/reference/src/ctqmc_fortran (module `mctqmc`).  This file
is the checker's side of the parity tests, the generator of tests/golden/*.npz and the cpu_baseline
leg of bench.py (kind "reference").  Nothing in the product path (w2dynamics_amd/) may import it.

Call order follows `w2dyn/dmft/impurity.py:571-653` (CtHybSolver.solve).
"""
from __future__ import annotations

import contextlib
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")


def available() -> bool:
    return any(f.startswith("CTQMC") and f.endswith(".so") for f in (os.listdir(_REF) if os.path.isdir(_REF) else []))


def load():
    if _REF not in sys.path:
        sys.path.insert(0, _REF)
    import CTQMC  # noqa: the reference's f2py module
    return CTQMC.mctqmc


@contextlib.contextmanager
def quiet(enabled=True):
    """The Fortran code prints to the C-level stdout/stderr; silence both during a call."""
    if not enabled:
        yield
        return
    sys.stdout.flush(); sys.stderr.flush()
    saved = os.dup(1), os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(devnull, 1); os.dup2(devnull, 2)
        yield
    finally:
        os.dup2(saved[0], 1); os.dup2(saved[1], 2)
        os.close(devnull); os.close(saved[0]); os.close(saved[1])


class RefSolver:
    """Thin stateful wrapper: init -> (set config) -> warmup -> measure -> read results."""

    def __init__(self, problem, pstring, seed, fourpnt=0, silent=True):
        self.q = load()
        self.silent = silent
        q = self.q
        with quiet(silent):
            q.init_rnd(int(seed))
            q.fancy_prog = False
            q.init_paras(pstring)
            q.fourpnt = fourpnt
            q.simid = 0
            q.init_solver(problem.umatrix, problem.ftau, problem.muimp, problem.screening, 1)
        self.alive = True

    def set_taudiffmax(self, x):
        with quiet(self.silent):
            self.q.set_taudiffmax(float(x))

    def warmup(self, n):
        with quiet(self.silent):
            self.q.ctqmc_warmup(int(n))

    def measure(self, isector=1, icomponent=1):
        with quiet(self.silent):
            self.q.ctqmc_measure(int(isector), int(icomponent))

    def worm_warmup(self, n, isector, icomponent):
        with quiet(self.silent):
            self.q.ctqmc_worm_warmup(int(n), int(isector), int(icomponent))

    def set_wormeta(self, eta):
        self.q.wormeta[:] = float(eta)

    def init_counters(self):
        self.q.init_counters()

    def calibrate(self, list_, phase1pairnum, phase2taugrid, phase2pairnum, nprogress):
        with quiet(self.silent):
            self.q.ctqmc_calibrate(bool(list_), int(phase1pairnum), int(phase2taugrid), int(phase2pairnum), int(nprogress))

    def count_steps(self, isector, icomponent, nfix):
        sw = np.array(0, dtype=np.int64); sz = np.array(0, dtype=np.int64)
        with quiet(self.silent):
            self.q.count_steps(int(isector), int(icomponent), int(nfix), sw, sz)
        return int(sw), int(sz)

    def get_config(self):
        q = self.q
        n, osst, ost = q.get_mc_config_scalars()
        if n > 0:
            taus, orbs, spins, cas, hashybs, osst, ost = q.get_mc_config(n)
        else:
            taus = np.zeros(0); orbs = spins = cas = hashybs = np.zeros(0, dtype=np.int32)
        return dict(taus=np.array(taus), orbs=np.array(orbs), spins=np.array(spins), cas=np.array(cas),
                    hashybs=np.array(hashybs), outer_sst=int(osst), outer_state=int(ost))

    def set_config(self, cfg):
        q = self.q
        n = len(cfg["taus"])
        with quiet(self.silent):
            if n == 0:
                q.set_empty_mc_config(cfg["outer_sst"], cfg["outer_state"])
            else:
                q.set_mc_config(n, cfg["taus"], cfg["orbs"], cfg["spins"], cfg["cas"], cfg["hashybs"],
                                cfg["outer_sst"], cfg["outer_state"])

    def results(self, names=("gtau", "gleg", "occ", "histo", "mean_sign", "accadd", "accrem", "accglob",
                             "accshift", "densitymatrix", "rho1", "rho2", "outersuperstatehisto",
                             "statessuperstates", "eigenstatesenergies", "occbasismapping", "accpair",
                             "time_warmup", "time_sim", "time_sim_steps", "cntsampling", "giw")):
        out = {}
        for n in names:
            try:
                v = getattr(self.q, n)
            except Exception:
                continue
            if v is None:
                continue
            out[n] = np.array(v).copy()
        return out

    def close(self):
        if self.alive:
            with quiet(self.silent):
                self.q.dest_ctqmc()
                self.q.dest_paras()
            self.alive = False
