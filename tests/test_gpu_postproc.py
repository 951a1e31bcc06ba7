"""GPU (-m gpu): the solver-level chain G(l) -> G(iw) -> Sigma(iw) (jackknife over accumulator bins) on sampled data.

north_star: "Full-run observables (G(l), occupations, Sigma(iw)) must agree within stated Monte Carlo error bars."
The reference side is the reference's own long runs (8 seeds, tests/golden/stat_*.npz) pushed through the same
post-processing, whose arithmetic is pinned against the reference's Python modules in tests/test_postproc.py.
"""
import numpy as np
import pytest

import util
from w2dynamics_amd import postproc, synth
from w2dynamics_amd.solver import HipCtHybSolver

pytestmark = pytest.mark.gpu


def _solve(problem, over, nwalk=512, wpb=16, seed=77):
    s = HipCtHybSolver(over, seed, nwalkers=nwalk, walkers_per_bin=wpb)
    s.set_problem(problem)
    r = s.solve(0)
    s.postprocessing(r)
    s.ctqmc.close()
    return r


def test_noninteracting_self_energy_vanishes():
    """U = 0: G(iw) = 1/(iw + mu - Delta(iw)) and Sigma = 0, both within the jackknife error bars."""
    p = synth.make_problem(1, 10.0, 0.0, mu=0.3, nftau=2001)
    over = dict(Nwarmups=5000, Nmeas=400, NCorr=30, TaudiffMax=-1.0, Nftau=2001, Ntau=100, NLegMax=40, Niw=12)
    r = _solve(p, over)
    niw = 24
    g0inv = p.g0inviw(niw)
    giw, gerr = r["giw"].mean(), r["giw"].stderr()
    exact = postproc.promote_diagonal(1.0 / postproc.extract_diagonal(g0inv))
    d = postproc.extract_diagonal
    z = np.abs(d(giw) - d(exact)) / (d(gerr) + 3e-4)      # 3e-4: truncation of the Legendre series at 40 coefficients
    assert z.max() < 5.0, z.max()
    siw, serr = r["siw"].mean(), r["siw"].stderr()
    assert r["siw"].ntotal == 32
    zs = np.abs(d(siw)) / (d(serr) + 3e-3)
    assert zs.max() < 5.0, (zs.max(), np.abs(d(siw)).max())
    assert np.abs(d(r["smom"].mean())).max() < 1e-12       # no interaction, no moments


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4"])
def test_self_energy_vs_reference_runs(name):
    """Sigma(iw) from the HIP run against Sigma(iw) from the reference's 8-seed mean G(l), within combined errors; its
    high-frequency limit against the moments from rho1 / rho2 (Hartree term U <n> for the half-filled model)."""
    g = util.load_golden("stat_%s.npz" % name)
    over = dict(g["overrides"])
    over.update(Nmeas=max(50, int(over["Nmeas"]) * 8 // 512 * 2), Nwarmups=int(over["Nwarmups"]) // 4, Niw=10)
    p = synth.config(g["config"], int(g["nftau"]))
    r = _solve(p, over, seed=2025)
    niw = 20
    g0d = postproc.extract_diagonal(p.g0inviw(niw))
    d = postproc.extract_diagonal
    siw, serr = d(r["siw"].mean()), d(r["siw"].stderr())
    # reference: mean G(l) and its error over seeds -> G(iw) -> Sigma, error by linear propagation through 1/G
    nleg = int(over["NLegMax"])
    gref = postproc.extract_diagonal(postproc.get_giw({"gleg": g["gleg_mean"]}, niw, "legendre", nleg))
    tf = np.abs(postproc.leg2mat(nleg, niw))
    gref_err = np.sqrt(np.tensordot(np.asarray(g["gleg_err"]) ** 2, tf ** 2, (-1, -1))).transpose(2, 0, 1)
    sref = g0d - 1.0 / gref
    sref_err = gref_err / np.abs(gref) ** 2
    z = np.abs(siw - sref) / np.sqrt(serr ** 2 + sref_err ** 2)
    assert z.max() < 5.0, z.max()
    assert (z ** 2).mean() < 2.5, (z ** 2).mean()
    # Hartree limit: Re Sigma(iw_max) -> sigma_inf within errors and the 1/iw tail (a few per cent at |w| ~ 6/beta*pi)
    smom = d(r["smom"].mean())                         # (2, band, spin): sigma_inf, sigma_1
    w = postproc.matfreq(p.beta, "fermi", niw)
    tail = smom[0][None] + smom[1][None] / (1j * w[:, None, None])
    assert np.abs(siw[-1].real - tail[-1].real).max() < 5 * serr[-1].max() + 0.05
    occ = np.einsum("bsbs->bs", r["occ"].mean())
    assert np.abs(occ - 0.5).max() < 0.02              # half filling


def test_result_set_has_the_reference_keys_and_hdf5_layout():
    """f-2: `HipCtHybSolver.solve()` returns the quantities of `CtHybSolver.get_resultset` (impurity.py:1198-1283) under
    the reference's HDF5 names, and `hdf5_layout` gives what `write_impurity_result` (hdfout.py:369-397) puts below
    `ineq-NNN`: `<quantity>/value` and `<quantity>/error` with the shapes of the reference's module variables."""
    from w2dynamics_amd.solver import hdf5_layout
    p = synth.make_problem(2, 10.0, 2.0, j=0.25, nftau=501)
    over = dict(Nwarmups=2000, Nmeas=40, NCorr=20, TaudiffMax=4.0, Nftau=501, Ntau=50, NLegMax=20, Niw=8, MeasGiw=1, MeasSusz=1,
                AccumGtau=2, N4iwf=2, N4iwb=1, MeasG2iw=1, MeasG4iwPP=1)
    s = HipCtHybSolver(over, 3, nwalkers=64, walkers_per_bin=8)
    s.set_problem(p, compute_fourpoint=4)
    r = s.solve(0)
    s.postprocessing(r)
    always = {"gtau", "gtau-full", "rhist", "lhist", "hist", "ssts-states", "energies-eigenstates", "occbasis-mapping", "hist-sst",
              "hist-state", "sign-sst", "sign-state", "contrib-sst", "contrib-state", "occ", "gleg", "gleg-full", "accept-ins",
              "accept-ins4", "accept-rem", "accept-rem4", "accept-pair-tau", "accept-glob", "accept-shift", "accept-flavourchange",
              "sign", "time-warmup", "time-simulation", "time-sampling"}
    here = {"giw-meas", "g2iw", "densitymatrix", "rho1", "rho2", "g4iw", "g4iw-pp", "ntau-n0", "gtau-blocks"}
    derived = {"giw", "smom", "siw", "time-qmc"}                    # impurity.py:610-640, 208-225
    assert set(r) == always | here | derived
    h = hdf5_layout(r)
    nd, nst = 2, 16
    shapes = {"gtau": (nd, 2, 50), "gleg": (nd, 2, 20), "occ": (nd, 2, nd, 2), "hist": (nd, 2, 501), "giw-meas": (nd, 2, 16),
              "gtau-full": (nd, 2, nd, 2, 50), "gleg-full": (nd, 2, nd, 2, 20), "rho1": (nd, 2, nd, 2), "rho2": (nd, 2) * 4,
              "densitymatrix": (nst, nst), "hist-sst": (nst,), "sign-state": (nst,), "ntau-n0": (nd, 2, nd, 2, 51),
              "g4iw": (nd, 2, nd, 2, 4, 4, 3), "g4iw-pp": (nd, 2, nd, 2, 4, 4, 3), "g2iw": (nd, 2, 6, 6), "accept-ins": (16,),
              "sign": (), "accept-glob": (), "accept-pair-tau": (nd, 2, 50), "lhist": (100,), "occbasis-mapping": (nst, nd, 2),
              "giw": (16, nd, 2, nd, 2), "siw": (16, nd, 2, nd, 2), "smom": (2, nd, 2, nd, 2)}
    for key, shp in shapes.items():
        assert set(h[key]) == {"value", "error"}, key
        assert h[key]["value"].shape == shp and h[key]["error"].shape == shp, (key, h[key]["value"].shape)
    assert h["gtau-blocks"]["value"].shape[:3] == (nd, 2, 50)
    assert set(h["time-qmc"]) == {"value"} and h["time-qmc"]["value"].shape == ()
    assert r["gleg"].ntotal == 8 and np.all(h["gleg"]["error"][:, :, 0] > 0)
    assert abs(float(h["sign"]["value"])) == 1.0
    s.ctqmc.close()
