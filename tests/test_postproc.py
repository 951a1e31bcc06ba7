"""CPU: host-side post-processing (w2dynamics_amd/postproc.py) against outputs of the reference's own Python modules
(tests/golden/postproc.npz, made by tests/golden/gen_postproc_golden.py) and against defining properties where the
reference module cannot be imported here (its statistics module needs mpi4py: the jackknife is "parity pinned by
properties")."""
import os

import numpy as np
import pytest

from w2dynamics_amd import postproc, synth
from w2dynamics_amd.statistics import DistributedSample

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postproc.npz"))
TOL = 1e-12


def close(a, b, tol=TOL):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * max(np.abs(np.asarray(b)).max(), 1.0)


def test_matsubara_frequencies_and_legendre_matrix_match_reference():
    assert close(postproc.matfreq(7.5, "fermi", 20), G["matfreq_fermi_7.5_20"])
    assert close(postproc.matfreq(7.5, "bose", 11), G["matfreq_bose_7.5_11"])
    assert close(postproc.leg2mat(30, 20), G["leg2mat_30_20"])
    with pytest.raises(ValueError):
        postproc.matfreq(1.0, "fermi", 3)


def test_legendre_to_matsubara_with_truncation_matches_reference():
    assert close(postproc.giw_from_legendre(G["gleg"], 20, nleg=30), G["giw_nleg30"])      # matrix longer than the data
    assert close(postproc.giw_from_legendre(G["gleg"], 20, nleg=16), G["giw_nleg16"])      # data longer than the matrix
    lat = postproc.get_giw({"gleg": G["gleg"]}, 20, "legendre", nleg=30)
    assert lat.shape == (20, 2, 2, 2, 2) and close(lat, G["giw_lattice"])
    assert close(postproc.extract_diagonal(lat), G["giw_lattice_diag"])
    assert close(postproc.get_giw({"giw-meas": -G["giw_nleg30"]}, 20, "none"), G["giw_lattice"])


def test_self_energy_moments_match_reference():
    assert close(postproc.get_siw_mom(G["smom_U"], G["smom_rho1"], G["smom_rho2"]), G["smom"])


def test_jackknife_properties():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(7, 5)) + 3.0
    s = DistributedSample(x)
    jk = postproc.DistributedJackknife(s)
    assert close(jk.mean(), x.mean(0))
    assert close(jk.leaveoneout, (x.sum(0)[None] - x) / 6.0)
    # a linear map returns the transformed samples themselves
    lin = jk.transform(lambda v: 2.0 * v - 1.0)
    assert close(lin.local, 2.0 * x - 1.0) and lin.ntotal == 7
    # y_i = n f(xbar) - (n-1) f(x'_i); the mean of the pseudovalues removes the 1/n bias of f(xbar) = xbar^2 exactly
    sq = jk.transform(lambda v: v * v)
    assert close(sq.local, 7 * x.mean(0) ** 2 - 6 * ((x.sum(0)[None] - x) / 6.0) ** 2)
    unbiased = x.mean(0) ** 2 - x.var(0, ddof=1) / 7
    assert close(sq.mean(), unbiased, 1e-10)
    # fewer than two samples: the transformed mean (statistics.py:328-329)
    one = postproc.DistributedJackknife(DistributedSample(x[:1])).transform(lambda v: v * v)
    assert one.nlocal == 1 and close(one.local[0], x[0] ** 2)


def test_dyson_equation_on_exact_input():
    """G = 1/(iw + mu - Delta - Sigma) with a known Sigma is returned by siw_dyson, diagonal and matrix form."""
    p = synth.make_problem(2, 10.0, 2.0, j=0.25, nftau=101)
    niw = 16
    g0inv = p.g0inviw(niw)
    w = postproc.matfreq(p.beta, "fermi", niw)
    sig = postproc.promote_diagonal((1.0 + 0.3 / (1j * w))[:, None, None] * np.ones((1, 2, 2)))
    giw = np.linalg.inv((g0inv - sig).reshape(niw, 4, 4)).reshape(g0inv.shape)
    sample = DistributedSample(np.array([giw, giw, giw]))
    for diag in (True, False):
        siw = postproc.siw_dyson(sample, g0inv, diagonal_only=diag)
        assert siw.local.shape == (3,) + g0inv.shape
        assert close(siw.mean(), sig, 1e-10)


def test_bethe_hybridisation_in_frequency_matches_ftau():
    """Delta(iw) used for G0 is the Fourier transform of the F(tau) handed to the solver: F(tau) = -Delta(beta - tau)
    (selfcons.py:477-492), i.e. Delta(iw) = -int_0^beta e^{iw tau} Delta(tau), Delta(tau) = -F(beta - tau)."""
    beta, n = 10.0, 20001
    f = synth.bethe_ftau(beta, n)
    tau = np.linspace(0.0, beta, n)
    w = postproc.matfreq(beta, "fermi", 8)
    delta_tau = -f[::-1]                                  # Delta(tau) = -F(beta - tau)
    wts = np.full(n, beta / (n - 1)); wts[0] *= 0.5; wts[-1] *= 0.5
    num = (np.exp(1j * w[:, None] * tau[None]) * delta_tau[None] * wts[None]).sum(1)
    assert np.abs(num - postproc.bethe_fiw(beta, 8)).max() < 1e-5
