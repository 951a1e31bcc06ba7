"""Post-processing of the solver results on the host, as the reference's caller does it after `ctqmc_measure`:

    G(l) -> G(iw)                 `CtHybSolver.get_giw`, FTType = legendre   (w2dyn/dmft/impurity.py:1078-1082,
                                   w2dyn/auxiliaries/transform.py:221-254 leg2mat, 480-522 transform)
    G(iw) from the direct sum     FTType = none: giw = -giw-meas             (impurity.py:1064-1065)
    (band, spin, iw) -> (iw, band, spin, band, spin)   `lattice_convention`  (impurity.py:35-42, orbspin.py:41-58)
    Sigma(iw) = G0^-1 - G^-1      Dyson equation on jackknife pseudovalues   (impurity.py:249-303,
                                   w2dyn/auxiliaries/statistics.py:251-339 DistributedJackknife)
    Sigma moments from rho1/rho2  `get_siw_mom`                              (w2dyn/auxiliaries/postprocessing.py:353-387)

Everything here is numpy on a few hundred KB; the sample axis is the accumulator bins of all ranks
(`statistics.DistributedSample`), so the one exchange step stays the all-reduce inside `DistributedSample`.
"""
from __future__ import annotations

import numpy as np
import scipy.special as sps

from .statistics import DistributedSample


def matfreq(beta, mattype, matorder):
    """Matsubara frequencies symmetric around zero (transform.py:88-112)."""
    half = int(matorder / 2)
    if mattype == "fermi":
        if matorder % 2:
            raise ValueError("Number of Fermi frequencies must be even")
        return (2 * np.pi / beta) * (np.arange(-half, half) + 0.5)
    if mattype == "bose":
        if matorder % 2 != 1:
            raise ValueError("Number of Bose frequencies must be odd")
        return (2 * np.pi / beta) * np.arange(-half, half + 1)
    raise ValueError("mattype must be either 'fermi' or 'bose'")


def leg2mat(legorder, matorder):
    """T[n, l] = (-1)^n i^(l+1) sqrt(2l+1) j_l((2n+1) pi / 2) (Boehnke et al. 2011, E2; transform.py:221-254), with the
    spherical Bessel function written through J_(l+1/2) as the reference does."""
    half = int(matorder / 2)
    ws = matfreq(2, "fermi", matorder) + 0j
    ls = np.arange(int(legorder))
    nfac = (-1.0) ** np.arange(-half, half) * np.sqrt(np.pi / (2 * ws))
    lfac = (1j) ** (ls + 1) * np.sqrt(2 * ls + 1)
    return nfac[:, None] * lfac * sps.jv(ls + 0.5, ws[:, None])


def transform_last(tfmat, data):
    """`transform(tfmat, data, onmismatch=Truncate.tofirst)` for the last axis (transform.py:480-512): the longer of
    the two (matrix columns, data axis) is cut to the first entries of the shorter."""
    tfmat = np.asarray(tfmat); data = np.asarray(data)
    n = min(tfmat.shape[-1], data.shape[-1])
    return np.tensordot(data[..., :n], tfmat[..., :n], (-1, -1))


def promote_diagonal(diag):
    """(..., band, spin) -> (..., band, spin, band, spin) with the input on the flavour diagonal (orbspin.py:41-49)."""
    diag = np.asarray(diag)
    shp = diag.shape
    nfl = shp[-2] * shp[-1]
    flat = diag.reshape(-1, nfl)
    full = np.zeros(flat.shape + (nfl,), diag.dtype)
    fl = np.arange(nfl)
    full[:, fl, fl] = flat
    return full.reshape(shp + shp[-2:])


def extract_diagonal(arr):
    """(..., band, spin, band, spin) -> (..., band, spin) (orbspin.py:51-58)."""
    arr = np.asarray(arr)
    shp = arr.shape
    nfl = shp[-2] * shp[-1]
    return arr.reshape(-1, nfl, nfl).diagonal(0, -2, -1).copy().reshape(shp[:-2])


def lattice_convention(qtty):
    """(band, spin, x) -> (x, band, spin, band, spin) (impurity.py:35-42)."""
    return promote_diagonal(np.asarray(qtty).transpose(2, 0, 1))


def giw_from_legendre(gleg, niw, nleg=None):
    """G(iw) on `niw` fermionic frequencies from Legendre coefficients (band, spin, l): -leg2mat . G(l)
    (impurity.py:1078-1082; the sign undoes the solver's positive G convention)."""
    gleg = np.asarray(gleg)
    nleg = gleg.shape[-1] if nleg is None else nleg
    return transform_last(-leg2mat(nleg, niw), gleg)


def get_giw(result, niw, fttype="legendre", nleg=None):
    """`CtHybSolver.get_giw` for a band/spin-diagonal solver result, returned in lattice convention."""
    if fttype == "none":
        giw = -np.asarray(result["giw-meas"])
    elif fttype == "legendre":
        giw = giw_from_legendre(result["gleg"], niw, nleg)
    else:
        raise ValueError("Invalid transform type: `%s'" % fttype)
    return lattice_convention(giw)


class DistributedJackknife:
    """Leave-one-out statistics x'_i = (n xbar - x_i)/(n-1) and bias-corrected pseudovalues
    y'_i = n f(xbar) - (n-1) f(x'_i) of a `DistributedSample` (statistics.py:251-339)."""

    def __init__(self, sample):
        self.mpi_comm = sample.mpi_comm
        self.nlocal = sample.nlocal
        self.ntotal = sample.ntotal
        ssum = sample.sum()
        if self.ntotal < 2:
            self.leaveoneout = None
        else:
            self.leaveoneout = (ssum - np.asarray(sample.local)) / (self.ntotal - 1)
        self._mean = ssum / self.ntotal

    def mean(self):
        return self._mean

    def transform(self, f):
        if self.ntotal < 2:
            return DistributedSample([f(self._mean)], self.mpi_comm, self.ntotal)
        ymean = f(self._mean)
        yi = np.empty((self.nlocal,) + ymean.shape, ymean.dtype)
        yi[...] = self.ntotal * ymean
        for i in range(self.nlocal):
            yi[i] -= (self.ntotal - 1) * f(self.leaveoneout[i])
        return DistributedSample(yi, self.mpi_comm, self.ntotal)


def siw_dyson(giw_sample, g0inviw, diagonal_only=True):
    """Sigma(iw) = G0^-1(iw) - G^-1(iw) on jackknife pseudovalues of G(iw) (impurity.py:249-261, 302-303).
    `giw_sample`: DistributedSample of (iw, band, spin, band, spin) arrays; returns a DistributedSample."""
    g0inviw = np.asarray(g0inviw)
    if diagonal_only:
        g0d = extract_diagonal(g0inviw)

        def get_siw(giw):
            return promote_diagonal(g0d - 1.0 / extract_diagonal(giw))
    else:
        def get_siw(giw):
            shp = giw.shape
            nfl = shp[1] * shp[2]
            return g0inviw - np.linalg.inv(giw.reshape(-1, nfl, nfl)).reshape(shp)
    return DistributedJackknife(giw_sample).transform(get_siw)


def get_siw_mom(u_matrix, rho1, rho2):
    """Sigma(iw -> inf) = sigma_inf + sigma_1/(iw) from the one- and two-particle density matrices (Wang, Dang, Millis,
    PRB 84, 073104; postprocessing.py:353-387).  Returns (band, spin, 2)."""
    u_matrix = np.asarray(u_matrix)
    nfl = u_matrix.shape[0]
    nb = nfl // 2
    r1 = np.asarray(rho1).reshape(nfl, nfl)
    r2 = np.asarray(rho2).reshape(nfl, nfl, nfl, nfl)
    ua = -0.5 * (u_matrix - u_matrix.swapaxes(0, 1) - u_matrix.swapaxes(2, 3) + u_matrix.swapaxes(0, 1).swapaxes(2, 3))
    sigma_inf = np.tensordot(ua, r1, axes=([1, 2], [0, 1]))
    L = -0.5 * np.tensordot(ua, ua, axes=([2, 3], [0, 1]))
    first = 0.25 * np.tensordot(ua, ua, axes=(1, 2))
    second = -np.tensordot(ua, ua, axes=(3, 0))
    k1 = np.tensordot(first, r2, axes=([1, 2, 3, 4], [2, 3, 0, 1]))
    k2 = np.tensordot(second, r2, axes=([1, 2, 3, 4], [0, 2, 1, 3]))
    sigma_1 = -(k1 + k2 + np.tensordot(L, r1, axes=([1, 2], [0, 1])) - np.tensordot(sigma_inf, sigma_inf, axes=(1, 0)))
    smoms = np.asarray([sigma_inf, sigma_1])
    smoms = np.diagonal(smoms, axis1=1, axis2=2).copy()
    return smoms.transpose().reshape(nb, 2, -1)


def bethe_fiw(beta, niw, D=1.0):
    """Hybridisation function Delta(iw) = (D/2)^2 g(iw) of the Bethe bath of `synth.bethe_ftau`,
    g(iw) = 2 (iw - i sgn(w) sqrt(w^2 + D^2)) / D^2, on `niw` fermionic frequencies."""
    w = matfreq(beta, "fermi", niw)
    return 0.5j * (w - np.sign(w) * np.sqrt(w * w + D * D))


def g0inviw_diagonal(beta, niw, mu, fiw):
    """G0^-1(iw) = iw + mu - Delta(iw) for band/spin-diagonal mu (band, spin) and Delta (iw, band, spin), in lattice
    convention (iw, band, spin, band, spin)."""
    w = matfreq(beta, "fermi", niw)
    mu = np.asarray(mu, dtype=float)
    return promote_diagonal(1j * w[:, None, None] + mu[None] - np.asarray(fiw))
