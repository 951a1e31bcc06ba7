"""Post-processing of the solver results on the host, as the reference's caller does it after `ctqmc_measure`:

    G(l) -> G(iw)                 `CtHybSolver.get_giw`, FTType = legendre   (w2dyn/dmft/impurity.py:1078-1082,
                                   w2dyn/auxiliaries/transform.py:221-254 leg2mat, 480-522 transform)
    G(iw) from the direct sum     FTType = none: giw = -giw-meas             (impurity.py:1064-1065)
    (band, spin, iw) -> (iw, band, spin, band, spin)   `lattice_convention`  (impurity.py:35-42, orbspin.py:41-58)
    Sigma(iw) = G0^-1 - G^-1      Dyson equation on jackknife pseudovalues   (impurity.py:249-303,
                                   w2dyn/auxiliaries/statistics.py:251-339 DistributedJackknife)
The functions keep the names the reference's caller uses; the formulations are this build's own (mirrored spherical
Bessel matrix, index-contraction form of the moments, vectorised jackknife) and are pinned on outputs of the reference's
modules in tests/golden/postproc.npz.
    Sigma moments from rho1/rho2  `get_siw_mom`                              (w2dyn/auxiliaries/postprocessing.py:353-387)

Everything here is numpy on a few hundred KB; the sample axis is the accumulator bins of all ranks
(`statistics.DistributedSample`), so the one exchange step stays the all-reduce inside `DistributedSample`.
"""
from __future__ import annotations

import numpy as np
import scipy.special as sps

from .statistics import DistributedSample


def matfreq(beta, mattype, matorder):
    """Matsubara frequencies symmetric around zero: `matorder` fermionic (even count) or bosonic (odd count) ones
    (the convention of transform.py:88-112)."""
    kinds = {"fermi": 1, "bose": 0}
    if mattype not in kinds:
        raise ValueError("mattype is 'fermi' or 'bose', not %r" % (mattype,))
    odd = kinds[mattype]
    if (matorder + odd) % 2 != 1:
        raise ValueError("%d %s frequencies cannot be symmetric around zero (need an %s count)"
                         % (matorder, mattype, "even" if odd else "odd"))
    first = -(matorder // 2)
    return (np.pi / beta) * (2 * np.arange(first, first + matorder) + odd)


def leg2mat(legorder, matorder):
    """T[n, l] = (-1)^n i^(l+1) sqrt(2l+1) j_l((2n+1) pi / 2) (Boehnke et al. 2011, eq. E2; the matrix of
    transform.py:221-254).  Built for the positive frequencies from scipy's spherical Bessel function and mirrored:
    T[-n-1, l] = conj(T[n, l]) because G(tau) is real."""
    if matorder % 2:
        raise ValueError("leg2mat needs an even number of fermionic frequencies")
    half = matorder // 2
    n = np.arange(half)
    l = np.arange(int(legorder))
    pos = sps.spherical_jn(l[None, :], (n[:, None] + 0.5) * np.pi) * np.sqrt(2.0 * l + 1.0) * 1j ** (l + 1)
    pos[1::2] *= -1.0
    return np.concatenate([pos[::-1].conj(), pos])


def transform_last(tfmat, data):
    """Apply `tfmat` (new, old) to the last axis of `data`; if the lengths differ, the longer of the two is cut to the
    leading entries of the shorter (what `transform(..., Truncate.tofirst)` of transform.py:480-512 does)."""
    tfmat = np.asarray(tfmat); data = np.asarray(data)
    n = min(tfmat.shape[-1], data.shape[-1])
    return np.einsum("...l,nl->...n", data[..., :n], tfmat[..., :n])


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
    """Delete-one jackknife over a `DistributedSample` (the statistics of statistics.py:251-339): replicas
    x'_i = xbar + (xbar - x_i)/(n-1) and bias-corrected pseudovalues y_i = f(xbar) + (n-1) (f(xbar) - f(x'_i)),
    whose mean and standard error are the jackknife estimate of f and its error.  Needs one all-reduce (the sample's
    cached moments); the replicas of this rank are formed in one vector operation."""

    def __init__(self, sample):
        sample._need_observations("DistributedJackknife")
        self.comm = sample.comm
        self.nlocal, self.ntotal = sample.nlocal, sample.ntotal
        self._mean = sample.mean()
        self.leaveoneout = None
        if self.ntotal > 1:
            self.leaveoneout = self._mean + (self._mean - np.asarray(sample.local)) / (self.ntotal - 1)

    @property
    def mpi_comm(self):
        return self.comm

    def mean(self):
        return self._mean

    def transform(self, f, batched=False):
        """Pseudovalues of f as a DistributedSample.  `batched`: f accepts the whole (nlocal, ...) replica block."""
        centre = np.asarray(f(self._mean))
        if self.leaveoneout is None:                     # a single observation: nothing to leave out
            return DistributedSample(centre[None], self.comm, self.ntotal)
        if batched:
            frep = np.asarray(f(self.leaveoneout))
        else:
            frep = np.stack([np.asarray(f(rep)) for rep in self.leaveoneout]) if self.nlocal else np.empty((0,) + centre.shape, centre.dtype)
        return DistributedSample(centre + (self.ntotal - 1) * (centre - frep), self.comm, self.ntotal)


def siw_dyson(giw_sample, g0inviw, diagonal_only=True):
    """Sigma(iw) = G0^-1(iw) - G^-1(iw) on jackknife pseudovalues of G(iw) (impurity.py:249-261, 302-303).
    `giw_sample`: DistributedSample of (iw, band, spin, band, spin) arrays; returns a DistributedSample."""
    g0inviw = np.asarray(g0inviw)
    nfl = g0inviw.shape[-2] * g0inviw.shape[-1]
    if diagonal_only:
        g0d = extract_diagonal(g0inviw)

        def dyson(giw):                                  # works on one replica and on a stack of them
            return promote_diagonal(g0d - 1.0 / extract_diagonal(giw))
    else:
        def dyson(giw):
            return g0inviw - np.linalg.inv(giw.reshape(giw.shape[:-4] + (nfl, nfl))).reshape(giw.shape)
    return DistributedJackknife(giw_sample).transform(dyson, batched=True)


def get_siw_mom(u_matrix, rho1, rho2):
    """High-frequency expansion Sigma(iw -> inf) = s0 + s1/(iw) from <c+c> and <c+c+cc> (Wang, Dang, Millis, PRB 84,
    073104; the quantity of postprocessing.py:353-387), written as explicit index contractions with the
    antisymmetrised interaction A_abcd = -(U_abcd - U_bacd - U_abdc + U_badc)/2:

        s0_ad = A_abcd <c+_b c_c>
        s1_ah = s0_ab s0_bh - A_abcd A_efbh <ef|cd>/4 + A_abcd A_dfgh <bf|cg> + A_abcd A_cdeh <c+_b c_e>/2

    Returns the flavour diagonals as (band, spin, 2)."""
    u = np.asarray(u_matrix)
    nfl = u.shape[0]
    r1 = np.asarray(rho1).reshape((nfl,) * 2)
    r2 = np.asarray(rho2).reshape((nfl,) * 4)
    A = -0.5 * (u - np.einsum("bacd->abcd", u) - np.einsum("abdc->abcd", u) + np.einsum("badc->abcd", u))
    s0 = np.einsum("abcd,bc->ad", A, r1)
    s1 = (s0 @ s0
          - 0.25 * np.einsum("abcd,efbh,efcd->ah", A, A, r2, optimize=True)
          + np.einsum("abcd,dfgh,bfcg->ah", A, A, r2, optimize=True)
          + 0.5 * np.einsum("abcd,cdeh,be->ah", A, A, r1, optimize=True))
    return np.stack([s0.diagonal(), s1.diagonal()], axis=-1).reshape(nfl // 2, 2, 2)


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
