"""Deterministic synthetic impurity problems: Bethe-lattice bath, density-density U.

Conventions follow the reference's caller side:
  * F(tau) > 0 on `Nftau` equidistant points including both ends, F(tau) = -Delta(beta - tau)
    (`w2dyn/dmft/selfcons.py:477-492`), array layout (Nd, 2, Nd, 2, Nftau) as passed to `init_solver`
    after the transpose of `w2dyn/dmft/impurity.py:526`;
  * `muimp` is passed *negated* (`impurity.py:529`), i.e. init_solver receives -mu on the diagonal;
  * U tensor indices [b1,s1,b2,s2,b3,s3,b4,s4] filled as in `w2dyn/dmft/interaction.py:182-189`
    (Density) and reshaped to (2Nd)^4 with flavour index 2*band+spin (`impurity.py:530-532`).
"""
from __future__ import annotations

import numpy as np


def bethe_ftau(beta: float, nftau: int, D: float = 1.0, nquad: int = 4000) -> np.ndarray:
    """F(tau) = int de (D/2)^2 rho(e) exp(-(beta-tau) e)/(1+exp(-beta e)), rho = semicircle of half width D.

    Gauss-Chebyshev (second kind) quadrature, numerically stable for both signs of e."""
    i = np.arange(1, nquad + 1)
    theta = i * np.pi / (nquad + 1)
    eps = D * np.cos(theta)
    w = 2.0 / (nquad + 1) * np.sin(theta) ** 2
    tau = np.linspace(0.0, beta, nftau)
    e = eps[None, :]
    t = tau[:, None]
    pos = np.exp(-(beta - t) * np.abs(e)) / (1.0 + np.exp(-beta * np.abs(e)))   # e >= 0
    neg = np.exp(-t * np.abs(e)) / (1.0 + np.exp(-beta * np.abs(e)))            # e < 0
    kern = np.where(e >= 0, pos, neg)
    return (D / 2.0) ** 2 * (kern * w[None, :]).sum(axis=1)


def density_umatrix(nd: int, u: float, v: float = 0.0, j: float = 0.0) -> np.ndarray:
    """Density-density U tensor (interaction.py:182-189) reshaped to (2Nd,)*4."""
    U = np.zeros((nd, 2, nd, 2, nd, 2, nd, 2))
    for b1 in range(nd):
        for s in (0, 1):
            U[b1, s, b1, 1 - s, b1, s, b1, 1 - s] = u
        for b3 in range(nd):
            if b3 == b1:
                continue
            for s in (0, 1):
                for sp in (0, 1):
                    U[b1, s, b3, sp, b1, s, b3, sp] = v
                U[b1, s, b3, s, b3, s, b1, s] = j
    return U.reshape((2 * nd,) * 4)


def kanamori_umatrix(nd: int, u: float, v: float = 0.0, j: float = 0.0) -> np.ndarray:
    """Kanamori U tensor (interaction.py:232-241): density-density part plus pair hopping and spin flip."""
    U = density_umatrix(nd, u, v, j).reshape((nd, 2) * 4)
    for b1 in range(nd):
        for b2 in range(nd):
            if b1 == b2:
                continue
            for s in (0, 1):
                U[b1, s, b1, 1 - s, b2, s, b2, 1 - s] = j      # pair hopping
                U[b1, s, b2, 1 - s, b2, s, b1, 1 - s] = j      # spin flip
    return U.reshape((2 * nd,) * 4)


class Problem:
    """One impurity problem in the layout `init_solver` expects."""

    def __init__(self, nd, beta, umatrix, ftau_full, muimp_full_negated, hamiltonian="Density",
                 quantum_numbers="Nt Szt Azt", bath_scale=None, bath_D=1.0):
        self.nd = nd
        self.beta = float(beta)
        self.umatrix = np.ascontiguousarray(umatrix, dtype=np.float64)
        self.ftau = np.ascontiguousarray(ftau_full, dtype=np.float64)
        self.muimp = np.ascontiguousarray(muimp_full_negated, dtype=np.float64)
        self.screening = np.zeros_like(self.ftau)
        self.hamiltonian = hamiltonian
        self.quantum_numbers = quantum_numbers
        self.bath_scale = np.ones(nd) if bath_scale is None else np.asarray(bath_scale, dtype=float)   # F_b = scale_b * F_Bethe
        self.bath_D = bath_D

    def g0inviw(self, niw):
        """Inverse Weiss field G0^-1(iw) = iw + mu - Delta(iw) of this problem on `niw` fermionic frequencies, in the
        caller's layout (iw, band, spin, band, spin) (`ImpurityProblem.g0inviw`, w2dyn/dmft/impurity.py:123)."""
        from . import postproc
        fiw = postproc.bethe_fiw(self.beta, niw, self.bath_D)[:, None, None] * self.bath_scale[None, :, None] * np.ones((1, 1, 2))
        mu = -np.einsum("bsbs->bs", self.muimp)          # init_solver receives -mu
        return postproc.g0inviw_diagonal(self.beta, niw, mu, fiw)

    @property
    def nftau(self):
        return self.ftau.shape[-1]


def make_problem(nd: int, beta: float, u: float, j: float = 0.0, v: float | None = None, mu: float | None = None,
                 D: float = 1.0, nftau: int = 5001, spin_split: float = 0.0, kanamori: bool = False,
                 quantum_numbers: str | None = None, hop: float = 0.0, fmix: float = 0.0) -> Problem:
    """Half-filled (by default) density-density problem on a Bethe bath.

    mu defaults to U/2 + (Nd-1)(V - J/2) (particle-hole symmetric point for this interaction).
    `spin_split` adds +-h to the two spin species so that tests can break the symmetry; `hop` adds a spin-conserving
    hopping between neighbouring orbitals to the local one-particle Hamiltonian (off-diagonal muimp, diagonal bath); `fmix`
    adds the same fraction of F(tau) between neighbouring orbitals of equal spin, slightly different in the two directions
    (off-diagonal hybridisation, for runs with offdiag = 1)."""
    if v is None:
        v = u - 2.0 * j
    if mu is None:
        mu = u / 2.0 + (nd - 1) * (v - j / 2.0)
    f = bethe_ftau(beta, nftau, D)
    ftau = np.zeros((nd, 2, nd, 2, nftau))
    muimp = np.zeros((nd, 2, nd, 2))
    scale = 1.0 + 0.05 * np.arange(nd)
    for b in range(nd):
        for s in (0, 1):
            ftau[b, s, b, s, :] = f * (1.0 + 0.05 * b)    # mildly orbital dependent: catches flavour mix-ups
            muimp[b, s, b, s] = -(mu + (0.5 - s) * 2.0 * spin_split)
    for b in range(nd - 1):
        for s in (0, 1):
            muimp[b, s, b + 1, s] = muimp[b + 1, s, b, s] = -hop
            if fmix:
                ftau[b, s, b + 1, s, :] = fmix * f
                ftau[b + 1, s, b, s, :] = fmix * f
    if kanamori:
        return Problem(nd, beta, kanamori_umatrix(nd, u, v, j), ftau, muimp, hamiltonian="Kanamori",
                       quantum_numbers=quantum_numbers or "Nt Szt Qzt", bath_scale=scale, bath_D=D)
    return Problem(nd, beta, density_umatrix(nd, u, v, j), ftau, muimp, quantum_numbers=quantum_numbers or "Nt Szt Azt",
                   bath_scale=scale, bath_D=D)


# the BASELINE.json configurations (C5's worm part is not covered yet) and small variants for tests
def config(name: str, nftau: int = 5001) -> Problem:
    name = name.upper()
    if name == "C1":
        return make_problem(1, 10.0, 2.0, nftau=nftau)
    if name == "C2":
        return make_problem(1, 100.0, 2.0, nftau=nftau)
    if name == "C2L":      # C2 at half the temperature: orders of 30-40 per flavour (estimator kernels: more than one register tile)
        return make_problem(1, 200.0, 2.0, nftau=nftau)
    if name == "C3":
        return make_problem(2, 50.0, 2.0, j=0.25, nftau=nftau)
    if name == "C4":
        return make_problem(3, 50.0, 2.0, j=0.25, nftau=nftau, kanamori=True)
    if name == "C4S":      # small Kanamori case for fast parity tests
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True)
    if name == "C3Q":      # density-density U with blocks larger than one state (QNs without Azt): matrix trace path
        return make_problem(2, 20.0, 2.0, j=0.25, nftau=nftau, quantum_numbers="Nt Szt")
    # blocks of 9..16 states (outer states taken in batches on the GPU)
    if name == "C4N":      # 3-orbital Kanamori with Nt, Szt only: largest block 9 states
        return make_problem(3, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt Szt")
    if name == "C5K":      # 5-orbital Kanamori (d shell): 1024 states, largest block 10 states
        return make_problem(5, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True)
    if name == "C5KB":     # the same at the temperature of BASELINE.json's 5-orbital configuration
        return make_problem(5, 50.0, 2.0, j=0.25, nftau=nftau, kanamori=True)
    # automatic block partitioning (QuantumNumbers = All, Operator.F90:303-442)
    if name == "C4A":
        return make_problem(3, 50.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All")
    if name == "C4SA":
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All")
    if name == "C4SX":     # Kanamori + inter-orbital hopping: Qzt is not conserved, the partitioning has to find the blocks
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All", hop=0.3)
    # off-diagonal hybridisation (offdiag = 1): F(tau) between the two orbitals; with the hopping the blocks mix the orbitals
    if name == "C4SO":
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All", hop=0.3, fmix=0.3)
    if name == "C3O":      # density-density: pairs of unequal flavours never fit, the off-diagonal F enters through the determinant alone
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, fmix=0.3)
    if name == "C4O":      # three orbitals, Kanamori, hopping
        return make_problem(3, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All", hop=0.2, fmix=0.25)
    if name == "C4OB":     # the same at the temperature of BASELINE.json's 3-orbital configuration
        return make_problem(3, 50.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="Nt All", hop=0.2, fmix=0.25)
    if name == "C4SQ":     # "All" on its own: the starting partition is one block, whose operators connect nothing
        return make_problem(2, 10.0, 2.0, j=0.25, nftau=nftau, kanamori=True, quantum_numbers="All")   # (Operator.F90:150): every state ends up alone
    if name == "C3A":      # density-density: every Fock state is its own block
        return make_problem(2, 50.0, 2.0, j=0.25, nftau=nftau, quantum_numbers="All")
    if name == "C5Z":
        return make_problem(5, 100.0, 2.0, j=0.25, nftau=nftau)
    raise ValueError("unknown or not-yet-supported configuration %r" % name)
