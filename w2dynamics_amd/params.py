"""Parameter string handling for the CT-HYB solver boundary.

The reference's Python driver serialises the whole ``[QMC]`` section plus a handful of problem keys into
one string of ``key = value #`` lines and hands it to the solver, which re-parses it
(`w2dyn/dmft/impurity.py:500-514, 1016-1025`; `src/ctqmc_fortran/Parameters.F90:281-320`).  The solver's
typed getters match keys by *prefix* of the stored identifier (`Parameters.F90:77-105`:
``index(tmp%ID, ID) == 1`` on the first stored entry that matches), which is reproduced here.

`QMC_DEFAULTS` is the build's own table of the ``[QMC]`` keys and their defaults (values as in
`w2dyn/auxiliaries/configspec:56-180`); it exists so that tests, the bench and the harness can assemble
a complete parameter string without the reference tree being present.
"""
from __future__ import annotations

from collections import OrderedDict

# [QMC] section defaults (configspec:56-180).  Data, not code.
QMC_DEFAULTS = OrderedDict([
    ("Nwarmups", 10000), ("Nwarmups2Plus", -1), ("Nmeas", 10000), ("NCorr", 100), ("NSeed", 43890),
    ("NBin", 1), ("MaxHisto", 500), ("TaudiffMax", -1.0), ("MeasDensityMatrix", 1),
    ("PrintDensityMatrixBasis", 0), ("MeasExpResDensityMatrix", 0), ("MeasGiw", 0), ("Gtau_mean_step", 0),
    ("Gtau_mid_step", 0), ("sign_step", 0), ("NLookup_nfft", 10000), ("EnableOuterTruncation", ""),
    ("truncation", 1), ("EPSLANC", 1e-15), ("EPSTRACEVEC", 1e-15), ("EPSSANDWICH", 1e-15), ("EPSEVEC", 1e-15),
    ("EPSBLOCK", 1e-13), ("PercentageGlobalMove", 0.005), ("PercentageTauShiftMove", 0.005),
    ("PercentageOuterMove", 0.0), ("Ntau", 1000), ("Nftau", 5000), ("Niw", 2000), ("NLegMax", 100),
    ("NLegOrder", 100), ("FourPnt", 0), ("N4tau", 1), ("N4leg", 2), ("N4iwf", 1), ("N4iwb", 0), ("N2iwb", 100),
    ("N3iwb", 100), ("N3iwf", 100), ("AccumGtau", 1), ("WriteCovMeanGiw", False), ("WriteCovMeanSigmaiw", False),
    ("MeasurementTiming", -1), ("MeasG2iw", 0), ("MeasG4iwPP", 0), ("MeasGSigmaiw", 0), ("MeasGtauDetRat", 0),
    ("FixedPreallocatedVecs", 0), ("Eigenbasis", 1), ("statesampling", 0), ("ReuseMCConfig", 0),
    ("Update_Trace", 1), ("Nice_SelfEnergy", 0), ("meas_dm_old", 0), ("offdiag", 0), ("full_offdiag", 0),
    ("offd_strength", 0.0), ("flavourchange_moves", 0), ("PercentageWormInsert", 0.0),
    ("PercentageWormReplace", 0.0), ("WormEta", 1.0), ("WormMeasGiw", 0), ("WormMeasGtau", 0),
    ("WormMeasGSigmaiw", 0), ("WormMeasG4iw", 0), ("WormMeasH4iw", 0), ("WormMeasP2iwPH", 0),
    ("WormMeasP2iwPP", 0), ("WormMeasP2tauPH", 0), ("WormMeasP2tauPP", 0), ("WormMeasP3iwPH", 0),
    ("WormMeasP3iwPP", 0), ("WormMeasQQ", 0), ("WormMeasQQtau", 0), ("WormMeasNQQdag", 0), ("WormMeasQQdd", 0),
    ("WormMeasQQQQ", 0), ("WormMeasUcaca", 0), ("WormMeasUcacatau", 0), ("WormMeasUccaa", 0),
    ("WormMeasUccaatau", 0), ("WormMeasQUDdag", 0), ("WormSearchEta", 0), ("WormComponents", None),
    ("WormPHConvention", 0), ("ZPHConvention", 0), ("G4ph", 1), ("G4pp", 0), ("nfft_mode_g4iw", 0),
    ("Nbatchsize_nfft_worm", 5000000), ("segment", 0), ("MeasSusz", 0), ("Percentage4OperatorMove", 0.0),
    ("MeasSsquare", 0),
])


def config_to_pstring(config) -> str:
    """Same serialisation as `impurity.py:1016-1018`: sorted ``key = value #`` lines."""
    return "".join("%s = %s #\n" % (key, value) for key, value in sorted(config.items()))


def symm_to_pstring(symmetries) -> str:
    """`impurity.py:1021-1025`."""
    s = "NSymMove = %d #\n" % len(symmetries)
    s += "".join("SymMove%02d = %s #\n" % (i + 1, " ".join(str(int(v) + 1) for v in vals))
                 for i, vals in enumerate(symmetries))
    return s


def make_pstring(qmc_overrides=None, symmetries=(), **problem) -> str:
    """Assemble a full parameter string: QMC defaults + overrides, then the problem keys
    (`impurity.py:500-513`), then the symmetry-move block (`symmetries`: permutations of the 2 Nd flavours in the
    order band + Nd * spin, 0-based, as the reference's `problem.symmetry_moves`)."""
    qmc = dict(QMC_DEFAULTS)
    if qmc_overrides:
        for k, v in qmc_overrides.items():
            if k not in qmc:
                raise KeyError("unknown [QMC] key %r" % k)
            qmc[k] = v
    prob = {"Hamiltonian": "Density", "beta": 10.0, "Nd": 1, "ParaMag": 0, "QuantumNumbers": "Nt Szt Azt",
            "Phonon": 0, "g_phonon": "", "omega0": "", "Screening": 0, "Uw": 0}
    prob.update(problem)
    return config_to_pstring(qmc) + config_to_pstring(prob) + symm_to_pstring(list(symmetries))


class ParameterSet:
    """Parsed ``key = value #`` string with the reference's lookup semantics.

    `Parameters.F90:281-320` stores entries in file order; `get_*_Parameter` (77-105 and the typed wrappers)
    walks the list and returns the first entry whose identifier *starts with* the requested name.  A missing
    key is a fatal error in the reference; here it raises ``KeyError``.
    """

    def __init__(self, text: str):
        self.entries = []
        for line in text.splitlines():
            if "=" not in line:
                continue
            key, _, rest = line.partition("=")
            key = key.strip()
            if not key or key.startswith("#"):
                continue
            value = rest.split("#", 1)[0].strip()
            self.entries.append((key, value))

    def _find(self, name: str) -> str:
        for key, value in self.entries:
            if key.startswith(name):
                return value
        raise KeyError("parameter %r not found in parameter string" % name)

    def has(self, name: str) -> bool:
        return any(key.startswith(name) for key, _ in self.entries)

    def get_str(self, name: str) -> str:
        return self._find(name)

    def get_int(self, name: str) -> int:
        v = self._find(name)
        if v in ("True", "False"):
            return int(v == "True")
        return int(float(v)) if v else 0

    def get_real(self, name: str) -> float:
        v = self._find(name)
        return float(v) if v else 0.0

    def get_int_list(self, name: str):
        return [int(x) for x in self._find(name).split()]
