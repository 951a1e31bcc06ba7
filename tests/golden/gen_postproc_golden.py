"""Generates tests/golden/postproc.npz by importing the REFERENCE's own Python post-processing modules
(w2dyn.auxiliaries.transform / postprocessing, w2dyn.dmft.orbspin) from /root/reference.

Run in the build container only:
    python tests/golden/gen_postproc_golden.py
The fixture is data: seeded inputs and what the reference's functions returned for them.  (The reference's
`statistics` module cannot be imported here: it needs mpi4py.  The jackknife is therefore pinned by its defining
properties in tests/test_postproc.py, not by a reference output.)
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, "/root/reference")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    from w2dyn.auxiliaries import transform as tf
    from w2dyn.auxiliaries import postprocessing as pp
    from w2dyn.dmft import orbspin

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from w2dynamics_amd import synth  # noqa: E402

rng = np.random.default_rng(20240611)
out = {}
# Legendre -> Matsubara: matrix, frequencies, transform with truncation on either side, lattice convention
out["leg2mat_30_20"] = tf.leg2mat(30, 20)
out["matfreq_fermi_7.5_20"] = tf.matfreq(7.5, "fermi", 20)
out["matfreq_bose_7.5_11"] = tf.matfreq(7.5, "bose", 11)
gleg = rng.normal(size=(2, 2, 24)) * np.exp(-0.2 * np.arange(24))
out["gleg"] = gleg
giw = tf.transform(-tf.leg2mat(30, 20), gleg, None, onmismatch=tf.Truncate.tofirst, warn=False)     # matrix too big
out["giw_nleg30"] = giw
out["giw_nleg16"] = tf.transform(-tf.leg2mat(16, 20), gleg, None, onmismatch=tf.Truncate.tofirst, warn=False)   # data too big
out["giw_lattice"] = orbspin.promote_diagonal(giw.transpose(2, 0, 1))
out["giw_lattice_diag"] = orbspin.extract_diagonal(out["giw_lattice"])
# self-energy moments from density matrices: 2-orbital Kanamori U, random symmetric rho1 / rho2
U = synth.kanamori_umatrix(2, 2.0, 1.5, 0.25)
rho1 = rng.normal(size=(4, 4)); rho1 = 0.5 * (rho1 + rho1.T)
rho2 = rng.normal(size=(4, 4, 4, 4))
out["smom_U"] = U; out["smom_rho1"] = rho1.reshape(2, 2, 2, 2); out["smom_rho2"] = rho2.reshape((2, 2) * 4)
out["smom"] = pp.get_siw_mom(U, out["smom_rho1"], out["smom_rho2"])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "postproc.npz"), **out)
print("wrote postproc.npz:", {k: v.shape for k, v in out.items()})
