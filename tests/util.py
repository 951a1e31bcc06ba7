"""Shared helpers of the test-suite: problems, parameter strings, fixture loading."""
import ast
import os

import numpy as np

from w2dynamics_amd import params, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["overrides"] = ast.literal_eval(str(d["overrides"]))
    d["config"] = str(d["config"]); d["nftau"] = int(d["nftau"])
    return d


def setup(cfgname, nftau, overrides):
    p = synth.config(cfgname, nftau=nftau)
    o = dict(overrides); o.setdefault("Nftau", nftau)
    sym = o.pop("SymMoves", ())         # user symmetry moves of the global move: permutations of the flavours (band + Nd * spin), 0-based
    ps = params.make_pstring(o, symmetries=sym, beta=p.beta, Nd=p.nd, Hamiltonian=p.hamiltonian, QuantumNumbers=p.quantum_numbers)
    return p, ps, params.ParameterSet(ps)


def cfg_from(d, prefix):
    return {k: d[prefix + k] for k in ("taus", "orbs", "spins", "cas", "hashybs")} | {
        "outer_sst": int(d[prefix + "outer_sst"]), "outer_state": int(d[prefix + "outer_state"])}
