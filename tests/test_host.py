"""CPU: host logic - parameter string semantics, synthetic inputs, C-ABI exports, error behaviour without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from w2dynamics_amd import params, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pstring_roundtrip_and_prefix_lookup():
    ps = params.make_pstring(dict(Nmeas=123, TaudiffMax=2.5), beta=7.0, Nd=2)
    p = params.ParameterSet(ps)
    assert p.get_int("Nmeas") == 123 and p.get_real("TaudiffMax") == 2.5 and p.get_real("beta") == 7.0
    assert p.get_str("QuantumNumbers") == "Nt Szt Azt"
    # Parameters.F90:86: first stored identifier that starts with the requested name
    q = params.ParameterSet("Nmeasure = 5 #\nNmeas = 7 #\n")
    assert q.get_int("Nmeas") == 5
    with pytest.raises(KeyError):
        p.get_int("DoesNotExist")
    assert params.ParameterSet("x = 1e3 #\n").get_int("x") == 1000
    with pytest.raises(KeyError):
        params.make_pstring(dict(NotAKey=1))


def test_pstring_is_sorted_like_the_reference():
    ps = params.config_to_pstring({"b": 1, "a": 2, "C": 3})
    assert ps == "C = 3 #\na = 2 #\nb = 1 #\n"       # impurity.py:1017 sorted(config.items())


def test_bethe_ftau_properties():
    beta = 20.0
    f = synth.bethe_ftau(beta, 401)
    assert np.all(f > 0)
    assert np.allclose(f, f[::-1], rtol=1e-10)                    # particle-hole symmetric bath
    assert f[0] + f[-1] == pytest.approx(0.25, rel=1e-6)          # F(0)+F(beta) = second moment (D/2)^2
    assert f[200] < f[0]


def test_density_umatrix_convention():
    U = synth.density_umatrix(2, 2.0, 1.5, 0.25).reshape(2, 2, 2, 2, 2, 2, 2, 2)
    assert U[0, 0, 0, 1, 0, 0, 0, 1] == 2.0 and U[0, 0, 1, 1, 0, 0, 1, 1] == 1.5 and U[0, 0, 1, 0, 1, 0, 0, 0] == 0.25
    assert U[0, 0, 0, 0, 0, 0, 0, 0] == 0.0


def test_c_abi_exports_every_declared_symbol():
    from w2dynamics_amd import mctqmc
    hdr = open(os.path.join(ROOT, "include", "w2dyn_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(w2dyn_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(mctqmc.EXPORTED_SYMBOLS)
    if not os.path.exists(mctqmc.LIB_PATH):
        pytest.skip("libw2dyn_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(mctqmc.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s


def test_product_fails_loudly_without_gpu():
    import torch
    from w2dynamics_amd import mctqmc
    if not os.path.exists(mctqmc.LIB_PATH):
        pytest.skip("library not built")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(mctqmc.W2DynError):
        mctqmc.MCTQMC(device=0)


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "w2dynamics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "cthyb_oracle" not in txt and "refdrive" not in txt, f
