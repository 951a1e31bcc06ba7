"""GPU (-m gpu): the allocation-free entry point w2dyn_measure_once_noalloc against w2dyn_measure_once, through the C ABI.

Two runs in a row in one process, the second with fewer chains and a smaller frequency box, so that the static scratch
still holds the first run's (larger) tables when the second one starts.  Same kernels and the same order of additions on
both sides: the results must be identical bit for bit (bar of the other parity tests: 1e-10) - except G(l) and G(tau),
whose kernel (the same in both entry points) cuts the element list into equal ranges and adds the two parts of a cut
(bin, flavour) segment with atomics in arrival order: equal to rounding (1e-13 of the largest entry)."""
import numpy as np
import pytest

import util
from w2dynamics_amd.mctqmc import MCTQMC, W2DynError

pytestmark = pytest.mark.gpu


def _run(noalloc, nw, nwf, nwb, seed):
    over = dict(Ntau=100, NLegMax=12, Niw=8, MeasGiw=1, Nwarmups=2000, Nmeas=4, NCorr=20, TaudiffMax=4.0, N4iwf=nwf, N4iwb=nwb,
                MeasG2iw=1, MeasG4iwPP=1, PercentageGlobalMove=0.05, PercentageTauShiftMove=0.05)
    p, ps, pset = util.setup("C3", 501, over)
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.fourpnt = 4
    q.init_rnd(seed); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(4.0); q.ctqmc_warmup(over["Nwarmups"])
    q.set_sector(1, 1); q.measure_begin()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")                 # the runtime the library itself is linked against

    def free_bytes():                                   # hipMemGetInfo: free bytes of the device
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    q.sync()
    free0 = free_bytes()
    for _ in range(over["Nmeas"]):
        q.sweep(over["NCorr"])
        if noalloc:
            q.measure_once_noalloc()
        else:
            q.measure_once()
    q.sync()
    if noalloc:                                         # the property itself: not a byte of device memory taken or given back
        assert free_bytes() == free0
    q.measure_finish()
    res = {k: np.array(getattr(q, k)) for k in ("g4iw", "g4iw_pp", "g2iw", "gleg", "gtau", "giw", "occ", "histo")}
    res["g4iw_bins"] = np.array(q.g4iw_bins)
    q.dest_ctqmc(); q.close()
    return res


def test_noalloc_measurement_equals_measure_once_twice_in_a_row():
    for nw, nwf, nwb, seed in ((6, 3, 2, 11), (3, 2, 1, 12)):       # the second call is the smaller one
        ref = _run(False, nw, nwf, nwb, seed)
        got = _run(True, nw, nwf, nwb, seed)
        assert np.abs(ref["g4iw"]).max() > 0.0
        for k in ref:
            assert got[k].shape == ref[k].shape, k
            if k in ("gleg", "gtau"):
                assert np.abs(got[k] - ref[k]).max() <= 1e-13 * np.abs(ref[k]).max(), k
            else:
                assert np.array_equal(got[k], ref[k]), k


def test_noalloc_measurement_rejects_oversized_and_wrong_mode():
    over = dict(Ntau=100, NLegMax=12, Niw=8, Nwarmups=0, Nmeas=1, NCorr=1, TaudiffMax=4.0, N4iwf=20, N4iwb=10)
    p, ps, pset = util.setup("C3", 501, over)
    q = MCTQMC(nwalkers=512, walkers_per_bin=8)       # 512 chains x 4 flavours x 60 (60 + 21) x 16 B = 159 MB > 32 MiB
    q.fourpnt = 4
    q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(4.0); q.set_sector(1, 1); q.measure_begin()
    assert q.scalar("scratch_bytes") == 32 * 2 ** 20
    with pytest.raises(W2DynError) as e:
        q.measure_once_noalloc()
    assert e.value.code == -1 and "exceed" in str(e.value)
    q.dest_ctqmc(); q.close()
    q = MCTQMC(nwalkers=4)                              # FourPnt = 0: nothing of this entry point's to do
    q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(4.0); q.set_sector(1, 1); q.measure_begin()
    with pytest.raises(W2DynError) as e:
        q.measure_once_noalloc()
    assert e.value.code == -1
    q.dest_ctqmc(); q.close()
