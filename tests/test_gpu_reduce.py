"""GPU (-m gpu): the exchange step on the device.  A 1-rank RCCL group ("nccl" backend, 127.0.0.1) carries

  * the packed host-side all-reduce of `TorchComm` (numpy -> GPU tensor -> RCCL -> numpy),
  * the device-made moments of `w2dyn_reduce_pack` (kernel -> RCCL in place -> one small copy), checked against the
    statistics of the per-bin results fetched to the host (numpy mean / std with ddof = 1, tolerance 1e-10 of the scale),
  * a whole `HipCtHybSolver.solve()` with FourPnt = 4 and one worm component.
"""
import numpy as np
import pytest

from w2dynamics_amd import synth
from w2dynamics_amd.mctqmc import MCTQMC
from w2dynamics_amd import params as P

pytestmark = pytest.mark.gpu

BASE = dict(Nwarmups=2000, Nmeas=40, NCorr=20, TaudiffMax=6.0, Nftau=501, Ntau=60, NLegMax=20, Niw=8, MeasGiw=1,
            N4iwf=3, N4iwb=2, MeasG2iw=1, MeasG4iwPP=1)


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1)
    yield dist
    if own:
        dist.destroy_process_group()


def _close(a, b, scale=None, tol=1e-10):
    a, b = np.asarray(a), np.asarray(b)
    s = max(np.abs(b).max(), 1e-300) if scale is None else scale
    return np.abs(a - b).max() <= tol * s


def test_packed_allreduce_through_rccl(rccl_group):
    from w2dynamics_amd.statistics import DistributedSample, TorchComm
    comm = TorchComm()
    assert comm.backend == "nccl" and comm.Get_size() == 1
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=(5, 7)), rng.normal(size=9) + 1j * rng.normal(size=9)
    ra, rb, rc = comm.allreduce_packed([a, b, np.array([3], dtype=np.int64)])
    assert np.array_equal(ra, a) and np.array_equal(rb, b) and rc[0] == 3
    x = rng.normal(size=(11, 4)) + 2.0
    s = DistributedSample(x, comm)
    assert _close(s.mean(), x.mean(0)) and _close(s.stderr(), x.std(0, ddof=1) / np.sqrt(11))


def test_device_moments_match_host_bins(rccl_group):
    from w2dynamics_amd.statistics import TorchComm
    comm = TorchComm()
    q = MCTQMC(nwalkers=256, walkers_per_bin=8)
    p = synth.config("C3", nftau=501)
    over = dict(P.QMC_DEFAULTS); over.update(BASE)
    q.init_rnd(11)
    q.init_paras(P.config_to_pstring(over) + P.config_to_pstring({
        "Hamiltonian": p.hamiltonian, "beta": p.beta, "Nd": p.nd, "ParaMag": 0, "QuantumNumbers": p.quantum_numbers,
        "Phonon": 0, "g_phonon": "", "omega0": "", "Screening": 0, "Uw": 0}) + P.symm_to_pstring([]))
    q.fourpnt = 4
    q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(6.0)
    q.ctqmc_warmup(2000)
    q.ctqmc_measure(1, 1)
    names = ["gleg", "gtau", "giw", "histo", "g4iw", "g4iw_pp", "g2iw"]
    pack = q.reduce_pack(names, comm)
    assert pack.tensor.is_cuda and pack.tensor.dtype.is_floating_point
    pack.allreduce()
    for n in names:
        bins = getattr(q, n + "_bins")
        s = pack.sample(n)
        assert s.ntotal == bins.shape[0]
        scale = np.abs(bins).max()
        assert _close(s.mean(), bins.mean(0), scale), n
        assert _close(s.stderr(), bins.std(0, ddof=1) / np.sqrt(bins.shape[0]), scale), n
    # a name the run did not accumulate is refused, and so is a call after the accumulators are gone
    from w2dynamics_amd.mctqmc import W2DynError
    with pytest.raises(W2DynError):
        q.reduce_pack(["g4iw_worm"], comm)
    q.dest_ctqmc()
    with pytest.raises(W2DynError):
        q.reduce_pack(["gleg"], comm)
    q.close()


def test_solver_runs_over_rccl_group(rccl_group):
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd.statistics import TorchComm
    s = HipCtHybSolver(dict(BASE), seed=5, comm=TorchComm(), nwalkers=128, walkers_per_bin=8)
    assert s.comm.backend == "nccl"
    s.set_problem(synth.config("C3", nftau=501), compute_fourpoint=4)
    r = s.solve()
    s.postprocessing(r)
    assert r["gleg"].ntotal == 16 and np.all(np.isfinite(r["siw"].mean()))
    for key in ("g4iw", "g4iw-pp", "g2iw"):
        m, e = r[key].mean(), r[key].stderr()
        assert np.iscomplexobj(m) and np.all(np.isfinite(m)) and np.all(e >= 0) and np.abs(m).max() > 0
    assert r["g4iw"].mean().shape == (2, 2, 2, 2, 6, 6, 5) and r["g2iw"].mean().shape == (2, 2, 10, 10)
    # one worm component: device-made moments of the worm estimator, normalised by nflavours^4
    s.qmc.update(PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, WormSearchEta=1, Nwarmups2Plus=3000, Nmeas=100)
    s.set_problem(synth.config("C3", nftau=501), compute_fourpoint=8)
    rw = s.solve_component(0, 4, 16, [])
    g = rw["g4iw-worm/00016"]
    assert g.mean().shape == (6, 6, 5) and np.all(np.isfinite(g.mean())) and np.abs(g.mean()).max() > 0 and np.all(g.stderr() >= 0)
    s.ctqmc.close()
