"""N>1 path on CPU: two `gloo` ranks exercise the exchange step of the hot path (the only collective it has).

What is checked mirrors what the reference's `DistributedSample` promises (w2dyn/auxiliaries/statistics.py:95-209):
mean / variance / standard error / covariance over observations distributed over ranks equal the numpy results on the
concatenated sample, for real and complex data, with unequal local counts, and the packed single-collective reduction
used by bench.py / HipCtHybSolver returns the plain sums.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_data(rank):
    rng = np.random.default_rng(100 + rank)
    n = 3 + 2 * rank                         # ragged: 3 observations on rank 0, 5 on rank 1
    real = rng.standard_normal((n, 2, 2, 7))
    cplx = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4))
    return real, cplx


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from w2dynamics_amd.statistics import IN_PLACE, SUM, DistributedSample, TorchComm, join
        comm = TorchComm()
        assert comm.Get_size() == world and comm.Get_rank() == rank
        real, cplx = _local_data(rank)
        res = {}
        s = DistributedSample(real, comm)
        res["ntotal"] = s.ntotal
        res["mean"] = s.mean(); res["var"] = s.var(); res["stderr"] = s.stderr(); res["sum"] = s.sum()
        sc = DistributedSample(cplx, comm)
        res["cmean"] = sc.mean(); res["cstderr"] = sc.stderr(); res["ccov"] = sc.cov()
        res["apply"] = s.apply(lambda x: x ** 2).mean()
        j = join(s, DistributedSample(real + 1.0, comm))
        res["join_ntotal"] = j.ntotal; res["join_mean"] = j.mean()
        buf = np.full((2, 3), float(rank + 1))
        comm.Allreduce(IN_PLACE, buf, op=SUM)
        res["allreduce"] = buf
        a, b, c = comm.allreduce_packed([real.sum(0), cplx.sum(0), np.array([rank + 1], dtype=np.int64)])
        res["packed_a"] = a; res["packed_b"] = b; res["packed_c"] = c
        # moments packed the way w2dyn_reduce_pack lays them out (here made with numpy on a CPU tensor), all-reduced
        # as one tensor and read back as samples: [B | S | sum (x - S/B)^2 | S^2/B] per quantity, complex interleaved
        import torch
        from w2dynamics_amd.statistics import PackedMoments
        def record(x):
            flat = np.ascontiguousarray(x).reshape(x.shape[0], -1)
            flat = flat.view(np.float64) if np.iscomplexobj(flat) else flat
            s1 = flat.sum(0)
            return np.concatenate([[flat.shape[0]], s1, ((flat - s1 / flat.shape[0]) ** 2).sum(0), s1 ** 2 / flat.shape[0]])
        ra, rb = record(real), record(cplx)
        pm = PackedMoments(torch.from_numpy(np.concatenate([ra, rb])), {"real": (0, (ra.size - 1) // 3, real.shape[1:], False),
                                                                       "cplx": (ra.size, (rb.size - 1) // 3, cplx.shape[1:], True)}, comm).allreduce()
        res["pm_mean"] = pm.sample("real").mean(); res["pm_stderr"] = pm.sample("real").stderr()
        res["pm_cmean"] = pm.sample("cplx", 0.5).mean(); res["pm_cstderr"] = pm.sample("cplx", 0.5).stderr(); res["pm_n"] = pm.sample("cplx").ntotal
        # one observation per rank (sign, acceptance rates, rho1 ...): the error bar is the between-rank term alone.  Mean 1,
        # spread 1e-9: squaring before the shift would lose it in the rounding of n |m|^2
        tiny = np.array([[1.0 + (1e-9 if rank else -1e-9), 5.0 + 3e-9 * rank]])
        st = DistributedSample(tiny, comm)
        res["tiny_mean"] = st.mean(); res["tiny_stderr"] = st.stderr(); res["tiny_n"] = st.ntotal
        # a moments-only sample says so when asked for what it cannot give
        mo = pm.sample("real")
        msgs = []
        for call in (lambda: mo.apply(lambda x: x), lambda: join(mo, mo), lambda: mo.cov()):
            try:
                call(); msgs.append("")
            except ValueError as e:
                msgs.append(str(e))
        from w2dynamics_amd.postproc import DistributedJackknife
        try:
            DistributedJackknife(mo); msgs.append("")
        except ValueError as e:
            msgs.append(str(e))
        res["moments_only_msgs"] = np.array(msgs)
        comm.Barrier()
        # chain sharding: rank r owns chains [r*W, (r+1)*W) of one global family; no chain belongs to two ranks
        W = 5
        mine = np.arange(rank * W, (rank + 1) * W, dtype=np.float64)
        allc = np.zeros(world * W); allc[rank * W:(rank + 1) * W] = 1.0
        comm.Allreduce(IN_PLACE, allc, op=SUM)
        res["cover"] = allc; res["mine"] = mine
        if rank == 0:
            np.savez(out, **res)
    finally:
        dist.destroy_process_group()


def test_two_rank_statistics_and_packed_allreduce(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    reals, cplxs = zip(*[_local_data(k) for k in range(2)])
    real = np.concatenate(reals); cplx = np.concatenate(cplxs)
    n = real.shape[0]
    assert int(r["ntotal"]) == n == 8
    np.testing.assert_allclose(r["sum"], real.sum(0), rtol=1e-13)
    np.testing.assert_allclose(r["mean"], real.mean(0), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(r["var"], real.var(0, ddof=1), rtol=1e-12)
    np.testing.assert_allclose(r["stderr"], real.std(0, ddof=1) / np.sqrt(n), rtol=1e-12)
    np.testing.assert_allclose(r["cmean"], cplx.mean(0), rtol=1e-13)
    np.testing.assert_allclose(r["cstderr"], cplx.std(0, ddof=1) / np.sqrt(n), rtol=1e-12)
    d = cplx - cplx.mean(0)
    np.testing.assert_allclose(r["ccov"], np.einsum("ij,ik->jk", d, d.conj()) / (n - 1), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(r["apply"], (real ** 2).mean(0), rtol=1e-13)
    assert int(r["join_ntotal"]) == 2 * n
    np.testing.assert_allclose(r["join_mean"], real.mean(0) + 0.5, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(r["allreduce"], np.full((2, 3), 3.0))
    np.testing.assert_allclose(r["packed_a"], real.sum(0), rtol=1e-13)
    np.testing.assert_allclose(r["packed_b"], cplx.sum(0), rtol=1e-13)
    np.testing.assert_allclose(r["packed_c"], [3.0])
    np.testing.assert_array_equal(r["cover"], np.ones(10))
    np.testing.assert_allclose(r["pm_mean"], real.mean(0), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(r["pm_stderr"], real.std(0, ddof=1) / np.sqrt(n), rtol=1e-12)
    np.testing.assert_allclose(r["pm_cmean"], 0.5 * cplx.mean(0), rtol=1e-13)
    np.testing.assert_allclose(r["pm_cstderr"], 0.5 * cplx.std(0, ddof=1) / np.sqrt(n), rtol=1e-12)
    assert int(r["pm_n"]) == n
    tiny = np.array([[1.0 - 1e-9, 5.0], [1.0 + 1e-9, 5.0 + 3e-9]])
    assert int(r["tiny_n"]) == 2
    np.testing.assert_allclose(r["tiny_mean"], tiny.mean(0), rtol=1e-15)
    np.testing.assert_allclose(r["tiny_stderr"], tiny.std(0, ddof=1) / np.sqrt(2), rtol=1e-6)      # 1e-9 and 1.5e-9, not 0
    assert all("moments only" in m for m in r["moments_only_msgs"])


def test_single_process_comm_is_identity():
    sys.path.insert(0, ROOT)
    from w2dynamics_amd.statistics import DistributedSample, TorchComm
    x = np.arange(12.0).reshape(4, 3)
    s = DistributedSample(x, TorchComm())
    assert s.ntotal == 4
    np.testing.assert_allclose(s.mean(), x.mean(0))
    np.testing.assert_allclose(s.stderr(), x.std(0, ddof=1) / 2.0)
    with pytest.raises(ValueError):
        DistributedSample(x, TorchComm(), ntotal=3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # a single observation: NaN error bar, no division warning
        one = DistributedSample(x[:1], TorchComm())
        assert np.all(np.isnan(one.stderr())) and np.array_equal(one.mean(), x[0])
