"""Cross-rank statistics of the solver results: the "parallel reduction" of the hot path.

Mirrors `w2dyn/auxiliaries/statistics.py:19-248` (`DistributedSample`): a sample is an (n, ...) array of
observations distributed over ranks; mean / variance / standard error are obtained with sum-all-reduces
(`statistics.py:95-97` uses mpi4py `Allreduce(IN_PLACE, SUM)`).  Here the collective is `torch.distributed.all_reduce`
- RCCL over xGMI when the process group is "nccl" (one process per GPU), gloo on CPU in the tests.

Semantics note: the reference has one observation per MPI rank; the HIP solver has `nbins`
observations per GPU (accumulator bins of a few chains each), so `nlocal = nbins` and the error bar is taken over
all bins of all GPUs.  The formulas are unchanged.

`TorchComm` additionally offers the three mpi4py calls the reference's solver wrapper uses on its communicator
(`Allreduce`, `Barrier`, `Get_rank/Get_size`; impurity.py:379, 575, statistics.py:97), so that the reference's own
`DistributedSample` can run on top of RCCL unchanged.
"""
from __future__ import annotations

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

IN_PLACE = "IN_PLACE"
SUM = "SUM"


class TorchComm:
    """mpi4py-communicator look-alike on top of a torch.distributed process group (None = single process)."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.active = dist is not None and dist.is_available() and dist.is_initialized()
        if device is None and self.active and dist.get_backend(group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device or "cpu"

    def Get_rank(self):
        return dist.get_rank(self.group) if self.active else 0

    def Get_size(self):
        return dist.get_world_size(self.group) if self.active else 1

    def Barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        """In-place sum-all-reduce of a numpy array (sendbuf must be IN_PLACE as in statistics.py:97)."""
        if op != SUM:
            raise NotImplementedError("only SUM is needed by the solver statistics")
        if sendbuf is not IN_PLACE:
            np.copyto(recvbuf, sendbuf)
        if not self.active or self.Get_size() == 1:
            return
        arr = np.ascontiguousarray(recvbuf)
        if np.iscomplexobj(arr):
            t = torch.from_numpy(arr.view(np.float64).copy())
        else:
            t = torch.from_numpy(arr.copy())
        t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        out = t.cpu().numpy()
        if np.iscomplexobj(arr):
            out = out.view(np.complex128)
        recvbuf[...] = out.reshape(recvbuf.shape)

    def allreduce_packed(self, arrays):
        """Sum-all-reduce several float64/complex128/int64 arrays with ONE collective (bucketed: on xGMI a ring
        all-reduce is per-link bound, so few large messages beat many small ones)."""
        if not self.active or self.Get_size() == 1:
            return [np.array(a) for a in arrays]
        flat = [np.ascontiguousarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64).view(np.float64).ravel()
                for a in arrays]
        sizes = [f.size for f in flat]
        t = torch.from_numpy(np.concatenate(flat)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        out = t.cpu().numpy()
        res, pos = [], 0
        for a, n in zip(arrays, sizes):
            seg = out[pos:pos + n]; pos += n
            a = np.asarray(a)
            if np.iscomplexobj(a):
                res.append(seg.view(np.complex128).reshape(a.shape))
            else:
                res.append(seg.reshape(a.shape).astype(a.dtype if a.dtype.kind == "f" else np.float64))
        return res


class DistributedSample:
    """Same interface as the reference's class (statistics.py:19-248): local (n,...) observations + communicator."""

    def __init__(self, local, comm=None, ntotal=None):
        self.mpi_comm = comm if comm is not None else TorchComm()
        self.local = np.asarray(local)
        if self.local.ndim == 0:
            self.local = self.local.ravel()
        self.nlocal = self.local.shape[0]
        if ntotal is not None:
            if ntotal < self.nlocal:
                raise ValueError("ntotal must at least cover local")
            if self.mpi_comm.Get_size() == 1 and ntotal != self.nlocal:
                raise ValueError("ntotal inconsistent for single-core")
            self.ntotal = ntotal
        else:
            n = np.array(self.nlocal, np.float64).reshape(1)
            self._reduce_inplace(n)
            self.ntotal = int(round(float(n[0])))

    def _reduce_inplace(self, arr):
        self.mpi_comm.Allreduce(IN_PLACE, arr, op=SUM)

    def apply(self, func, inplace=False):
        new = np.array([func(self.local[i, ...]) for i in range(self.local.shape[0])])
        if inplace:
            self.local = new
            return self
        return DistributedSample(new, self.mpi_comm, self.ntotal)

    def sum(self):
        result = np.array(self.local.sum(0))
        self._reduce_inplace(result)
        return result

    def _sqdiff(self, mean, dtype=float):
        sq = np.array(sum(np.abs(b - mean).astype(dtype) ** 2 for b in self.local), dtype=dtype)
        self._reduce_inplace(sq)
        return sq

    def _bidiff(self, mean, dtype=float):
        mean = np.asarray(mean)
        flat = np.reshape(self.local, (self.local.shape[0], -1)) - mean.ravel()
        diff = np.einsum("ij,ik->jk", flat, np.conj(flat)).reshape(mean.shape * 2)
        diff = np.array(diff)
        self._reduce_inplace(diff)
        return diff

    def mean(self):
        return self.sum() / self.ntotal

    def var(self, mean=None, ddof=1):
        r = self._sqdiff(self.mean() if mean is None else mean)
        return r / (self.ntotal - ddof)

    def stddev(self, var=None, ddof=1):
        return np.sqrt(self.var(ddof=ddof) if var is None else var)

    def stderr(self, stddev=None, ddof=1):
        if stddev is None:
            r = self._sqdiff(self.mean())
            return np.sqrt(r / (self.ntotal * (self.ntotal - ddof)))
        return stddev / np.sqrt(self.ntotal)

    def cov(self, mean=None, ddof=1):
        return self._bidiff(self.mean() if mean is None else mean) / (self.ntotal - ddof)

    def cov_of_mean(self, cov=None):
        return (self.cov() if cov is None else cov) / self.ntotal


def join(*samples):
    """statistics.py:5-16"""
    if not samples:
        raise ValueError("Must pass list of at least one sample")
    comm = samples[0].mpi_comm
    local = np.concatenate(tuple(s.local for s in samples), axis=0)
    return DistributedSample(local, comm, sum(s.ntotal for s in samples))
