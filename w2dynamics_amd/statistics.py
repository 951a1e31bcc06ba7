"""Cross-rank statistics of the solver results: the exchange step of the hot path.

The reference wraps every result in `DistributedSample` (`w2dyn/auxiliaries/statistics.py`) and obtains mean and
standard error with one MPI all-reduce per quantity and statistic.  This module offers the same *interface* (the names
the reference's driver and result writer use: `mean`, `stderr`, `var`, `sum`, `apply`, `local`, `ntotal`, `join`), built
differently:

  * every sample is reduced to its moments once - per rank the count n_r, the sum, and the squared deviations from the
    rank's own mean - and ONE packed fp64 all-reduce combines (n_r, n_r m_r, M2_r) of all ranks; with more than one rank
    a second, equally packed all-reduce adds the between-rank part n_r |m_r - m|^2 about the GLOBAL mean m (two-pass:
    squaring after the shift, so a rank-to-rank spread of 1e-9 of the mean survives - samples with one observation per
    rank consist of this term alone); mean, variance and standard error are then read off the cached moments.
    On xGMI a ring all-reduce is bound by one link, so one bucket per pass instead of one collective per quantity and statistic;
  * the moments can also come straight from the GPU (`from_moments`, fed by `MCTQMC.reduce_pack`): the per-bin results
    are summed and squared by a kernel and the packed device buffer goes into `torch.distributed.all_reduce` (RCCL) as it
    is - no numpy round trip of data that already sits in HBM.

Semantics: the reference has one observation per MPI rank; here every GPU contributes `nbins` observations
(accumulator bins of a few chains each), so the error bar is taken over all bins of all GPUs.

`TorchComm` is an mpi4py-communicator look-alike (`Allreduce`, `Barrier`, `Get_rank`, `Get_size`) over a
torch.distributed process group - "nccl" = RCCL with one process per GPU, "gloo" in the CPU tests - so the reference's own
statistics class would run on it unchanged as well.
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


def _as_f64_view(a):
    """(flat float64 view of a contiguous copy, restore function)."""
    a = np.asarray(a)
    cplx = np.iscomplexobj(a)
    work = np.ascontiguousarray(a, dtype=np.complex128 if cplx else np.float64)
    flat = work.view(np.float64).ravel()

    def restore(buf):
        return (buf.view(np.complex128) if cplx else buf).reshape(a.shape)
    return flat, restore


class TorchComm:
    """Communicator over a torch.distributed process group (None / not initialised = single process)."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.active = dist is not None and dist.is_available() and dist.is_initialized()
        self.backend = dist.get_backend(group) if self.active else None
        if device is None and self.backend == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device or "cpu"

    def Get_rank(self):
        return dist.get_rank(self.group) if self.active else 0

    def Get_size(self):
        return dist.get_world_size(self.group) if self.active else 1

    def Barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    # -- device path --------------------------------------------------------------------------------------------
    def allreduce_tensor(self, t):
        """In-place sum-all-reduce of a torch tensor where it lives (a CUDA tensor goes to RCCL untouched)."""
        if self.active and (self.Get_size() > 1 or self.backend == "nccl"):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # -- host path ----------------------------------------------------------------------------------------------
    def _reduce_flat(self, flat):
        if not self.active or (self.Get_size() == 1 and self.backend != "nccl"):
            return flat
        t = torch.from_numpy(flat).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        """mpi4py-style sum-all-reduce into `recvbuf` (sendbuf may be IN_PLACE)."""
        if op != SUM:
            raise NotImplementedError("the solver statistics only sum")
        if sendbuf is not IN_PLACE:
            np.copyto(recvbuf, sendbuf)
        flat, restore = _as_f64_view(recvbuf)
        recvbuf[...] = restore(self._reduce_flat(flat.copy()))

    def allreduce_packed(self, arrays):
        """Sum-all-reduce several real / complex / integer arrays with ONE collective; returns new arrays."""
        views = [_as_f64_view(a) for a in arrays]
        if not views:
            return []
        out = self._reduce_flat(np.concatenate([v for v, _ in views]))
        res, pos = [], 0
        for v, restore in views:
            res.append(np.array(restore(out[pos:pos + v.size])))
            pos += v.size
        return res


class DistributedSample:
    """Observations x_1..x_n of one quantity spread over the ranks: `local` holds this rank's (n_r, ...) block.

    All statistics come from the cached moments of one packed all-reduce (module docstring)."""

    def __init__(self, local, comm=None, ntotal=None):
        self.comm = comm if comm is not None else TorchComm()
        block = np.asarray(local)
        self.local = block.reshape(1) if block.ndim == 0 else block
        self.nlocal = self.local.shape[0]
        self._moments = None
        self._ntotal = None
        if ntotal is not None:
            alone = self.comm.Get_size() == 1
            if ntotal < self.nlocal or (alone and ntotal != self.nlocal):
                raise ValueError("ntotal = %d does not fit %d local observations on %d rank(s)" % (ntotal, self.nlocal, self.comm.Get_size()))
            self._ntotal = int(ntotal)

    # the reference's attribute name for the communicator (its result writer reads it)
    @property
    def mpi_comm(self):
        return self.comm

    @classmethod
    def from_moments(cls, n, s1, m2, between, comm=None, reduced=True):
        """Sample known only through its moments - count, sum, squared deviations from the rank means and
        n_r |m_r|^2 - e.g. one record of `PackedMoments` (made on the GPU, all-reduced there).  `reduced=False`: the
        moments are this rank's and are all-reduced here.  `local` is not available for such a sample.  The
        between-rank part between - n |m|^2 is a difference of large numbers: it is left out on a single rank (where it
        is zero by construction) and only trusted down to ~1e-15 n |m|^2 across ranks, which is far below the scatter of
        the accumulator bins these records are made of."""
        self = cls.__new__(cls)
        self.comm = comm if comm is not None else TorchComm()
        self.local = None; self.nlocal = int(round(float(n)))
        s1 = np.asarray(s1); m2 = np.asarray(m2, dtype=np.float64); between = np.asarray(between, dtype=np.float64)
        if not reduced:
            nn, s1, m2, between = self.comm.allreduce_packed([np.array([float(n)]), s1, m2, between])
            n = nn[0]
        n = float(n)
        self._ntotal = int(round(n))
        if self.comm.Get_size() > 1:
            m2 = m2 + np.maximum(between - n * np.abs(s1 / n) ** 2, 0.0)
        self._moments = (n, s1, m2)
        return self

    def _need_observations(self, what):
        if self.local is None:
            raise ValueError("%s needs the observations, but this sample holds moments only (count, sum, squared deviations: a record "
                             "of MCTQMC.reduce_pack / PackedMoments); take mean(), var(), stderr() from it" % what)

    def _reduce(self):
        if self._moments is None:
            x = self.local
            n_r = float(self.nlocal)
            if self.nlocal:
                m_r = x.mean(axis=0)
                dev = x - m_r
                m2_r = (dev.real ** 2 + dev.imag ** 2).sum(axis=0) if np.iscomplexobj(x) else (dev ** 2).sum(axis=0)
            else:
                m_r = np.zeros(x.shape[1:], x.dtype); m2_r = np.zeros(x.shape[1:])
            n, s1, m2 = self.comm.allreduce_packed([np.array([n_r]), n_r * m_r, m2_r])
            n = float(n[0])
            if self._ntotal is not None and self._ntotal != int(round(n)):
                raise ValueError("ntotal = %d given, but the ranks hold %d observations" % (self._ntotal, int(round(n))))
            self._ntotal = int(round(n))
            if self.comm.Get_size() > 1:
                # parallel-axis formula: sum_i |x_i - m|^2 = sum_r [ M2_r + n_r |m_r - m|^2 ], the shift done before squaring
                between, = self.comm.allreduce_packed([n_r * np.abs(m_r - s1 / n) ** 2])
                m2 = m2 + between
            self._moments = (n, s1, m2)
        return self._moments

    @property
    def ntotal(self):
        if self._ntotal is None:
            self._reduce()
        return self._ntotal

    def sum(self):
        return np.array(self._reduce()[1])

    def mean(self):
        n, s1, _ = self._reduce()
        return s1 / n

    def var(self, mean=None, ddof=1):
        n, s1, m2 = self._reduce()
        if mean is not None:      # second moment about a caller-supplied point
            m2 = m2 + n * np.abs(np.asarray(mean) - s1 / n) ** 2
        if n - ddof <= 0:         # a single observation has no spread to speak of: NaN, quietly (the reference divides by zero)
            return np.full(np.shape(m2), np.nan)
        return m2 / (n - ddof)

    def stddev(self, var=None, ddof=1):
        return np.sqrt(self.var(ddof=ddof) if var is None else var)

    def stderr(self, stddev=None, ddof=1):
        n = self._reduce()[0]
        return (self.stddev(ddof=ddof) if stddev is None else stddev) / np.sqrt(n)

    def cov(self, mean=None, ddof=1):
        """Covariance matrix of the flattened observation, shape = shape + shape (second factor conjugated)."""
        self._need_observations("cov()")
        n, s1, _ = self._reduce()
        centre = (s1 / n if mean is None else np.asarray(mean)).ravel()
        d = self.local.reshape(self.nlocal, -1) - centre
        outer, = self.comm.allreduce_packed([d.T @ d.conj()])
        return outer.reshape(s1.shape * 2) / (n - ddof)

    def cov_of_mean(self, cov=None):
        return (self.cov() if cov is None else cov) / self._reduce()[0]

    def apply(self, func, inplace=False):
        """Sample of func(x_i)."""
        self._need_observations("apply()")
        mapped = np.array([func(obs) for obs in self.local])
        if inplace:
            self.local = mapped; self._moments = None
            return self
        return DistributedSample(mapped, self.comm, self._ntotal)


def join(*samples):
    """One sample holding the observations of all arguments (each rank concatenates its local blocks)."""
    if not samples:
        raise ValueError("join() needs at least one sample")
    for smp in samples:
        smp._need_observations("join()")
    return DistributedSample(np.concatenate([s.local for s in samples], axis=0), samples[0].comm, sum(s.ntotal for s in samples))


class PackedMoments:
    """Moments of several per-bin results in ONE device buffer (`MCTQMC.reduce_pack`, C ABI `w2dyn_reduce_pack`): the
    buffer is all-reduced where it lives (RCCL) and only the reduced moments are copied to the host."""

    def __init__(self, tensor, layout, comm=None):
        self.tensor = tensor            # 1-d float64 torch tensor on the GPU, aliasing the library's buffer until allreduce()
        self.layout = layout            # name -> (offset, doubles per bin, shape, is complex)
        self.comm = comm if comm is not None else TorchComm()
        self._host = None

    def allreduce(self):
        """Reduce over the ranks and copy the result to the host.  With RCCL the device buffer is reduced where it lies;
        any other backend (the gloo groups of the CPU tests and of one-GPU rehearsals) gets a host copy.  The alias of the
        library's buffer is dropped afterwards: the next `reduce_pack` or `w2dyn_destroy` may free that memory."""
        if self.tensor is None:
            raise RuntimeError("PackedMoments.allreduce called twice")
        if self.comm.active and self.comm.backend != "nccl":
            self._host = self.comm._reduce_flat(self.tensor.cpu().numpy())
        else:
            self.comm.allreduce_tensor(self.tensor)
            self._host = self.tensor.cpu().numpy()
        self.tensor = None
        return self

    def sample(self, name, scale=1.0):
        """DistributedSample (moments only) of `scale` times one packed result; call after `allreduce`."""
        if self._host is None:
            raise RuntimeError("PackedMoments.sample before allreduce")
        off, per, shape, cplx = self.layout[name][:4]
        move = self.layout[name][4] if len(self.layout[name]) > 4 else None       # (axis of the device order, where it belongs)
        rec = self._host[off:off + 1 + 3 * per]
        n, s1, m2, bt = rec[0], rec[1:1 + per], rec[1 + per:1 + 2 * per], rec[1 + 2 * per:]
        if cplx:
            s1 = s1[0::2] + 1j * s1[1::2]; m2 = m2[0::2] + m2[1::2]; bt = bt[0::2] + bt[1::2]
        s1, m2, bt = (np.moveaxis(a.reshape(shape), *move) if move else a.reshape(shape) for a in (s1, m2, bt))
        return DistributedSample.from_moments(n, scale * s1, scale ** 2 * m2, scale ** 2 * bt, self.comm)
