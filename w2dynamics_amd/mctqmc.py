"""`mctqmc`-shaped front end of libw2dyn_hip.so (ctypes), the drop-in for the reference's f2py module.

The reference's driver does ``from w2dyn.auxiliaries.CTQMC import mctqmc as ctqmc`` (`w2dyn/dmft/impurity.py:18`) and
then calls lower-case module procedures and reads lower-case module variables (`impurity.py:387-388, 514-515,
578-598, 650-651, 1198-1283`).  `MCTQMC` below offers the same names with the same argument order and array shapes;
`INTEGRATION.md` shows the one-line change on the reference side.

There is no CPU fallback: if the HIP library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W2DYN_HIP_LIB", os.path.join(_HERE, "libw2dyn_hip.so"))   # override: diagnostic builds

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class W2DynError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libw2dyn_hip error %d: %s" % (code, msg))
        self.code = code


def load_library(path: str = LIB_PATH):
    """Load the C ABI and declare every prototype of include/w2dyn_hip.h. Raises if the library is absent."""
    if not os.path.exists(path):
        raise ImportError("HIP extension %s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    # PyTorch ships a HIP runtime of its own.  When it is imported AFTER this library has brought in the system's, the process holds
    # two runtimes and the second one finds no device ("No HIP GPUs are available" from the first torch.cuda call: reduce_pack, the
    # RCCL group).  Imported first, its copy serves both.  Callers that never touch torch lose nothing but the import time.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    L.w2dyn_create.argtypes = [C.POINTER(vp), C.c_int]
    L.w2dyn_destroy.argtypes = [vp]
    L.w2dyn_last_error.argtypes = [vp]; L.w2dyn_last_error.restype = C.c_char_p
    L.w2dyn_set_layout.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int]
    L.w2dyn_init_rnd.argtypes = [vp, C.c_int64]
    L.w2dyn_init_paras.argtypes = [vp, C.c_char_p]
    L.w2dyn_dest_paras.argtypes = [vp]
    L.w2dyn_set_fourpnt.argtypes = [vp, C.c_int]
    L.w2dyn_init_solver.argtypes = [vp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int]
    L.w2dyn_set_taudiffmax.argtypes = [vp, C.c_double]
    L.w2dyn_init_counters.argtypes = [vp]
    L.w2dyn_ctqmc_warmup.argtypes = [vp, C.c_int64]
    L.w2dyn_ctqmc_calibrate.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.w2dyn_ctqmc_measure.argtypes = [vp, C.c_int, C.c_int]
    L.w2dyn_dest_ctqmc.argtypes = [vp]
    L.w2dyn_set_wormeta.argtypes = [vp, C.c_double]
    L.w2dyn_get_wormeta.argtypes = [vp, C.POINTER(C.c_double)]
    L.w2dyn_ctqmc_worm_warmup.argtypes = [vp, C.c_int64, C.c_int, C.c_int]
    L.w2dyn_count_steps.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.w2dyn_measure_begin.argtypes = [vp]
    L.w2dyn_set_sector.argtypes = [vp, C.c_int, C.c_int]
    L.w2dyn_save_mc_configs.argtypes = [vp]
    L.w2dyn_restore_mc_configs.argtypes = [vp]
    L.w2dyn_sweep.argtypes = [vp, C.c_int64, C.c_int]
    L.w2dyn_measure_once.argtypes = [vp]
    L.w2dyn_measure_once_noalloc.argtypes = [vp]
    L.w2dyn_measure_finish.argtypes = [vp]
    L.w2dyn_sync.argtypes = [vp]
    L.w2dyn_get_mc_config_scalars.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.w2dyn_get_mc_config.argtypes = [vp, C.c_int, C.c_int, _dp, _ip, _ip, _ip, _ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.w2dyn_set_mc_config.argtypes = [vp, C.c_int, C.c_int, _dp, _ip, _ip, _ip, _ip, C.c_int, C.c_int]
    L.w2dyn_get_minv.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    L.w2dyn_get_walker_state.argtypes = [vp, C.c_int, _dp]
    L.w2dyn_get_result.argtypes = [vp, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.w2dyn_get_scalar.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    L.w2dyn_kernel_time.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.w2dyn_set_timing.argtypes = [vp, C.c_int]
    L.w2dyn_reduce_pack.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]
    return L


EXPORTED_SYMBOLS = [
    "w2dyn_create", "w2dyn_destroy", "w2dyn_last_error", "w2dyn_set_layout", "w2dyn_init_rnd", "w2dyn_init_paras",
    "w2dyn_dest_paras", "w2dyn_set_fourpnt", "w2dyn_init_solver", "w2dyn_set_taudiffmax", "w2dyn_init_counters",
    "w2dyn_ctqmc_warmup", "w2dyn_ctqmc_measure", "w2dyn_dest_ctqmc", "w2dyn_measure_begin", "w2dyn_sweep",
    "w2dyn_measure_once", "w2dyn_measure_finish", "w2dyn_sync", "w2dyn_get_mc_config_scalars", "w2dyn_get_mc_config",
    "w2dyn_set_mc_config", "w2dyn_get_minv", "w2dyn_get_walker_state", "w2dyn_get_result", "w2dyn_get_scalar",
    "w2dyn_kernel_time", "w2dyn_set_wormeta", "w2dyn_get_wormeta", "w2dyn_ctqmc_worm_warmup", "w2dyn_count_steps",
    "w2dyn_ctqmc_calibrate", "w2dyn_set_sector", "w2dyn_save_mc_configs", "w2dyn_restore_mc_configs", "w2dyn_set_timing",
    "w2dyn_measure_once_noalloc", "w2dyn_reduce_pack",
]

_SCALARS = ("mean_sign", "aborted", "time_warmup", "time_sim", "time_sim_steps", "accglob", "accshift", "accadd4",
            "accrem4", "accflavc")


class MCTQMC:
    """One solver context on one GPU with the interface of the reference's `mctqmc` module.

    Extra, HIP-specific knobs (no reference counterpart): `nwalkers` chains per GPU, `first_walker` (global index of
    this context's first chain: rank * nwalkers in a multi-GPU run), `kmax` (capacity per flavour, 0 = automatic),
    `walkers_per_bin` (chains per accumulator bin; bins give the error bars)."""

    def __init__(self, device=0, nwalkers=1024, first_walker=0, kmax=0, walkers_per_bin=8, lib=None):
        self._L = lib or load_library()
        self._h = C.c_void_p()
        rc = self._L.w2dyn_create(C.byref(self._h), int(device))
        if rc:
            raise W2DynError(rc, "cannot create a solver context on GPU %d (is a HIP device visible?)" % device)
        self.nwalkers, self.first_walker, self.kmax, self.walkers_per_bin = nwalkers, first_walker, kmax, walkers_per_bin
        # writable module variables of the reference (impurity.py:388, 515, 578, 719)
        self.fancy_prog = False
        self.fourpnt = 0
        self.simid = 0
        self.wormeta = np.zeros(15)
        self._shape = None

    # -- helpers ---------------------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise W2DynError(rc, self._L.w2dyn_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.w2dyn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- module procedures -----------------------------------------------------------------------------------
    def init_rnd(self, nseed):
        self._chk(self._L.w2dyn_init_rnd(self._h, int(nseed)))

    def init_paras(self, parastr):
        self._chk(self._L.w2dyn_init_paras(self._h, parastr.encode()))

    def dest_paras(self):
        self._chk(self._L.w2dyn_dest_paras(self._h))

    def init_solver(self, u_matrix_in, ftau_full, muimp_full, screening_function, iter_no, nftau=None, nbands=None):
        u = np.ascontiguousarray(u_matrix_in, dtype=np.float64)
        f = np.ascontiguousarray(ftau_full, dtype=np.float64)
        m = np.ascontiguousarray(muimp_full, dtype=np.float64)
        sc = np.ascontiguousarray(screening_function, dtype=np.float64)
        nftau = f.shape[4] if nftau is None else nftau
        nbands = u.shape[0] // 2 if nbands is None else nbands
        if f.shape != (nbands, 2, nbands, 2, nftau) or m.shape != (nbands, 2, nbands, 2) or u.shape != (2 * nbands,) * 4:
            raise ValueError("init_solver: array shapes do not match (Nd,2,Nd,2,Nftau)/(Nd,2,Nd,2)/(2Nd)^4")
        self._chk(self._L.w2dyn_set_layout(self._h, int(self.nwalkers), int(self.first_walker), int(self.kmax), int(self.walkers_per_bin)))
        self._chk(self._L.w2dyn_set_fourpnt(self._h, int(self.fourpnt)))
        self._chk(self._L.w2dyn_init_solver(self._h, u, f, m, sc, int(iter_no), int(nftau), int(nbands)))
        self._nd = nbands
        self._n4iwf = int(self.scalar("n4iwf")); self._n4iwb = int(self.scalar("n4iwb")); self._ntau = int(self.scalar("ntau"))
        self.wormeta[:] = self.scalar("wormeta")          # wormEta = WormEta for every sector, CTQMC.F90:434

    def set_taudiffmax(self, x):
        self._chk(self._L.w2dyn_set_taudiffmax(self._h, float(x)))

    def init_counters(self):
        self._chk(self._L.w2dyn_init_counters(self._h))

    def ctqmc_calibrate(self, list_, phase1pairnum, phase2taugrid, phase2pairnum, nprogress):
        self._chk(self._L.w2dyn_ctqmc_calibrate(self._h, int(bool(list_)), int(phase1pairnum), int(phase2taugrid),
                                                int(phase2pairnum), int(nprogress)))

    def ctqmc_warmup(self, n):
        self._chk(self._L.w2dyn_ctqmc_warmup(self._h, int(n)))

    def _push_wormeta(self, isector):
        # `wormeta` is a writable module array wormEta(2:NWormSectors) in the reference (impurity.py:719, 755, 784)
        if int(isector) >= 2:
            self._chk(self._L.w2dyn_set_wormeta(self._h, float(self.wormeta[int(isector) - 2])))

    def ctqmc_measure(self, isector, icomponent):
        self._push_wormeta(isector)
        self._chk(self._L.w2dyn_ctqmc_measure(self._h, int(isector), int(icomponent)))

    def ctqmc_worm_warmup(self, nwarmups2, isector, icomponent):
        self._chk(self._L.w2dyn_ctqmc_worm_warmup(self._h, int(nwarmups2), int(isector), int(icomponent)))

    def count_steps(self, isector, icomponent, nfix, steps_worm=None, steps_z=None):
        """count_steps(isector, icomponent, nfix, steps_worm, steps_z) of the reference (CTQMC.F90:6296): the two
        0-d integer arrays are filled in place, as f2py does for intent(inout); the pair is returned as well."""
        self._push_wormeta(isector)
        a, b = C.c_int64(), C.c_int64()
        self._chk(self._L.w2dyn_count_steps(self._h, int(isector), int(icomponent), int(nfix), C.byref(a), C.byref(b)))
        if steps_worm is not None:
            steps_worm[...] = a.value
        if steps_z is not None:
            steps_z[...] = b.value
        return a.value, b.value

    def dest_ctqmc(self):
        self._chk(self._L.w2dyn_dest_ctqmc(self._h))

    # pieces of ctqmc_measure (bench / tests)
    def measure_begin(self): self._chk(self._L.w2dyn_measure_begin(self._h))

    def set_sector(self, isector, icomponent):
        self._push_wormeta(isector)
        self._chk(self._L.w2dyn_set_sector(self._h, int(isector), int(icomponent)))
    def sweep(self, nsteps, count_as_sampling=True): self._chk(self._L.w2dyn_sweep(self._h, int(nsteps), int(count_as_sampling)))
    def measure_once(self): self._chk(self._L.w2dyn_measure_once(self._h))
    def measure_once_noalloc(self): self._chk(self._L.w2dyn_measure_once_noalloc(self._h))
    def measure_finish(self): self._chk(self._L.w2dyn_measure_finish(self._h))
    def sync(self): self._chk(self._L.w2dyn_sync(self._h))

    def set_timing(self, on=True):
        """HIP-event timing of every launch (bench / profiling): off by default, it costs a host sync per launch."""
        self._chk(self._L.w2dyn_set_timing(self._h, int(bool(on))))

    def reduce_pack(self, names, comm=None):
        """Moments of the per-bin results `names` packed on the GPU for the cross-rank reduction (w2dyn_reduce_pack):
        a `statistics.PackedMoments` whose tensor aliases the library's device buffer - no copy, no host round trip."""
        import torch
        from .statistics import PackedMoments
        names = list(names)
        ptr, n = C.c_void_p(), C.c_size_t()
        offs, cnts = (C.c_size_t * len(names))(), (C.c_size_t * len(names))()
        self._chk(self._L.w2dyn_reduce_pack(self._h, ",".join(names).encode(), C.byref(ptr), C.byref(n), offs, cnts, len(names)))
        dev = int(self.scalar("device"))

        class _Alias:      # a raw device pointer in the shape torch understands
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<f8", "data": (ptr.value, False), "version": 2}
        tensor = torch.as_tensor(_Alias(), device=torch.device("cuda", dev))
        layout = {}
        nd, nwf, nwb = self._nd, self._n4iwf, self._n4iwb
        for i, name in enumerate(names):
            per = cnts[i]
            cplx = name not in ("gleg", "gtau", "histo", "gtau_worm")
            if name in ("gleg", "gtau", "histo", "giw"):
                shape = (nd, 2, per // (2 * nd * (2 if cplx else 1)))
            elif name in ("g4iw", "g4iw_pp"):      # device order: bosonic frequency in front of the two fermionic ones
                shape = (nd, 2, nd, 2, 2 * nwb + 1, 2 * nwf, 2 * nwf)
            elif name == "g2iw":
                shape = (nd, 2, 2 * (nwf + nwb), 2 * (nwf + nwb))
            elif name in ("g4iw_worm", "g4iwpp_worm"):
                shape = (2 * nwf, 2 * nwf, 2 * nwb + 1)
            else:
                shape = (per // (2 if cplx else 1),)
            assert int(np.prod(shape)) * (2 if cplx else 1) == per, (name, shape, per)
            layout[name] = (offs[i], per, shape, cplx) + (((4, -1),) if name in ("g4iw", "g4iw_pp") else ())      # -> (.., w1, w2, wb) as the reference
        return PackedMoments(tensor, layout, comm)

    def kernel_time(self, name, reset=False):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self._L.w2dyn_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # -- configuration ----------------------------------------------------------------------------------------
    def get_mc_config_scalars(self, walker=0):
        n, o, st = C.c_int(), C.c_int(), C.c_int()
        self._chk(self._L.w2dyn_get_mc_config_scalars(self._h, int(walker), C.byref(n), C.byref(o), C.byref(st)))
        return n.value, o.value, st.value

    def get_mc_config(self, n, walker=0):
        m = max(int(n), 1)
        taus = np.zeros(m); orbs = np.zeros(m, np.int32); spins = np.zeros(m, np.int32); cas = np.zeros(m, np.int32)
        hyb = np.zeros(m, np.int32); o, st = C.c_int(), C.c_int()
        self._chk(self._L.w2dyn_get_mc_config(self._h, int(walker), int(n), taus, orbs, spins, cas, hyb, C.byref(o), C.byref(st)))
        return taus[:n], orbs[:n], spins[:n], cas[:n], hyb[:n], o.value, st.value

    def set_mc_config(self, n, taus, orbs, spins, cas, hashybs, outer_sst, outer_state, walker=-1):
        f = lambda a, t: np.ascontiguousarray(a if n else np.zeros(1), dtype=t)
        self._chk(self._L.w2dyn_set_mc_config(self._h, int(walker), int(n), f(taus, np.float64), f(orbs, np.int32),
                                              f(spins, np.int32), f(cas, np.int32), f(hashybs, np.int32),
                                              int(outer_sst), int(outer_state)))

    def save_mc_configs(self):
        """Every chain's configuration kept inside the context across init_solver calls (CtHybConfig.get_from_ctqmc for all chains)."""
        self._chk(self._L.w2dyn_save_mc_configs(self._h))

    def restore_mc_configs(self):
        self._chk(self._L.w2dyn_restore_mc_configs(self._h))

    def set_empty_mc_config(self, outer_sst, outer_state, walker=-1):
        self.set_mc_config(0, [], [], [], [], [], outer_sst, outer_state, walker)

    def get_minv(self, walker, orb, spin):
        k = C.c_int()
        self._chk(self._L.w2dyn_get_minv(self._h, int(walker), int(orb), int(spin), C.byref(k), None, 0))
        out = np.zeros((k.value, k.value))
        if k.value:
            self._chk(self._L.w2dyn_get_minv(self._h, int(walker), int(orb), int(spin), C.byref(k), out.ctypes.data, out.size))
        return out

    def get_walker_state(self, walker):
        out = np.zeros(6)
        self._chk(self._L.w2dyn_get_walker_state(self._h, int(walker), out))
        return dict(det_sign=out[0], det_log=out[1], window_position=int(out[2]), npairs=int(out[3]), ndraw=int(out[4]), cfgsign=out[5])

    # -- results: module variables ---------------------------------------------------------------------------
    def scalar(self, name):
        v = C.c_double()
        self._chk(self._L.w2dyn_get_scalar(self._h, name.encode(), C.byref(v)))
        return v.value

    def result(self, name, dtype=np.float64):
        need = C.c_size_t()
        self._chk(self._L.w2dyn_get_result(self._h, name.encode(), None, 0, C.byref(need)))
        buf = np.zeros(need.value // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(self._L.w2dyn_get_result(self._h, name.encode(), buf.ctypes.data, buf.nbytes, None))
        return buf

    def _shapes(self):
        nd = self._nd
        nb = int(self.scalar("nbins"))
        g = lambda n: self.result(n).size
        return nd, nb, g

    def __getattr__(self, name):
        # module-variable reads of the reference; evaluated lazily from the library's result table
        if name.startswith("_"):
            raise AttributeError(name)
        if name in _SCALARS:
            return self.scalar(name)
        nd = self.__dict__.get("_nd")
        if nd is None:
            raise AttributeError(name)
        nf = 2 * nd
        if name in ("gtau", "gleg", "histo"):
            return self.result(name).reshape(nd, 2, -1)
        if name == "giw":
            a = self.result("giw").reshape(nd, 2, -1, 2)
            return a[..., 0] + 1j * a[..., 1]
        if name in ("g4iw", "g4iw_pp", "g2iw", "g4iw_bins", "g4iw_pp_bins", "g2iw_bins"):
            # Z-space two-particle estimators (FourPnt = 4): axes as the reference's Fortran arrays
            # g4iw(Nd,2,Nd,2,2*N4iwf,2*N4iwf,2*N4iwb+1), g2iw(Nd,2,n2,n2) [annihilator, creator frequency]
            a = self.result(name)
            a = a.reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            nwf, nwb = self._n4iwf, self._n4iwb
            lead = (int(self.scalar("nb4")),) if name.endswith("_bins") else ()
            if name.startswith("g2iw"):
                n2 = 2 * (nwf + nwb)
                return a.reshape(lead + (nd, 2, n2, n2))
            return a.reshape(lead + (nd, 2, nd, 2, 2 * nwf, 2 * nwf, 2 * nwb + 1))
        if name in ("g4iw_worm", "g4iwpp_worm", "g4iw_worm_bins", "g4iwpp_worm_bins"):
            # worm estimator of the sampled component: (2*N4iwf, 2*N4iwf, 2*N4iwb+1), CTQMC.F90:350, 359
            a = self.result(name).reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            nwf, nwb = self._n4iwf, self._n4iwb
            lead = (int(self.scalar("nbw")),) if name.endswith("_bins") else ()
            return a.reshape(lead + (2 * nwf, 2 * nwf, 2 * nwb + 1))
        if name == "g4tau":       # FourPnt & 1: G4tau(Nd,2,Nd,2,N4tau,N4tau,N4tau), CTQMC.F90:545
            nt = int(self.scalar("n4tau"))
            return self.result(name).reshape(nd, 2, nd, 2, nt, nt, nt)
        if name == "g4leg":       # FourPnt & 2: G4leg(Nd,2,Nd,2,N4tau,N4tau,2*N4iwb+1), filled up to N4leg, CTQMC.F90:549
            nt = int(self.scalar("n4tau"))
            a = self.result(name).reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            return a.reshape(nd, 2, nd, 2, nt, nt, 2 * self._n4iwb + 1)
        if name in ("gtau_blocks", "gtau_blocks_bins"):
            # AccumGtau = 2: gtau_blocks(Nd,2,Ntau,nlevels) (CTQMC.F90:508), mean over the chains (of a bin)
            a = self.result(name)
            nb = int(self.scalar("nbins"))
            if name.endswith("_bins"):
                a = a.reshape(nb, nd, 2, -1, self._ntau)
                return np.ascontiguousarray(a.transpose(0, 1, 2, 4, 3))
            a = a.reshape(nd, 2, -1, self._ntau)
            return np.ascontiguousarray(a.transpose(0, 1, 3, 2))
        if name in ("giw_worm", "giw_worm_bins", "gsigmaiw_worm", "gsigmaiw_worm_bins"):
            # one-particle worm estimators of the sampled component: giw_worm(2*Niw), CTQMC.F90:598; GSigmaiw_worm(2*Niw), sector 3, CTQMC.F90:621
            a = self.result(name).reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            return a.reshape(int(self.scalar("nbins")), -1) if name.endswith("_bins") else a
        if name.split("_bins")[0] in ("p2iw_worm", "p2iwpp_worm", "p3iw_worm", "p3iwpp_worm"):
            # worm estimators of the equal-time sectors 6 - 9: P2iw_worm(-N2iwb:N2iwb), P3iw_worm(-N3iwf:N3iwf-1, -N3iwb:N3iwb), CTQMC.F90:655, 690
            a = self.result(name).reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            lead = (int(self.scalar("nbe")),) if name.endswith("_bins") else ()
            return a.reshape(lead + ((-1, 2 * int(self.scalar("n3iwb")) + 1) if name.startswith("p3") else (-1,)))
        base = name.split("_bins")[0]
        if base in ("h4iw_worm", "qqqq_worm", "qq_worm", "quddag_worm", "nqqdag_worm", "qqdd_worm", "ucaca_worm", "uccaa_worm"):
            # worm estimators of the sectors 5, 10 - 16 (CTQMC.F90:385, 397, 402, 407, 412, 423, 625, 638): H4iw_worm / QQQQ_worm(2*N4iwf, 2*N4iwf,
            # 2*N4iwb+1), QQ_worm / quddag_worm(2*Niw), nqqdag_worm(2*N3iwf, 2*N3iwb+1), qqdd_worm(2*N3iwf, 2*N3iwf), ucaca_worm / uccaa_worm(2*N2iwb+1)
            a = self.result(name).reshape(-1, 2); a = a[:, 0] + 1j * a[:, 1]
            lead = (int(self.scalar("nbie")),) if name.endswith("_bins") else ()
            if base in ("h4iw_worm", "qqqq_worm"):
                shape = (2 * self._n4iwf, 2 * self._n4iwf, 2 * self._n4iwb + 1)
            elif base == "nqqdag_worm":
                shape = (-1, 2 * int(self.scalar("n3iwb")) + 1)
            elif base == "qqdd_worm":
                n = int(round(np.sqrt(a.size // (lead[0] if lead else 1)))); shape = (n, n)
            else:
                shape = (-1,)
            return a.reshape(lead + shape)
        if base in ("qqtau_worm", "ucacatau_worm", "uccaatau_worm"):
            a = self.result(name)
            return a.reshape(int(self.scalar("nbie")), -1) if name.endswith("_bins") else a
        if name.split("_bins")[0] in ("p2tau_worm", "p2taupp_worm"):
            a = self.result(name)
            return a.reshape(int(self.scalar("nbe")), -1) if name.endswith("_bins") else a
        if name in ("gtau_worm", "gtau_worm_bins"):
            a = self.result(name)
            return a.reshape(int(self.scalar("nbins")), -1) if name.endswith("_bins") else a
        if name in ("accwormadd", "accwormrem", "accwormrep"):
            return self.result(name)
        if name == "accpairtau":
            return self.result(name)
        if name == "accpairtau_chains":
            return self.result(name).reshape(self.nwalkers, -1)
        if name == "steps_worm_chains":
            return self.result(name, np.int64)
        if name == "worm_counters":
            return self.result(name).reshape(-1, 8)
        if name in ("ntau_n0", "ntau_n0_bins"):        # MeasSusz: ntau_n0(Nd,2,Nd,2,0:Ntau), CTQMC.F90:514
            a = self.result(name)
            lead = (int(self.scalar("nbs")),) if name.endswith("_bins") else ()
            return a.reshape(lead + (nd, 2, nd, 2, -1))
        if name in ("gtau_full", "gleg_full"):     # (annihilator flavour, creator flavour, ...): filled with offdiag = 1 (CTQMC.F90:2531, 2549)
            return self.result(name).reshape(nd, 2, nd, 2, -1)
        if name in ("gtau_full_bins", "gleg_full_bins"):
            return self.result(name).reshape(int(self.scalar("nbins")), nd, 2, nd, 2, -1)
        if name == "occ":
            return self.result("occ").reshape(nd, 2, nd, 2)
        if name == "rho1":
            return self.result("rho1").reshape(nf, nf)
        if name == "rho2":
            return self.result("rho2").reshape(nf, nf, nf, nf)
        if name == "densitymatrix":
            return self.result("densitymatrix").reshape(4 ** nd, 4 ** nd)
        if name in ("gtau_mean_step", "gtau_mid_step"):  # (Nd, 2, Nmeas): chain-mean series, CTQMC.F90:43, 510-511
            return self.result(name).reshape(nd, 2, -1)
        if name == "sign_step":
            return self.result(name)
        if name == "ssquare":                           # MeasSsquare: Ssquare(0:2*NBands), CTQMC.F90:161, 734
            return self.result("ssquare")
        if name == "ssquare_operator":                  # Ssquare_operator(NStates, NStates), occupation basis, CTQMC.F90:162, 746
            return self.result("ssquare_operator").reshape(4 ** nd, 4 ** nd)
        if name in ("outersuperstatehisto", "outerstatehisto", "signhistosuperstates", "signhistostates",
                    "tracecontribsuperstates", "tracecontribstates", "eigenstatesenergies", "accadd", "accrem",
                    "lhisto", "rhisto"):
            return self.result(name)
        if name == "statessuperstates":
            return self.result(name, np.int32)
        if name == "occbasismapping":
            return self.result(name, np.int32).reshape(4 ** nd, nd, 2)
        if name == "accpair":
            return self.result(name, np.int64).reshape(nd, 2, -1)
        if name == "cntsampling":
            return self.result(name, np.int64)
        if name.endswith("_bins"):
            base = name[:-5]
            nb = int(self.scalar("nbins"))
            a = self.result(name)
            if base in ("gtau", "gleg", "histo"):
                return a.reshape(nb, nd, 2, -1)
            if base == "giw":
                a = a.reshape(nb, nd, 2, -1, 2)
                return a[..., 0] + 1j * a[..., 1]
            if base == "occ":
                return a.reshape(nb, nd, 2, nd, 2)
            return a.reshape(nb, -1)
        raise AttributeError(name)
