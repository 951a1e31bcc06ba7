"""Not a test: look for chains whose measured configuration sign is zero (matrix path) and check them against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import util, pyoracle
from w2dynamics_amd.mctqmc import MCTQMC
over = dict(Nwarmups=0, Nmeas=1, NCorr=0, TaudiffMax=10.0, Ntau=200, NLegMax=40, Niw=8, Nftau=5001)
p, ps, pset = util.setup("C5KB", 5001, over)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
q = MCTQMC(nwalkers=W, first_walker=0, walkers_per_bin=1)
q.init_rnd(4321); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
q.set_taudiffmax(10.0); q.ctqmc_warmup(int(sys.argv[2]) if len(sys.argv) > 2 else 500)
q.ctqmc_measure(1, 1)
sg = np.nansum(q.signhistosuperstates_bins * q.outersuperstatehisto_bins, axis=-1); vis = q.outersuperstatehisto_bins.sum(axis=-1)
print("sg", np.unique(np.round(sg, 6))[:8], "vis", np.unique(np.round(vis, 6))[:8], flush=True)
bad = np.nonzero(sg > 0)[0]
print("mean_sign", q.mean_sign, "chains with |sign| != 1:", bad[:20], len(bad), flush=True)
ns = np.array([q.get_mc_config_scalars(int(w))[0] for w in range(0, W, 16)])
print("operators per chain: mean %.1f max %d" % (ns.mean(), ns.max()), flush=True)
for w in bad[:4]:
    w = int(w)
    st = q.get_walker_state(w); n, osst, ost = q.get_mc_config_scalars(w)
    g = q.get_mc_config(n, w)
    nk = int((g[0] <= 25.0).sum())
    print("  chain", w, "nops", n, "ops up to beta/2:", nk, "outer", osst, st, flush=True)
    o = pyoracle.OracleWalker(p, pset, ("philox", 4321, w))
    o.set_taudiffmax(10.0)
    rc = o.set_config(dict(taus=g[0], orbs=g[1], spins=g[2], cas=g[3], hashybs=g[4], outer_sst=osst, outer_state=ost))
    print("  oracle set_config rc", rc, "state", o.state(), flush=True)
    o.close()
q.dest_ctqmc(); q.close()
