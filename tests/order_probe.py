"""Not a test: expansion-order distribution of a workload (which inversion path the global move takes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util
from w2dynamics_amd.mctqmc import MCTQMC
cfg = sys.argv[1]
over = dict(NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nmeas=20, MaxHisto=200)
p, ps, pset = util.setup(cfg, 5001, over)
q = MCTQMC(nwalkers=2048, walkers_per_bin=8)
q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
q.set_taudiffmax(10.0); q.ctqmc_warmup(20000); q.ctqmc_measure(1, 1)
h = q.histo[0, 0]
ks = np.arange(h.size)
print(cfg, "kmax", int(q.scalar("kmax")), "<k> %.1f" % (h * ks).sum(), "P(k>16) %.3f P(k>32) %.3f P(k>48) %.3f" % (h[17:].sum(), h[33:].sum(), h[49:].sum()), "max k", ks[h > 0].max())
q.dest_ctqmc(); q.close()
