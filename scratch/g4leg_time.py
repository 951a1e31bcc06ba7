import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, util
from w2dynamics_amd.mctqmc import MCTQMC
nw = int(sys.argv[1]); nl = int(sys.argv[2]); nwb = int(sys.argv[3])
over = dict(Nwarmups=0, Nmeas=1, NCorr=200, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nftau=5001, N4tau=nl, N4leg=nl, N4iwb=nwb)
p, ps, pset = util.setup("C2", 5001, over)
q = MCTQMC(nwalkers=nw, walkers_per_bin=8); q.fourpnt = 2
q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
q.set_taudiffmax(10.0); q.ctqmc_warmup(20000); q.measure_begin()
q.sweep(200); q.measure_once(); q.sync(); q.kernel_time("meas_g4leg", reset=True)
for _ in range(3): q.sweep(200); q.measure_once()
q.sync(); t, n = q.kernel_time("meas_g4leg"); q.measure_finish()
k = q.histo; ks = np.arange(k.shape[-1]); k1 = (k * ks).sum(-1).mean(); k4 = ((k * ks ** 2).sum(-1) ** 2).mean()
flop = nw * 4 * k4 * (2 * nl * nl)      # what the unfactorised exchange sums (k^4 nl^2 FMAs) would cost, for comparison
print("g4leg chains=%d N4leg=%d N4iwb=%d <k>=%.1f: %.3f ms per pass (%d launches), equivalent to %.2f TFLOP/s of unfactorised k^4 nl^2 exchange sums, atomics=%s" % (nw, nl, nwb, k1, t, n, flop / (t * 1e-3) / 1e12, os.environ.get("W2DYN_G4LEG_ATOMICS", "")))
q.dest_ctqmc(); q.close()
