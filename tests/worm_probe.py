"""Not a test: sweep time and (diagnostic build -DW2DYN_STAMPS_WORM) cycles per move type of the 5-orbital worm workload."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from w2dynamics_amd import synth, params
from w2dynamics_amd.mctqmc import MCTQMC
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
over = dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1, N4iwf=10, N4iwb=5,
            PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, G4ph=1, G4pp=0)
prob = synth.config("C5Z", nftau=5001)
ps = params.make_pstring(over, beta=prob.beta, Nd=prob.nd, Hamiltonian=prob.hamiltonian, QuantumNumbers=prob.quantum_numbers)
q = MCTQMC(nwalkers=nw, walkers_per_bin=8)
q.fourpnt = 8
q.init_rnd(43890); q.init_paras(ps); q.init_solver(prob.umatrix, prob.ftau, prob.muimp, prob.screening, 1)
q.set_taudiffmax(10.0); q.ctqmc_warmup(100)
for _ in range(199): q.sweep(100)
q.wormeta[:] = 1.0 / (prob.beta * prob.nd) ** 4
for _ in range(12):
    sw, sz = q.count_steps(4, 12, 2000)
    if sw == 0 or sz == 0 or abs(sw - sz) / float(sw + sz) <= 0.1: break
    q.wormeta[:] = q.wormeta[:] * float(sz) / float(sw)
q.set_sector(4, 12); q.set_timing(True); q.measure_begin()
q.sweep(100); q.measure_once(); q.sync()
for k in ("sweep_worm", "meas_misc", "meas_g4w"): q.kernel_time(k, reset=True)
for _ in range(5): q.sweep(100); q.measure_once()
q.sync()
s, _ = q.kernel_time("sweep_worm"); m, _ = q.kernel_time("meas_misc"); g, _ = q.kernel_time("meas_g4w")
print("C5 worm nw=%d sweep %.3f ms (%.1f Msteps/s) misc %.3f g4w %.3f ms" % (nw, s, nw * 100 / s / 1e3, m, g), flush=True)
q.measure_finish()
c = q.result("counters"); wc = q.worm_counters.sum(axis=0) if hasattr(q, "worm_counters") else None
print("counters[:24]", [int(v) for v in c[:24]], flush=True)
print("worm counters", wc, flush=True)
q.dest_ctqmc(); q.close()
