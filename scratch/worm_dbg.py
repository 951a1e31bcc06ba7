import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from w2dynamics_amd import params, synth
from w2dynamics_amd.mctqmc import MCTQMC
W = bench.WORKLOADS["C5"]; over = W["overrides"]
prob = synth.config(W["config"], nftau=W["nftau"])
ps = params.make_pstring(over, beta=prob.beta, Nd=prob.nd, Hamiltonian=prob.hamiltonian, QuantumNumbers=prob.quantum_numbers)
q = MCTQMC(nwalkers=1024, walkers_per_bin=8); q.fourpnt = 8
q.init_rnd(1); q.init_paras(ps); q.init_solver(prob.umatrix, prob.ftau, prob.muimp, prob.screening, 1)
q.set_taudiffmax(10.0); q.ctqmc_warmup(20000)
for comp in (12, 1):
    q.wormeta[:] = 1.0
    for it in range(8):
        sw, sz = q.count_steps(4, comp, 2000)
        ch = q.steps_worm_chains
        print(comp, it, "eta", q.wormeta[2], "sw", sw, "sz", sz, "chains in worm>0:", (ch > 0).sum(), "max", ch.max(), flush=True)
        if sw == 0: break
        q.wormeta[:] = q.wormeta[:] * sz / sw
