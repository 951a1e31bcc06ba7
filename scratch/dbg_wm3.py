import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, util, pyoracle
from w2dynamics_amd.mctqmc import MCTQMC
g = util.load_golden("worm_C4Sa.npz"); over = dict(g["overrides"]); comp = int(g["component"])
sector = 4; NW = 3
p, ps, pset = util.setup(g["config"], g["nftau"], over)
q = MCTQMC(nwalkers=NW, walkers_per_bin=1); q.fourpnt = 8
q.init_rnd(21); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"])
q.measure_begin(); q.set_sector(sector, comp)
oo = []
for w in range(NW):
    o = pyoracle.OracleWalker(p, pset, ("philox", 21, w), fourpnt=8)
    o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"]); o.measure_begin(); o.set_sector(sector, comp)
    oo.append(o)
def snap_g(w):
    n, osst, _ = q.get_mc_config_scalars(w); c = q.get_mc_config(n, w)
    return (np.array(c[0]), list(c[1]), list(c[2]), list(c[3]), list(c[4]), osst, q.get_walker_state(w))
def snap_o(o):
    c = o.get_config(); return (c["taus"], list(c["orbs"]), list(c["spins"]), list(c["cas"]), list(c["hashybs"]), c["outer_sst"], o.state())
prev = [(snap_g(w), snap_o(oo[w])) for w in range(NW)]
done = set()
for step in range(1, 1500):
    q.sweep(1); q.sync()
    for w in range(NW):
        if w in done: continue
        oo[w].sweep(1)
        G, O = snap_g(w), snap_o(oo[w])
        nd_o = oo[w].L.ora_rng_philox_ndraw(oo[w].rng)
        if G[6]["ndraw"] != nd_o and w not in done:
            wc = oo[w].worm_counters()
            print("chain", w, "step", step, "ndraw gpu", G[6]["ndraw"], "oracle", nd_o, "prev gpu ndraw", prev[w][0][6]["ndraw"], "oracle worm ctr", wc, "Z", oo[w].results()["counters"])
            print("  before", np.round(prev[w][1][0], 5), prev[w][1][1], prev[w][1][2], prev[w][1][3], prev[w][1][4], prev[w][1][6])
            print("  after ", np.round(O[0], 5), O[4], O[6])
            done.add(w); continue
        same = len(G[0]) == len(O[0]) and (len(G[0]) == 0 or np.abs(G[0] - O[0]).max() < 1e-10) and G[4] == O[4] and G[5] == O[5]
        if not same:
            print("chain", w, "first difference at step", step)
            for tag, S in (("before gpu", prev[w][0]), ("before ora", prev[w][1]), ("after gpu", G), ("after ora", O)):
                print(" ", tag, np.round(S[0], 6), "orb", S[1], "sp", S[2], "ca", S[3], "hyb", S[4], "outer", S[5], S[6])
            done.add(w)
        prev[w] = (G, O)
    if len(done) == NW: break
print("done", done)
