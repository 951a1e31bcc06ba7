import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, util, pyoracle
from w2dynamics_amd.mctqmc import MCTQMC
g = util.load_golden("worm_C4Sa.npz"); over0 = g["overrides"]; comp = int(g["component"])
sector = 4; W = 1  # chain index
def run(nfix):
    over = dict(over0, Nmeas=1, NCorr=nfix)
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=3, walkers_per_bin=1); q.fourpnt = 8
    q.init_rnd(21); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(sector, comp)
    wc = q.worm_counters[W]; cnt = q.result("counters")
    n, osst, _ = q.get_mc_config_scalars(W); cfg = q.get_mc_config(n, W)
    st = q.get_walker_state(W)
    q.dest_ctqmc(); q.close()
    o = pyoracle.OracleWalker(p, pset, ("philox", 21, W), fourpnt=8)
    o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
    o.measure_sector(sector, comp)
    r = o.results(); oc = o.worm_counters(); c = o.get_config(); so = o.state()
    got = [r["counters"][11], oc["cnt_sampling_w"], oc["try_wadd"], oc["acc_wadd"], oc["try_wrem"], oc["acc_wrem"], oc["try_wrep"], oc["acc_wrep"]]
    o.close()
    same = np.array_equal(wc, got) and n == len(c["taus"]) and (n == 0 or np.abs(cfg[0] - c["taus"]).max() < 1e-10)
    return same, (wc, cfg, osst, st), (got, c, oc, r["counters"], so)
lo, hi = 400, 800
while hi - lo > 1:
    mid = (lo + hi) // 2
    if run(mid)[0]: lo = mid
    else: hi = mid
print("last good", lo, "first bad", hi)
for nf in (lo, hi):
    same, G, O = run(nf)
    print("nfix", nf, same)
    print("  gpu worm ctr", G[0], "outer", G[2], "state", G[3])
    print("  ora worm ctr", O[0], "outer", O[1]["outer_sst"], "pairs in worm: try_add", O[2]["wtry_add"], O[2]["wacc_add"], O[2]["wtry_rem"], O[2]["wacc_rem"], "Z ctr", O[3], "state", O[4])
    print("  gpu taus", np.round(G[1][0], 6), list(G[1][4]), list(G[1][1]), list(G[1][2]), list(G[1][3]))
    print("  ora taus", np.round(O[1]["taus"], 6), list(O[1]["hashybs"]), list(O[1]["orbs"]), list(O[1]["spins"]), list(O[1]["cas"]))
