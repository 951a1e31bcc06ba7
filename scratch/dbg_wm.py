import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, util, pyoracle
from w2dynamics_amd.mctqmc import MCTQMC
g = util.load_golden("worm_C4Sa.npz"); over0 = g["overrides"]; comp = int(g["component"])
sector = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for nfix in (400, 800, 1600, 3000):
    over = dict(over0, Nmeas=1, NCorr=nfix)
    p, ps, pset = util.setup(g["config"], g["nftau"], over)
    q = MCTQMC(nwalkers=3, walkers_per_bin=1); q.fourpnt = 8
    q.init_rnd(21); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"])
    q.ctqmc_measure(sector, comp)
    wc = q.worm_counters
    bad = False
    for w in range(3):
        o = pyoracle.OracleWalker(p, pset, ("philox", 21, w), fourpnt=8)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
        o.measure_sector(sector, comp)
        r = o.results(); oc = o.worm_counters(); c = o.get_config()
        got = [r["counters"][11], oc["cnt_sampling_w"], oc["try_wadd"], oc["acc_wadd"], oc["try_wrem"], oc["acc_wrem"], oc["try_wrep"], oc["acc_wrep"]]
        n, osst, _ = q.get_mc_config_scalars(w)
        cfg = q.get_mc_config(n, w)
        same = np.array_equal(wc[w], got) and n == len(c["taus"]) and (n == 0 or np.abs(cfg[0] - c["taus"]).max() < 1e-10)
        if not same:
            bad = True
            print("nfix", nfix, "chain", w, "gpu", wc[w], "oracle", got, "wtry_add/acc/rem/acc", oc["wtry_add"], oc["wacc_add"], oc["wtry_rem"], oc["wacc_rem"], "nops", n, len(c["taus"]))
            print("   gpu taus", np.round(cfg[0], 5), list(cfg[4]))
            print("   ora taus", np.round(c["taus"], 5), list(c["hashybs"]))
        o.close()
    q.dest_ctqmc(); q.close()
    print("nfix", nfix, "DIFF" if bad else "ok", flush=True)
    if bad: break
print("count_steps check")
over = dict(over0)
p, ps, pset = util.setup(g["config"], g["nftau"], over)
for nfix in (50, 200, 1000, 3000):
    q = MCTQMC(nwalkers=3, walkers_per_bin=1); q.fourpnt = 8
    q.init_rnd(21); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(over["Nwarmups"])
    sw, sz = q.count_steps(sector, comp, nfix)
    ch = list(q.steps_worm_chains)
    oo = []
    for w in range(3):
        o = pyoracle.OracleWalker(p, pset, ("philox", 21, w), fourpnt=8)
        o.set_taudiffmax(over["TaudiffMax"]); o.warmup(over["Nwarmups"])
        oo.append(o.count_steps(sector, comp, nfix)[0]); o.close()
    print("nfix", nfix, "gpu", ch, "oracle", oo, flush=True)
    q.dest_ctqmc(); q.close()
