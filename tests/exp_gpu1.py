import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import util, pyoracle
from w2dynamics_amd.mctqmc import MCTQMC

def run(cfgname, nftau, over, seed, nw=8):
    p, ps, pset = util.setup(cfgname, nftau, over)
    q = MCTQMC(nwalkers=nw, walkers_per_bin=1)
    q.init_rnd(seed); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    tdm = pset.get_real("TaudiffMax"); q.set_taudiffmax(tdm)
    t=time.time(); q.ctqmc_warmup(over["Nwarmups"]); q.sync(); print('gpu warmup', time.time()-t)
    ok = True
    for w in range(nw):
        o = pyoracle.OracleWalker(p, pset, ("philox", seed, w)); o.set_taudiffmax(tdm); o.warmup(over["Nwarmups"])
        c = o.get_config(); n, osst, _ = q.get_mc_config_scalars(w)
        st = q.get_walker_state(w); so = o.state()
        if n != len(c["taus"]):
            print('walker', w, 'nops gpu', n, 'oracle', len(c['taus']), 'ndraw', st['ndraw']); ok = False; continue
        g = q.get_mc_config(n, w)
        dt = np.abs(g[0]-c['taus']).max() if n else 0
        dm = 0
        for orb in range(1, p.nd+1):
            for sp in (1,2):
                a = q.get_minv(w, orb, sp); b = o.minv(orb, sp)
                if a.size: dm = max(dm, np.abs(a-b).max()/max(np.abs(b).max(),1e-300))
        print('walker', w, 'nops', n, 'dtau', dt, 'dM rel', dm, 'outer', osst, c['outer_sst'], 'detsign', st['det_sign'], so['det_sign'], 'wpos', st['window_position'], so['window_position'], 'np', st['npairs'], so['npairs'])
        o.close()
    # measurement
    q.ctqmc_measure(1,1)
    for w in range(min(nw,3)):
        o = pyoracle.OracleWalker(p, pset, ("philox", seed, w)); o.set_taudiffmax(tdm); o.warmup(over["Nwarmups"]); o.measure(); r = o.results()
        for k in ('gleg','gtau','giw','histo','occ'):
            a = getattr(q, k+'_bins')[w]; b = r[k]
            print('  w%d %-6s max %.3e diff %.3e' % (w, k, np.abs(b).max(), np.abs(a-b).max()))
        o.close()
    print('mean_sign', q.mean_sign, 'accadd', q.accadd[0], 'accrem', q.accrem[0], 'accglob', q.accglob, 'accshift', q.accshift)
    for k in ('sweep','meas_misc','meas_gleg','meas_giw'): print(k, q.kernel_time(k))
    q.dest_ctqmc(); q.close()

which = sys.argv[1] if len(sys.argv)>1 else 'a'
base = dict(Ntau=100, NLegMax=12, Niw=8, MeasGiw=1, NLookup_nfft=0)
if which in ('a','all'):
    run('C1', 501, dict(base, Nwarmups=3000, Nmeas=50, NCorr=20, TaudiffMax=2.0, PercentageGlobalMove=0.0, PercentageTauShiftMove=0.0), 1234)
if which in ('b','all'):
    run('C3', 501, dict(base, Nwarmups=3000, Nmeas=50, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.0, PercentageTauShiftMove=0.1), 77)
if which in ('c','all'):
    run('C3', 501, dict(base, Nwarmups=3000, Nmeas=50, NCorr=30, TaudiffMax=6.0, PercentageGlobalMove=0.1, PercentageTauShiftMove=0.1), 77)
    run('C2', 2001, dict(base, Nwarmups=20000, Nmeas=20, NCorr=200, TaudiffMax=10.0, NLegMax=100), 5)
