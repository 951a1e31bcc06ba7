"""Not a test: throughput probe of the sweep / estimator kernels for several chain counts (run on the GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util
from w2dynamics_amd.mctqmc import MCTQMC

def probe(cfg, nftau, over, nw, kmax=0, rounds=int(os.environ.get('PROBE_ROUNDS', 5)), thermal=20000):
    p, ps, pset = util.setup(cfg, nftau, over)
    q = MCTQMC(nwalkers=nw, walkers_per_bin=8, kmax=kmax)
    q.init_rnd(1); q.init_paras(ps); q.init_solver(p.umatrix, p.ftau, p.muimp, p.screening, 1)
    q.set_timing(True)
    q.set_taudiffmax(over["TaudiffMax"]); q.ctqmc_warmup(thermal); q.measure_begin()
    q.sweep(over["NCorr"]); q.measure_once(); q.sync()
    for k in ("sweep", "meas_gleg", "meas_misc"): q.kernel_time(k, reset=True)
    for _ in range(rounds): q.sweep(over["NCorr"]); q.measure_once()
    q.sync()
    s, _ = q.kernel_time("sweep"); g, _ = q.kernel_time("meas_gleg"); m, _ = q.kernel_time("meas_misc")
    print("%s nw=%6d kmax=%3d sweep %.3f ms (%.1f Msteps/s, %.2f us/step/chain-slot) gleg %.3f ms misc %.3f ms" % (
        cfg, nw, int(q.scalar("kmax")), s, nw * over["NCorr"] / s / 1e3, s * 1e3 / over["NCorr"], g, m), flush=True)
    q.measure_finish()
    c = q.result("counters")
    if len(c) >= 16 and c[12] > 0:
        tot = c[12] + c[13] + c[14] + c[15]
        print("   cycles/move: add %.0f (n=%d) rem %.0f (n=%d) glob %.0f (n=%d) shift %.0f (n=%d); share add %.2f rem %.2f glob %.2f shift %.2f" % (
            c[12] / max(c[0], 1), c[0], c[13] / max(c[3], 1), c[3], c[14] / max(c[6], 1), c[6], c[15] / max(c[9], 1), c[9],
            c[12] / tot, c[13] / tot, c[14] / tot, c[15] / tot), flush=True)
        if len(c) >= 24:
            print("   rem stages (cycles per rem try): window %.0f pick %.0f between %.0f trace+prob %.0f | add: draws+qn %.0f (per try), detratio %.0f trace/np/prob %.0f (per valid=%d), accept-update %.0f (per acc=%d)" % (
                c[16] / c[3], c[17] / c[3], c[18] / c[3], c[19] / c[3], c[20] / c[0], c[21] / max(c[0] - c[2], 1), c[22] / max(c[0] - c[2], 1), c[0] - c[2], c[23] / max(c[1], 1), c[1]), flush=True)
    if os.environ.get("PROBE_ACC") and len(c) >= 24:
        print("   accepted add (cycles per accept, n=%d): ranks+RM/MQ %.0f update+store %.0f det/log %.0f window list %.0f | accepted rem (n=%d): ranks+dnp %.0f row/col %.0f update+store %.0f det/log+window list %.0f" % (
            c[1], c[16] / c[1], c[17] / c[1], c[18] / c[1], c[19] / c[1], c[4], c[20] / c[4], c[21] / c[4], c[22] / c[4], c[23] / c[4]), flush=True)
    if os.environ.get("PROBE_PICK") and len(c) >= 24:
        print("   pick: n=%d, LDS path (nw>64) %.4f, mean nw %.1f, register-path cycles per pick %.0f" % (c[3], c[21] / max(c[3], 1), c[22] / max(c[3], 1), c[23] / max(c[3] - c[21], 1)), flush=True)
    if os.environ.get("PROBE_SEQ") and len(c) >= 24:
        print("   cached-window trace: calls %d, mean operators %.1f, full evaluations %.4f, rebuilds of the outside product %.4f (%.1f operators, %.0f cycles each), window part %.1f operators, %.0f cycles" % (
            c[16], c[23] / max(c[16], 1), c[22] / max(c[16], 1), c[17] / max(c[16], 1), c[18] / max(c[17], 1), c[20] / max(c[17], 1), c[19] / max(c[16] - c[22], 1), c[21] / max(c[16] - c[22], 1)), flush=True)
    if os.environ.get("PROBE_GLOB") and len(c) >= 24:
        print("   glob stages (cycles per global move): setup+qn %.0f invert %.0f parity %.0f logtrace %.0f install %.0f (acc=%d of %d)" % (
            c[16] / max(c[6], 1), c[17] / max(c[6], 1), c[18] / max(c[6], 1), c[19] / max(c[6], 1), c[20] / max(c[6], 1), c[7], c[6]), flush=True)
    q.dest_ctqmc(); q.close()

if __name__ == "__main__":
    args = sys.argv[1:]
    cfg = "C2"
    if args and not args[0].isdigit():
        cfg = args.pop(0)
    over = dict(NCorr=200 if cfg == "C2" else 100, TaudiffMax=10.0, Ntau=1000, NLegMax=100)
    if os.environ.get("PROBE_NORARE"):      # no global / tau-shift moves (register experiments with those moves compiled out)
        over.update(PercentageGlobalMove=0.0, PercentageTauShiftMove=0.0)
    for nw in [int(x) for x in (args or ["256", "2048", "8192", "16384"])]:
        probe(cfg, 5001, over, nw)
