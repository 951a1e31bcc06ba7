#!/usr/bin/env python3
"""Benchmark of the CT-HYB hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Headline workload (BASELINE.json configs[1]): 1-band Hubbard model, Bethe lattice D=1, U=2, beta=100, half filling,
TaudiffMax=10 (9 sliding windows), reference defaults otherwise (Ntau=1000, NLegMax=100, Nftau=5001,
PercentageGlobalMove = PercentageTauShiftMove = 0.005) and NCorr=200 as in BASELINE.md.
One bench "step" = one pass of the hot path over the batch of chains of every GPU: NCorr Monte-Carlo steps for each
chain (sweep kernel) followed by one pass of the estimators the workload switches on (sign/histograms, Legendre + tau
bins; Matsubara G(iw), g2iw/g4iw; worm estimator).
`value` = Metropolis steps per second of the whole job = n_gpus * chains * NCorr * K / time; measurements/s is
reported next to it.  Chains are thermalised before the timed region; all state is resident in HBM.

The default invocation (no --config) also runs the other BASELINE.json configurations (C1, C3, C4, C5 worm) and the
two Matsubara-estimator workloads (C2G, C3F) at 16384 / 4096 chains for a few steps each and reports them under
"configs" of the same JSON line, each with its roofline objects and (at N = 1) its host-CPU baseline; the headline
numbers of the line are those of C2, which runs last.  `--config X` runs X alone.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector == matrix peak (the microarch guide lists no fp64 row)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8 TB/s

_BASE = dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1)
_WORM = dict(N4iwf=10, N4iwb=5, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, G4ph=1, G4pp=0)
WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (default)
    "C2": dict(config="C2", nftau=5001, overrides=dict(_BASE, NCorr=200),
               text="C2: 1-band Hubbard, Bethe D=1, U=2, beta=100, half filling, TaudiffMax=10, NCorr=200, NLegMax=100, Ntau=1000, Nftau=5001, default move mix"),
    # C2 with the autocorrelation accumulator of G(tau) (AccumGtau = 2, one stack of log2(Nmeas) levels per chain): adds the
    # HBM-streaming kernel k_meas_gtau_ac to every step; extra bench line, not the headline
    "C2A": dict(config="C2", nftau=5001, overrides=dict(_BASE, NCorr=200, Nmeas=1024, AccumGtau=2),
                text="C2 + AccumGtau=2 (per-chain autocorrelation accumulator of G(tau), 10 levels)"),
    # C2 with the direct Matsubara estimator at the reference's default Niw = 2000 (k_meas_giw)
    "C2G": dict(config="C2", nftau=5001, chains=16384, overrides=dict(_BASE, NCorr=200, MeasGiw=1, Niw=2000),
                text="C2G: C2 + MeasGiw=1 with Niw=2000 (direct Matsubara G(iw) estimator, exact sums)"),
    # C3 with the Z-space two-particle Matsubara estimators at a production frequency box (k_meas_g2iw, k_meas_g4iw)
    "C3F": dict(config="C3", nftau=5001, chains=4096, fourpnt=4, overrides=dict(_BASE, N4iwf=30, N4iwb=10, MeasG2iw=1),
                text="C3F: C3 + FourPnt=4 (Z-space g2iw / g4iw, N4iwf=30, N4iwb=10: 1.2e6 complex bins for 2 orbitals)"),
    # the other single-GPU configurations
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (beta = 10: expansion order ~1.4 per flavour)
    "C1": dict(config="C1", nftau=5001, chains=16384, overrides=dict(_BASE),
               text="C1: 1-band Hubbard, Bethe D=1, U=2, beta=10, half filling, one window, NCorr=100, NLegMax=100, Ntau=1000"),
    "C3": dict(config="C3", nftau=5001, chains=16384, overrides=dict(_BASE),
               text="C3: 2-orbital density-density, U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    "C4": dict(config="C4", nftau=5001, chains=16384, overrides=dict(_BASE),
               text="C4: 3-orbital Kanamori (QNs Nt Szt Qzt, matrix local trace), U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    "C5Z": dict(config="C5Z", nftau=5001, chains=16384, overrides=dict(_BASE),
                text="C5 (Z-space part): 5-orbital density-density, U=2, J=0.25, beta=100, TaudiffMax=10, NCorr=100"),
    "C5K": dict(config="C5KB", nftau=5001, chains=16384, overrides=dict(_BASE),
                text="C5K: 5-orbital Kanamori (d shell, QNs Nt Szt Qzt: 352 blocks of up to 10 states, matrix local trace), U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    # C4 with the moves outside the default mix (k_sweep<3, ..., XM>: two pairs at once, pair moves that change the outer block)
    "C4X": dict(config="C4", nftau=5001, chains=16384, overrides=dict(_BASE, Percentage4OperatorMove=0.05, PercentageOuterMove=0.05),
                text="C4X: C4 with Percentage4OperatorMove=0.05 and PercentageOuterMove=0.05"),
    # off-diagonal hybridisation (offdiag = 1, k_sweep_od / k_meas_full): 3-orbital Kanamori with inter-orbital hopping and F(tau) between the orbitals
    "C4O": dict(config="C4OB", nftau=5001, chains=8192, overrides=dict(_BASE, offdiag=1),
                text="C4O: 3-orbital Kanamori with inter-orbital hopping 0.2 and hybridisation 0.25 F(tau) between neighbouring orbitals (QNs Nt All), "
                     "offdiag=1, beta=50, TaudiffMax=10, NCorr=100"),
    # worm sampling on the matrix trace path: 3-orbital Kanamori, component 8 = flavours 0011
    "C4W": dict(config="C4", nftau=5001, chains=8192, worm=dict(component=8, nfix=2000), overrides=dict(_BASE, **_WORM),
                text="C4W: 3-orbital Kanamori (matrix local trace), beta=50, worm sampling in sector 4 (G4iw, component 8 = flavours 0011, "
                     "N4iwf=10, N4iwb=5, PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
    # BASELINE.json configs[4] on one GPU: worm sampling of one component of the two-particle Green's function (sector 4)
    "C5": dict(config="C5Z", nftau=5001, chains=16384, worm=dict(component=12, nfix=2000), overrides=dict(_BASE, **_WORM),
               text="C5: 5-orbital density-density, U=2, J=0.25, beta=100, TaudiffMax=10, NCorr=100, worm sampling in sector 4 (G4iw, "
                    "component 12 = flavours 0011, N4iwf=10, N4iwb=5, PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
    # sector 2: worm sampling of one diagonal component of the one-particle Green's function at the reference's default Niw (k_meas_gw)
    "C2S": dict(config="C2", nftau=5001, chains=16384, worm=dict(sector=2, component=1, nfix=2000),
                overrides=dict(_BASE, NCorr=200, Niw=2000, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasGiw=1, WormMeasGtau=1),
                text="C2S: C2 with worm sampling in sector 2 (giw_worm / gtau_worm of component 1 = flavours 00, Niw=2000, "
                     "PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
    # sector 10: the symmetric improved estimator (QQ: two three-operator equal-time objects with U-matrix factors, six worm operators;
    # k_sweep_worm<6, 0, IE>, k_meas_wie) at the reference's default Niw
    "C2Q": dict(config="C2", nftau=5001, chains=16384, worm=dict(sector=10, component=1, nfix=2000),
                overrides=dict(_BASE, NCorr=200, Niw=2000, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasQQ=1, WormMeasQQtau=1),
                text="C2Q: C2 with worm sampling in sector 10 (qq_worm / qqtau_worm of component 1, the symmetric improved estimator, Niw=2000, "
                     "PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
    # sector 12: eight worm operators (two three-operator objects and an equal-time pair; k_sweep_worm<8, 0, IE>, closed-form local trace)
    "C2N": dict(config="C2", nftau=5001, chains=16384, worm=dict(sector=12, component=4, nfix=2000),
                overrides=dict(_BASE, NCorr=200, N3iwf=10, N3iwb=5, PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasNQQdag=1),
                text="C2N: C2 with worm sampling in sector 12 (nqqdag_worm of component 4 = flavours 0011, eight worm operators, N3iwf=10, N3iwb=5, "
                     "PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
}
HEADLINE = "C2"
SIDE_WORKLOADS = ["C1", "C3", "C4", "C5", "C2G", "C3F", "C2Q"]      # run ahead of the headline by the default invocation


def _cpu_chain(args):
    """One reference (or oracle) chain on one core; returns (mc_steps, seconds, kind, gleg)."""
    seed, nmeas, nwarm, wname, eta = args
    wl = dict(WORKLOADS[wname])
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from w2dynamics_amd import params, synth
    p = synth.config(wl["config"], nftau=wl["nftau"])
    over = dict(wl["overrides"]); over.update(Nmeas=nmeas, Nwarmups=nwarm)
    if over.get("MeasGiw"):
        over["NLookup_nfft"] = 0          # the reference built without NFFT takes the exact-sum transform (CTQMC.F90:578-595)
    ps = params.make_pstring(over, beta=p.beta, Nd=p.nd, Hamiltonian=p.hamiltonian, QuantumNumbers=p.quantum_numbers)
    worm = wl.get("worm")
    wsec = worm.get("sector", 4) if worm else 1
    fourpnt = (8 if wsec in (4, 5, 11) else 0) if worm else wl.get("fourpnt", 0)
    try:
        import refdrive
        if not refdrive.available():
            raise ImportError("oracle/_ref not built")
        s = refdrive.RefSolver(p, ps, seed, fourpnt=fourpnt)
        s.set_taudiffmax(over["TaudiffMax"])
        s.warmup(nwarm)
        if worm:
            s.set_wormeta(eta)
            s.worm_warmup(0, wsec, worm["component"])
        t0 = time.perf_counter(); s.measure(*((wsec, worm["component"]) if worm else (1, 1))); dt = time.perf_counter() - t0
        gl = None if worm else np.array(s.results(("gleg",))["gleg"], dtype=np.float64)
        s.close()
        return nmeas * over["NCorr"], dt, "reference", gl
    except Exception:
        import pyoracle
        w = pyoracle.OracleWalker(p, params.ParameterSet(ps), ("std", seed), fourpnt=fourpnt)
        w.set_taudiffmax(over["TaudiffMax"])
        w.warmup(nwarm)
        if worm:
            w.set_wormeta(eta)
        t0 = time.perf_counter()
        if worm:
            w.measure_sector(wsec, worm["component"])
        else:
            w.measure()
        dt = time.perf_counter() - t0
        gl = None if worm else np.array(w.results()["gleg"], dtype=np.float64)
        w.close()
        return nmeas * over["NCorr"], dt, "port", gl


# measurements per CPU chain: about 10-20 s of CPU work for the headline, a few seconds for the side workloads (the
# reference's G(iw) / g4iw estimators cost milliseconds to seconds per measurement: far fewer of them)
_CPU_NMEAS = {"C2Q": 3000, "C2N": 3000, "C4X": 1000, "C4O": 300, "C2": 6000, "C2A": 6000, "C2G": 200, "C3F": 30, "C1": 3000, "C3": 2000, "C4": 1000, "C5Z": 1000, "C5K": 300, "C4W": 1000, "C5": 1000}


def cpu_baseline(cores, wname, nwarm=50000, eta=1.0):
    import multiprocessing as mp
    nmeas = _CPU_NMEAS.get(wname, 1000)
    if wname != HEADLINE:
        nwarm = 20000
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_chain, [(1000 + i, nmeas, nwarm, wname, eta) for i in range(cores)])
    rate = sum(s / t for s, t, _, _ in res)
    kind = res[0][2]
    gerr = None
    if res[0][3] is not None and cores > 1:      # error of G(l) over the chains (one per core), and the wall time it took
        gl = np.array([r[3] for r in res])
        gerr = {"max_stderr_l_lt_20": float((gl.std(axis=0, ddof=1) / np.sqrt(cores))[..., :20].max()),
                "wall_s": float(max(t for _, t, _, _ in res))}
    ncorr = WORKLOADS[wname]["overrides"]["NCorr"]
    return {"value": rate, "unit": "MC steps/s", "cores": cores, "kind": kind, "gleg_err": gerr,
            "meas_per_s": sum((s / ncorr) / t for s, t, _, _ in res),
            "sample": "%d independent chains (one per core, as the reference's MPI ranks), each %d warm-up steps untimed + "
                      "%d measurements x NCorr=%d steps timed, same %s workload" % (cores, nwarm, nmeas, ncorr, wname)}


def issue_profile(config):
    """Instruction mix of the sweep kernel from the committed rocprofv3 --pmc passes (profiles/*_sweep_instmix.json, made
    by profiles/instmix.sh): the kernel is bound by vector-instruction issue and dependent-chain latency, not by HBM."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sweep_instmix.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config", "C2") == config:
            return d, os.path.basename(f)
    return None, None


def pmc_traffic(kernel, chains, config, sum_all=False):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*_pmc.json, made by
    profiles/collect.sh + pmc_summary.py on the same command); None when no pass exists for this chain count.
    sum_all: the timed step is several kernels whose names start with `kernel` (the g4iw assembly), add them up."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("chains_per_gpu") != chains or d.get("config", "C2") != config:
            continue
        hits = [v["traffic"] for name, v in d.get("kernels", {}).items() if name.startswith(kernel)]
        if hits:
            return (sum(hits) if sum_all else hits[0]), os.path.basename(f)
    return None, None


class Env:
    """Process-wide state of one bench invocation (rank layout, communicator)."""
    pass


def run_workload(env, wname, walkers, steps, warmup, thermal, wpb, with_cpu, cpu_cores=0, fixed_eta=None):
    """Thermalise, time `steps` passes of the hot path of workload `wname`, reduce; rank 0 returns the result dict."""
    import torch
    import torch.distributed as dist
    from w2dynamics_amd import params, synth
    from w2dynamics_amd.mctqmc import MCTQMC
    from w2dynamics_amd.statistics import DistributedSample

    wl = WORKLOADS[wname]
    world, rank, comm, red_dev, rehearse = env.world, env.rank, env.comm, env.red_dev, env.rehearse
    over = wl["overrides"]
    prob = synth.config(wl["config"], nftau=wl["nftau"])
    ps = params.make_pstring(over, beta=prob.beta, Nd=prob.nd, Hamiltonian=prob.hamiltonian, QuantumNumbers=prob.quantum_numbers)
    q = MCTQMC(device=env.local_rank, nwalkers=walkers, first_walker=rank * walkers, walkers_per_bin=wpb)
    worm = wl.get("worm")
    wsec = worm.get("sector", 4) if worm else 1
    q.fourpnt = (8 if wsec in (4, 5, 11) else 0) if worm else wl.get("fourpnt", 0)
    q.init_rnd(43890); q.init_paras(ps)
    q.init_solver(prob.umatrix, prob.ftau, prob.muimp, prob.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"])
    ncorr = over["NCorr"]
    q.ctqmc_warmup(ncorr)                          # thermalise (untimed), in launches of the same size as the timed ones
    for _ in range(max(thermal // ncorr - 1, 0)):
        q.sweep(ncorr)
    eta, eta_tries = 1.0, 0
    if worm:
        if fixed_eta is not None:
            q.wormeta[:] = fixed_eta
        else:
            # eta search of solve_component (impurity.py:717-742): rescale until Z and worm space get equal numbers of
            # steps, to the reference's criterion |sw - sz| / (sw + sz) <= 0.1; starting point: insertion prefactor
            # (beta Nd)^4 eta = 1 (CTQMC.F90:5680-5684).  The reference runs this loop without a bound on its length
            q.wormeta[:] = 1.0 / (prob.beta * prob.nd) ** (2 if wsec in (2, 10) else 4)       # sector 12: beta^3 Nd^8 eta; the search rescales from wherever it starts
            for eta_tries in range(1, 101):
                sw, sz = q.count_steps(wsec, worm["component"], worm["nfix"])
                if sw != 0 and sz != 0 and abs(sw - sz) / float(sw + sz) <= 0.1:
                    break
                q.wormeta[:] = q.wormeta[:] * float(max(sz, 1)) / float(max(sw, 1))
        eta = float(q.wormeta[2])
        if world > 1:      # all ranks use the mean eta (impurity.py:739-740, 784)
            t = torch.tensor([eta], dtype=torch.float64, device=red_dev); dist.all_reduce(t); eta = float(t.item()) / world
            q.wormeta[:] = eta
        q.set_sector(wsec, worm["component"])
    sweep_key = "sweep_worm" if worm else "sweep"
    q.set_timing(True)      # HIP-event pairs around every launch, read out after the timed region (no host wait per launch)
    q.measure_begin()
    for _ in range(warmup):
        q.sweep(ncorr); q.measure_once()
    q.sync()
    tkeys = (sweep_key, "meas_misc", "meas_gleg", "meas_full", "meas_g4w", "meas_gw", "meas_wie", "meas_gtau_ac", "meas_giw", "meas_g2iw", "meas_g4iw")
    wkey, wres, wkern = {4: ("meas_g4w", "g4iw_worm", "k_meas_g4w"), 10: ("meas_wie", "qq_worm", "k_meas_wie"),
                         12: ("meas_wie", "nqqdag_worm", "k_meas_wie")}.get(wsec, ("meas_gw", "giw_worm", "k_meas_gw"))
    for k in tkeys:
        q.kernel_time(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        q.sweep(ncorr); q.measure_once()
    q.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the exchange step of the path (untimed here, it runs once per solve; its duration is reported): the moments of the
    # per-bin results are made on the GPU and the packed device buffer goes through ONE RCCL all-reduce; mean and
    # standard error over the accumulator bins of all ranks come out of it (w2dyn_reduce_pack, statistics.PackedMoments)
    q.measure_finish()
    t1 = time.perf_counter()
    if rehearse:           # gloo rehearsal on one GPU: host path
        sg, sh = DistributedSample(q.gleg_bins, comm), DistributedSample(q.histo_bins, comm)
    else:
        names = (["gleg_full", "gtau_full", "gleg", "histo"] if over.get("offdiag") else ["gleg", "gtau", "histo"]) + ([wres] if worm else []) + (["giw"] if over.get("MeasGiw") else []) + \
                (["g4iw", "g2iw"] if q.fourpnt & 4 else [])
        pack = q.reduce_pack(names, comm).allreduce()
        sg, sh = pack.sample("gleg_full" if over.get("offdiag") else "gleg"), pack.sample("histo")
    gl, hi, gl_err, nb = sg.mean(), sh.mean(), sg.stderr(), float(sg.ntotal)
    t_reduce = time.perf_counter() - t1
    oc = DistributedSample(q.occ_bins, comm).mean()

    kt = {k: q.kernel_time(k) for k in tkeys}
    sweep_ms, nsweep = kt[sweep_key]
    offd = bool(over.get("offdiag"))
    gleg_ms, ngleg = kt[wkey if worm else ("meas_full" if offd else "meas_gleg")]
    misc_ms, _ = kt["meas_misc"]
    out = None
    if rank == 0:
        W = walkers
        cnt = q.result("counters")
        names = ["tryadd", "accadd", "qnadd", "tryrem", "accrem", "qnrem", "tryglob", "accglob", "qnglob", "tryshift", "accshift", "sampling"]
        c = dict(zip(names, cnt))
        if worm:       # pair moves inside worm space and worm replacements touch M the same way (counter order: enum Counter)
            c["tryadd"] += cnt[25]; c["accadd"] += cnt[26]; c["tryrem"] += cnt[27]; c["accrem"] += cnt[28]
            c["tryadd"] += cnt[35]; c["accadd"] += cnt[36]
        ks = np.arange(hi.shape[-1])
        k1 = float((hi * ks).sum(-1).mean()); k2 = float((hi * ks ** 2).sum(-1).mean())
        nflav = 2 * prob.nd
        if offd:       # one matrix over all flavours: its order is the sum of the flavours' orders (the square of the mean stands in for the mean square)
            k1 = nflav * k1; k2 = k1 * k1
        nlaunch_steps = W * ncorr * (steps + warmup)     # counters cover warm-up rounds too
        scale = (W * ncorr) / nlaunch_steps               # per-launch share
        # algorithmic bytes of one sweep launch: det-ratio proposals that pass the quantum-number
        # check read M (8k^2), 2k F-table pairs (32k) and both tau arrays (16k); every accepted pair move rewrites M
        # (read 8k^2 + write 8k^2); a removal proposal reads one element.
        prop = (c["tryadd"] - c["qnadd"]) * (8 * k2 + 48 * k1)
        acc = (c["accadd"] + c["accrem"]) * 16 * k2
        rem = c["tryrem"] * 8
        bytes_sweep = (prop + acc + rem) * scale
        ach_gbs = bytes_sweep / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        flops_gleg = W * (1 if offd else nflav) * k2 * (5 * over["NLegMax"] + 6)
        if worm:       # 8 (2 N4iwf)^2 (2 N4iwb + 1) flop per worm-space measurement
            nworm_meas = float(q.result("cntmeas_worm")[0]) / max(steps + warmup, 1)
            flops_gleg = nworm_meas * (8 * (2 * over["N4iwf"]) ** 2 * (2 * over["N4iwb"] + 1) if wsec == 4 else
                                       (8 * 2 * over["N3iwf"] * (2 * over["N3iwb"] + 1) if wsec == 12 else 8 * 2 * over["Niw"]))
        ach_tf = flops_gleg / (gleg_ms * 1e-3) / 1e12 if gleg_ms > 0 else 0.0
        value = world * W * ncorr * steps / dt
        kname, mname = ("k_sweep_worm", wkern) if worm else (("k_sweep_od", "k_meas_full") if offd else ("k_sweep", "k_meas_gleg"))
        tr_sweep, tr_src = pmc_traffic(kname, W, wname)
        tr_gleg, _ = pmc_traffic(mname, W, wname)
        out = {
            "metric": "Metropolis steps/sec (node)", "value": value, "unit": "MC steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["text"], "chains_per_gpu": W,
                       "walkers_per_bin": wpb, "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "measurements_per_s": world * W * steps / dt,
            "mean_order_per_flavour": k1 / nflav if offd else k1, "accept_ins": c["accadd"] / max(c["tryadd"], 1), "occ_double": float(oc[0, 0, 0, 1]),
            "allreduce_ms": t_reduce * 1e3,
            # the warm-up rounds are measured into the same accumulators: the error below belongs to steps+warmup rounds
            "gleg_err": {"max_stderr_l_lt_20": float(gl_err[..., :20].max()), "bins": int(nb),
                         "wall_s": dt * (steps + warmup) / steps},
            "roofline": {"kernel": kname, "bound": "hbm", "note": "algorithmic bytes against HBM peak as the contract asks; the chain state is L2/LDS resident and the kernel is bound by vector-instruction issue and latency: see roofline_issue",
                         "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": tr_sweep, "traffic_source": tr_src, "avg_launch_ms": sweep_ms, "launches": nsweep,
                         "algorithmic_bytes_per_launch": bytes_sweep},
            "roofline_meas": {"kernel": mname, "bound": "mfma", "note": "fp64 vector FMA bound (fp64 vector peak == fp64 MFMA peak on MI355X)",
                              "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS,
                              "traffic": tr_gleg, "avg_launch_ms": gleg_ms, "launches": ngleg, "algorithmic_flops_per_launch": flops_gleg},
            "kernel_ms": {kname: sweep_ms, mname: gleg_ms, "k_meas_misc": misc_ms},
        }
        # the bound that matters for the sweep kernel ("fp64 vector issue rate and wavefront divergence"): vector
        # instructions issued per Monte-Carlo step against the wave-instructions the algorithm itself needs, i.e. the
        # FMA work of the proposals and accepted updates at 64 lanes: 2k^2/64 per det-ratio proposal, 4k^2/64 per accepted move
        prof, prof_src = issue_profile(wname)
        alg_valu = ((c["tryadd"] - c["qnadd"]) * (2 * k2 + 10 * k1) + (c["accadd"] + c["accrem"]) * 4 * k2) / 64.0 / max(nlaunch_steps, 1)
        out["roofline_issue"] = {"kernel": kname, "bound": "valu-issue", "algorithmic_valu_instr_per_step": alg_valu,
                                 "steps_per_s_per_simd": value / world / 1024.0,
                                 "source": prof_src}
        if prof:
            pm = prof["per_mc_step"]; wf = prof.get("wave_time_fraction", {})
            out["roofline_issue"].update({"issued_valu_instr_per_step": pm["SQ_INSTS_VALU"], "issued_salu_instr_per_step": pm["SQ_INSTS_SALU"],
                                          "frac": alg_valu / pm["SQ_INSTS_VALU"], "valu_busy_of_simd_time": wf.get("valu_busy_simd"),
                                          "waves_per_simd": prof.get("waves_per_simd"), "wave_waiting_fraction": wf.get("waiting")})
        if over.get("AccumGtau") == 2:
            # algorithmic bytes of k_meas_gtau_ac: every touched level of every (chain, flavour, tau bin) is one
            # read-modify-write of (spill, mean, sqdiff) = 48 B; launch number n touches 1 + (trailing one bits of n) levels
            ac_ms, nac = kt["meas_gtau_ac"]
            nlev = int(np.log2(over["Nmeas"]))
            touched = [min(1 + ((n ^ (n + 1)).bit_length() - 1), nlev) for n in range(warmup, warmup + steps)]
            ac_bytes = W * nflav * over["Ntau"] * 48.0 * float(np.mean(touched))
            ac_gbs = ac_bytes / (ac_ms * 1e-3) / 1e9 if ac_ms > 0 else 0.0
            tr_ac, _ = pmc_traffic("k_meas_gtau_ac", W, wname)
            out["roofline_accum"] = {"kernel": "k_meas_gtau_ac", "bound": "hbm", "achieved": ac_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ac_gbs / HBM_PEAK_GBS, "traffic": tr_ac, "avg_launch_ms": ac_ms, "launches": nac,
                                     "algorithmic_bytes_per_launch": ac_bytes, "levels": nlev}
            out["kernel_ms"]["k_meas_gtau_ac"] = ac_ms
        if over.get("MeasGiw"):
            # G(iw) direct = sum_f 8 k_f^2 2Niw flop per chain (a complex x real multiply-add per element
            # and frequency).  The kernel executes a quarter of that: conjugate symmetry halves the frequencies, and a
            # complex x real multiply-add is two FMAs = 4 flop
            g_ms, ng = kt["meas_giw"]
            fl = W * nflav * 8.0 * k2 * 2 * over["Niw"]
            tf = fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            tr, _ = pmc_traffic("k_meas_giw", W, wname)
            out["roofline_giw"] = {"kernel": "k_meas_giw", "bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": tf / FP64_PEAK_TFLOPS, "traffic": tr, "avg_launch_ms": g_ms, "launches": ng,
                                   "algorithmic_flops_per_launch": fl, "executed_flops_per_launch": fl / 4.0,
                                   "note": "algorithmic flops as SURVEY 8(d) counts them (8 per element and frequency, all 2 Niw frequencies); executed: 4 per element and positive frequency"}
            out["kernel_ms"]["k_meas_giw"] = g_ms
        if q.fourpnt & 4:
            nwf, nwb = over["N4iwf"], over["N4iwb"]
            n2 = 2 * (nwf + nwb)
            g2_ms, ng2 = kt["meas_g2iw"]; g4_ms, ng4 = kt["meas_g4iw"]
            fl2 = W * nflav * 8.0 * n2 * (k2 + n2 * k1)                      # sum_f 8 n2 k (k + n2)
            tf2 = fl2 / (g2_ms * 1e-3) / 1e12 if g2_ms > 0 else 0.0
            n4 = nflav ** 2 * (2 * nwf) ** 2 * (2 * nwb + 1)
            by4 = W * n4 * 32.0                                              # 32 B of accumulator traffic per element and measurement
            gb4 = by4 / (g4_ms * 1e-3) / 1e9 if g4_ms > 0 else 0.0
            tr2, _ = pmc_traffic("k_meas_g2iw", W, wname); tr4, _ = pmc_traffic("k_meas_g4iw", W, wname, sum_all=True)
            out["roofline_g2iw"] = {"kernel": "k_meas_g2iw", "bound": "mfma", "achieved": tf2, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf2 / FP64_PEAK_TFLOPS, "traffic": tr2, "avg_launch_ms": g2_ms, "launches": ng2, "algorithmic_flops_per_launch": fl2,
                                    # what bounds it at production boxes (n2 >> k): every chain's n2 x n2 table is written once
                                    "hbm_bytes_written_per_launch": float(W) * nflav * n2 * n2 * 16.0,
                                    "hbm_write_gbs": float(W) * nflav * n2 * n2 * 16.0 / (g2_ms * 1e-3) / 1e9 if g2_ms > 0 else 0.0}
            out["roofline_g4iw"] = {"kernel": "k_meas_g4iw (+ k_meas_g4iw_exch, k_meas_g2acc: one timed step)", "bound": "hbm", "achieved": gb4, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": gb4 / HBM_PEAK_GBS, "traffic": tr4, "avg_launch_ms": g4_ms, "launches": ng4,
                                    "algorithmic_bytes_per_launch": by4, "complex_macs_per_launch": float(W) * n4,
                                    "fp64_tflops": float(W) * n4 * 8.0 / (g4_ms * 1e-3) / 1e12 if g4_ms > 0 else 0.0,
                                    "note": "SURVEY 8(d) prices the assembly as one 32-B read-modify-write of every bin per measurement; the kernel keeps the bins of a "
                                            "workgroup in registers across the chains of a launch, so the bins cross HBM once per launch and frac can exceed 1"}
            out["kernel_ms"].update({"k_meas_g2iw": g2_ms, "k_meas_g4iw": g4_ms})
        if worm:
            cs = q.cntsampling
            out["worm"] = {"eta": eta, "eta_search_rounds": eta_tries, "steps_z": int(cs[0]), "steps_worm": int(cs[wsec - 1]), "sector": wsec, "component": worm["component"]}
        if world == 1 and with_cpu:
            # the GPU box gives one GPU's job 16 of the host's cores (worker pools beyond that are killed by its process
            # guard); os.cpu_count() reports the whole machine, stated next to the cores actually used
            cores = cpu_cores or max(1, min(16, os.cpu_count() or 1))
            try:
                out["cpu_baseline"] = cpu_baseline(cores, wname, eta=eta)
                out["cpu_baseline"]["host_cpus"] = os.cpu_count()
                out["cpu_baseline"]["cores_note"] = "16 = this GPU's share of the host (1/8 of an 8-GPU node's cores); the reference scales linearly in ranks"
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
                ce = out["cpu_baseline"].get("gleg_err")
                if ce:       # Monte-Carlo errors fall as 1/sqrt(time): CPU error scaled to the GPU's wall time
                    ge = out["gleg_err"]
                    out["gleg_err_at_fixed_time"] = {"wall_s": ge["wall_s"], "gpu": ge["max_stderr_l_lt_20"],
                                                     "cpu_same_wall_time": ce["max_stderr_l_lt_20"] * (ce["wall_s"] / ge["wall_s"]) ** 0.5}
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "MC steps/s", "cores": cores, "kind": "port", "sample": "failed: %r" % (e,)}
    q.dest_ctqmc(); q.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--walkers", type=int, default=int(os.environ.get("W2DYN_WALKERS", 0)), help="chains per GPU (default: 131072 for C2, the workload's own figure otherwise)")
    ap.add_argument("--walkers-per-bin", type=int, default=int(os.environ.get("W2DYN_WPB", 8)))
    ap.add_argument("--thermal", type=int, default=20000, help="untimed thermalisation MC steps per chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS), help="run this workload alone (default: the side workloads, then C2, the configuration the metric is quoted on)")
    ap.add_argument("--no-side", action="store_true", help="headline workload only")
    ap.add_argument("--eta", type=float, default=None, help="worm workloads: use this eta instead of searching for it (profiling passes)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from w2dynamics_amd.statistics import TorchComm

    env = Env()
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    env.rank = int(os.environ.get("RANK", "0"))
    env.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # W2DYN_REHEARSE=1 (not for measurements): all ranks on GPU 0 with the gloo backend, to exercise the N>1 code path on
    # a one-GPU box; the real thing is one rank per GPU over RCCL
    env.rehearse = os.environ.get("W2DYN_REHEARSE", "0") == "1"
    if env.rehearse:
        env.local_rank = 0
    if env.world > 1:
        torch.cuda.set_device(env.local_rank)
        if env.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", env.local_rank))
    env.comm = TorchComm()
    env.red_dev = "cpu" if env.rehearse else "cuda"

    head = a.config or HEADLINE
    side = {}
    if a.config is None and not a.no_side:
        for wname in SIDE_WORKLOADS:
            wl = WORKLOADS[wname]
            try:
                r = run_workload(env, wname, wl.get("chains", 16384), min(a.steps, 10), min(a.warmup, 2), min(a.thermal, 20000),
                                 a.walkers_per_bin, with_cpu=not a.no_cpu_baseline, cpu_cores=a.cpu_cores)
            except Exception as e:      # a side workload must not take the headline down with it
                r = {"error": repr(e)}
            if env.rank == 0:
                side[wname] = r
    walkers = a.walkers or (131072 if head in ("C2", "C2A") else WORKLOADS[head].get("chains", 16384))
    out = run_workload(env, head, walkers, a.steps, a.warmup, a.thermal, a.walkers_per_bin, with_cpu=not a.no_cpu_baseline,
                       cpu_cores=a.cpu_cores, fixed_eta=a.eta)
    if env.rank == 0:
        if side:
            keep = ("value", "unit", "measurements_per_s", "ms_per_step", "steps", "warmup", "config", "mean_order_per_flavour", "roofline", "roofline_meas",
                    "roofline_giw", "roofline_g2iw", "roofline_g4iw", "roofline_issue", "kernel_ms", "worm", "cpu_baseline", "gpu_over_cpu", "error")
            out["configs"] = {k: {kk: v[kk] for kk in keep if kk in v} for k, v in side.items()}
        print(json.dumps(out))
    if env.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
