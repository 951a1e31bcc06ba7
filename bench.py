#!/usr/bin/env python3
"""Benchmark of the CT-HYB hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): 1-band Hubbard model, Bethe lattice D=1, U=2, beta=100, half filling,
TaudiffMax=10 (9 sliding windows), reference defaults otherwise (Ntau=1000, NLegMax=100, Nftau=5001,
PercentageGlobalMove = PercentageTauShiftMove = 0.005) and NCorr=200 as in BASELINE.md.
One bench "step" = one pass of the hot path over the batch of chains of every GPU: NCorr Monte-Carlo steps for each
chain (sweep kernel) followed by one pass of the Z-space estimators (sign/histograms, Legendre + tau bins).
`value` = Metropolis steps per second of the whole job = n_gpus * chains * NCorr * K / time; measurements/s is
reported next to it.  Chains are thermalised before the timed region; all state is resident in HBM.
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

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (default)
    "C2": dict(config="C2", nftau=5001, overrides=dict(Nftau=5001, NCorr=200, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
               text="C2: 1-band Hubbard, Bethe D=1, U=2, beta=100, half filling, TaudiffMax=10, NCorr=200, NLegMax=100, Ntau=1000, Nftau=5001, default move mix"),
    # C2 with the autocorrelation accumulator of G(tau) (AccumGtau = 2, one stack of log2(Nmeas) levels per chain): adds the
    # HBM-streaming kernel k_meas_gtau_ac to every step; extra bench line, not the headline
    "C2A": dict(config="C2", nftau=5001, overrides=dict(Nftau=5001, NCorr=200, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1024, AccumGtau=2),
                text="C2 + AccumGtau=2 (per-chain autocorrelation accumulator of G(tau), 10 levels)"),
    # the other single-GPU configurations (parity-test cases; selectable with --config for extra bench lines)
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (beta = 10: expansion order ~1.4 per flavour)
    "C1": dict(config="C1", nftau=5001, overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
               text="C1: 1-band Hubbard, Bethe D=1, U=2, beta=10, half filling, one window, NCorr=100, NLegMax=100, Ntau=1000"),
    "C3": dict(config="C3", nftau=5001, overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
               text="C3: 2-orbital density-density, U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    "C4": dict(config="C4", nftau=5001, overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
               text="C4: 3-orbital Kanamori (QNs Nt Szt Qzt, matrix local trace), U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    "C5Z": dict(config="C5Z", nftau=5001, overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
                text="C5 (Z-space part): 5-orbital density-density, U=2, J=0.25, beta=100, TaudiffMax=10, NCorr=100"),
    "C5K": dict(config="C5KB", nftau=5001, overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1),
                text="C5K: 5-orbital Kanamori (d shell, QNs Nt Szt Qzt: 352 blocks of up to 10 states, matrix local trace), U=2, J=0.25, beta=50, TaudiffMax=10, NCorr=100"),
    # worm sampling on the matrix trace path: 3-orbital Kanamori, component 8 = flavours 0011
    "C4W": dict(config="C4", nftau=5001, worm=dict(component=8, nfix=2000),
                overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1, N4iwf=10, N4iwb=5,
                               PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, G4ph=1, G4pp=0),
                text="C4W: 3-orbital Kanamori (matrix local trace), beta=50, worm sampling in sector 4 (G4iw, component 8 = flavours 0011, "
                     "N4iwf=10, N4iwb=5, PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
    # BASELINE.json configs[4] on one GPU: worm sampling of one component of the two-particle Green's function (sector 4)
    "C5": dict(config="C5Z", nftau=5001, worm=dict(component=12, nfix=2000),
               overrides=dict(Nftau=5001, NCorr=100, TaudiffMax=10.0, Ntau=1000, NLegMax=100, Nwarmups=0, Nmeas=1, N4iwf=10, N4iwb=5,
                              PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=1.0, WormMeasG4iw=1, G4ph=1, G4pp=0),
               text="C5: 5-orbital density-density, U=2, J=0.25, beta=100, TaudiffMax=10, NCorr=100, worm sampling in sector 4 (G4iw, "
                    "component 12 = flavours 0011, N4iwf=10, N4iwb=5, PercentageWormInsert=PercentageWormReplace=0.2, eta balanced by count_steps)"),
}
WORKLOAD = WORKLOADS["C2"]


def _cpu_chain(args):
    """One reference (or oracle) chain on one core; returns (mc_steps, seconds, kind)."""
    seed, nmeas, nwarm, wname, eta = args
    global WORKLOAD
    WORKLOAD = dict(WORKLOADS[wname])
    if WORKLOAD.get("worm"):
        WORKLOAD["worm"] = dict(WORKLOAD["worm"], eta=eta)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from w2dynamics_amd import params, synth
    p = synth.config(WORKLOAD["config"], nftau=WORKLOAD["nftau"])
    over = dict(WORKLOAD["overrides"]); over.update(Nmeas=nmeas, Nwarmups=nwarm)
    ps = params.make_pstring(over, beta=p.beta, Nd=p.nd, Hamiltonian=p.hamiltonian, QuantumNumbers=p.quantum_numbers)
    try:
        import refdrive
        if not refdrive.available():
            raise ImportError("oracle/_ref not built")
        worm = WORKLOAD.get("worm")
        s = refdrive.RefSolver(p, ps, seed, fourpnt=8 if worm else 0)
        s.set_taudiffmax(over["TaudiffMax"])
        s.warmup(nwarm)
        if worm:
            s.set_wormeta(worm["eta"])
            s.worm_warmup(0, 4, worm["component"])
        t0 = time.perf_counter(); s.measure(*((4, worm["component"]) if worm else (1, 1))); dt = time.perf_counter() - t0
        gl = None if worm else np.array(s.results(("gleg",))["gleg"], dtype=np.float64)
        s.close()
        return nmeas * over["NCorr"], dt, "reference", gl
    except Exception:
        import pyoracle
        worm = WORKLOAD.get("worm")
        w = pyoracle.OracleWalker(p, params.ParameterSet(ps), ("std", seed), fourpnt=8 if worm else 0)
        w.set_taudiffmax(over["TaudiffMax"])
        w.warmup(nwarm)
        if worm:
            w.set_wormeta(worm["eta"])
        t0 = time.perf_counter()
        if worm:
            w.measure_sector(4, worm["component"])
        else:
            w.measure()
        dt = time.perf_counter() - t0
        gl = None if worm else np.array(w.results()["gleg"], dtype=np.float64)
        w.close()
        return nmeas * over["NCorr"], dt, "port", gl


def cpu_baseline(cores, wname, nmeas=6000, nwarm=50000, eta=1.0):
    import multiprocessing as mp
    if wname != "C2":
        nmeas = 3000
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
    return {"value": rate, "unit": "MC steps/s", "cores": cores, "kind": kind, "gleg_err": gerr,
            "meas_per_s": sum((s / WORKLOADS[wname]["overrides"]["NCorr"]) / t for s, t, _, _ in res),
            "sample": "%d independent chains (one per core, as the reference's MPI ranks), each %d warm-up steps untimed + "
                      "%d measurements x NCorr=%d steps timed, same %s workload" % (cores, nwarm, nmeas, WORKLOADS[wname]["overrides"]["NCorr"], wname)}


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


def pmc_traffic(kernel, chains, config):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*_pmc.json, made by
    profiles/collect.sh + pmc_summary.py on the same command); None when no pass exists for this chain count."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("chains_per_gpu") != chains or d.get("config", "C2") != config:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel):
                return v["traffic"], os.path.basename(f)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--walkers", type=int, default=int(os.environ.get("W2DYN_WALKERS", 131072)), help="chains per GPU")
    ap.add_argument("--walkers-per-bin", type=int, default=int(os.environ.get("W2DYN_WPB", 8)))
    ap.add_argument("--thermal", type=int, default=20000, help="untimed thermalisation MC steps per chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOADS), help="workload (default: the configuration the metric is quoted on)")
    a = ap.parse_args()
    global WORKLOAD
    WORKLOAD = WORKLOADS[a.config]

    import torch
    import torch.distributed as dist
    from w2dynamics_amd import params, synth
    from w2dynamics_amd.mctqmc import MCTQMC
    from w2dynamics_amd.statistics import TorchComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # W2DYN_REHEARSE=1 (not for measurements): all ranks on GPU 0 with the gloo backend, to exercise the N>1 code path on
    # a one-GPU box; the real thing is one rank per GPU over RCCL
    rehearse = os.environ.get("W2DYN_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    comm = TorchComm()
    red_dev = "cpu" if rehearse else "cuda"

    over = WORKLOAD["overrides"]
    prob = synth.config(WORKLOAD["config"], nftau=WORKLOAD["nftau"])
    ps = params.make_pstring(over, beta=prob.beta, Nd=prob.nd, Hamiltonian=prob.hamiltonian, QuantumNumbers=prob.quantum_numbers)
    q = MCTQMC(device=local_rank, nwalkers=a.walkers, first_walker=rank * a.walkers, walkers_per_bin=a.walkers_per_bin)
    worm = WORKLOAD.get("worm")
    if worm:
        q.fourpnt = 8
    q.init_rnd(43890); q.init_paras(ps)
    q.init_solver(prob.umatrix, prob.ftau, prob.muimp, prob.screening, 1)
    q.set_taudiffmax(over["TaudiffMax"])
    ncorr = over["NCorr"]
    q.ctqmc_warmup(ncorr)                          # thermalise (untimed), in launches of the same size as the timed ones
    for _ in range(max(a.thermal // ncorr - 1, 0)):
        q.sweep(ncorr)
    eta = 1.0
    if worm:
        # eta search of solve_component (impurity.py:717-742): rescale until Z and worm space get equal numbers of steps
        # starting point: insertion prefactor (beta Nd)^4 eta = 1 (CTQMC.F90:5680-5684)
        q.wormeta[:] = 1.0 / (prob.beta * prob.nd) ** 4
        for _ in range(12):
            sw, sz = q.count_steps(4, worm["component"], worm["nfix"])
            if sw == 0 or sz == 0 or abs(sw - sz) / float(sw + sz) <= 0.1:
                break
            q.wormeta[:] = q.wormeta[:] * float(sz) / float(sw)
        eta = float(q.wormeta[2])
        if world > 1:      # all ranks use the mean eta (impurity.py:739-740, 784)
            t = torch.tensor([eta], dtype=torch.float64, device=red_dev); dist.all_reduce(t); eta = float(t.item()) / world
            q.wormeta[:] = eta
        q.set_sector(4, worm["component"])
    sweep_key = "sweep_worm" if worm else "sweep"
    q.set_timing(True)      # HIP-event pairs around every launch, read out after the timed region (no host wait per launch)
    q.measure_begin()
    for _ in range(a.warmup):
        q.sweep(ncorr); q.measure_once()
    q.sync()
    for k in (sweep_key, "meas_misc", "meas_gleg", "meas_g4w", "meas_gtau_ac"):
        q.kernel_time(k, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
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
    from w2dynamics_amd.statistics import DistributedSample
    t1 = time.perf_counter()
    if rehearse:           # gloo rehearsal on one GPU: host path
        sg, sh = DistributedSample(q.gleg_bins, comm), DistributedSample(q.histo_bins, comm)
    else:
        pack = q.reduce_pack(["gleg", "gtau", "histo"] + (["g4iw_worm"] if worm else []), comm).allreduce()
        sg, sh = pack.sample("gleg"), pack.sample("histo")
    gl, hi, gl_err, nb = sg.mean(), sh.mean(), sg.stderr(), float(sg.ntotal)
    t_reduce = time.perf_counter() - t1
    oc = DistributedSample(q.occ_bins, comm).mean()

    sweep_ms, nsweep = q.kernel_time(sweep_key)
    gleg_ms, ngleg = q.kernel_time("meas_g4w" if worm else "meas_gleg")
    misc_ms, _ = q.kernel_time("meas_misc")
    if rank == 0:
        W = a.walkers
        cnt = q.result("counters")
        names = ["tryadd", "accadd", "qnadd", "tryrem", "accrem", "qnrem", "tryglob", "accglob", "qnglob", "tryshift", "accshift", "sampling"]
        c = dict(zip(names, cnt))
        if worm:       # pair moves inside worm space and worm replacements touch M the same way (counter order: enum Counter)
            c["tryadd"] += cnt[25]; c["accadd"] += cnt[26]; c["tryrem"] += cnt[27]; c["accrem"] += cnt[28]
            c["tryadd"] += cnt[35]; c["accadd"] += cnt[36]
        ks = np.arange(hi.shape[-1])
        k1 = float((hi * ks).sum(-1).mean()); k2 = float((hi * ks ** 2).sum(-1).mean())
        nlaunch_steps = W * ncorr * (a.steps + a.warmup)     # counters cover warm-up rounds too
        scale = (W * ncorr) / nlaunch_steps                   # per-launch share
        # algorithmic bytes of one sweep launch: det-ratio proposals that pass the quantum-number
        # check read M (8k^2), 2k F-table pairs (32k) and both tau arrays (16k); every accepted pair move rewrites M
        # (read 8k^2 + write 8k^2); a removal proposal reads one element.
        prop = (c["tryadd"] - c["qnadd"]) * (8 * k2 + 48 * k1)
        acc = (c["accadd"] + c["accrem"]) * 16 * k2
        rem = c["tryrem"] * 8
        bytes_sweep = (prop + acc + rem) * scale
        ach_gbs = bytes_sweep / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        flops_gleg = W * prob.nd * 2 * k2 * (5 * over["NLegMax"] + 6)
        if worm:       # 8 (2 N4iwf)^2 (2 N4iwb + 1) flop per worm-space measurement
            nworm_meas = float(q.result("cntmeas_worm")[0]) / max(a.steps + a.warmup, 1)
            flops_gleg = nworm_meas * 8 * (2 * over["N4iwf"]) ** 2 * (2 * over["N4iwb"] + 1)
        ach_tf = flops_gleg / (gleg_ms * 1e-3) / 1e12 if gleg_ms > 0 else 0.0
        value = world * W * ncorr * a.steps / dt
        kname, mname = ("k_sweep_worm", "k_meas_g4w") if worm else ("k_sweep", "k_meas_gleg")
        tr_sweep, tr_src = pmc_traffic(kname, W, a.config)
        tr_gleg, _ = pmc_traffic(mname, W, a.config)
        out = {
            "metric": "Metropolis steps/sec (node)", "value": value, "unit": "MC steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOAD["text"], "chains_per_gpu": W,
                       "walkers_per_bin": a.walkers_per_bin, "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "measurements_per_s": world * W * a.steps / dt,
            "mean_order_per_flavour": k1, "accept_ins": c["accadd"] / max(c["tryadd"], 1), "occ_double": float(oc[0, 0, 0, 1]),
            "allreduce_ms": t_reduce * 1e3,
            # the warm-up rounds are measured into the same accumulators: the error below belongs to steps+warmup rounds
            "gleg_err": {"max_stderr_l_lt_20": float(gl_err[..., :20].max()), "bins": int(nb),
                         "wall_s": dt * (a.steps + a.warmup) / a.steps},
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
        prof, prof_src = issue_profile(a.config)
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
            ac_ms, nac = q.kernel_time("meas_gtau_ac")
            nlev = int(np.log2(over["Nmeas"]))
            touched = [min(1 + ((n ^ (n + 1)).bit_length() - 1), nlev) for n in range(a.warmup, a.warmup + a.steps)]
            ac_bytes = W * prob.nd * 2 * over["Ntau"] * 48.0 * float(np.mean(touched))
            ac_gbs = ac_bytes / (ac_ms * 1e-3) / 1e9 if ac_ms > 0 else 0.0
            tr_ac, _ = pmc_traffic("k_meas_gtau_ac", W, a.config)
            out["roofline_accum"] = {"kernel": "k_meas_gtau_ac", "bound": "hbm", "achieved": ac_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ac_gbs / HBM_PEAK_GBS, "traffic": tr_ac, "avg_launch_ms": ac_ms, "launches": nac,
                                     "algorithmic_bytes_per_launch": ac_bytes, "levels": nlev}
            out["kernel_ms"]["k_meas_gtau_ac"] = ac_ms
        if worm:
            cs = q.cntsampling
            out["worm"] = {"eta": eta, "steps_z": int(cs[0]), "steps_worm": int(cs[3]), "component": worm["component"]}
        if world == 1 and not a.no_cpu_baseline:
            # the GPU box gives one GPU's job 16 of the host's cores (worker pools beyond that are killed by its process
            # guard); os.cpu_count() reports the whole machine, stated next to the cores actually used
            cores = a.cpu_cores or max(1, min(16, os.cpu_count() or 1))
            try:
                out["cpu_baseline"] = cpu_baseline(cores, a.config, eta=eta)
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
        print(json.dumps(out))
    q.dest_ctqmc(); q.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
