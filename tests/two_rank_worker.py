"""Not a test: one rank of tests/test_gpu_two_ranks.py (started as a fresh process, before any GPU call of its own).

    python tests/two_rank_worker.py <rank> <world> <port> <out.npz> <chains per rank>

world = 2: two ranks share GPU 0 over a gloo group, rank r runs chains [r W, (r+1) W); world = 1: one context with all
2 W chains, no process group.  Every rank runs the same solver calls; rank 0 writes what it got.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    rank, world, port, out, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    import torch.distributed as dist
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from w2dynamics_amd import synth
    from w2dynamics_amd.solver import HipCtHybSolver
    from w2dynamics_amd.statistics import DistributedSample, TorchComm
    comm = TorchComm()
    nw = W if world > 1 else 2 * W
    base = dict(Nwarmups=2000, Nmeas=20, NCorr=20, TaudiffMax=6.0, Nftau=501, Ntau=60, NLegMax=20, Niw=8, MeasGiw=1,
                N4iwf=3, N4iwb=2, MeasG2iw=1, MeasG4iwPP=1)
    res = {}
    prob = synth.config("C3", nftau=501)
    # 1. Z-space solve with FourPnt = 4: one-particle results through DistributedSample, two-particle arrays through
    #    reduce_pack().allreduce() (device-made moments; staged through the host for the gloo group)
    s = HipCtHybSolver(dict(base), seed=5, comm=comm, nwalkers=nw, walkers_per_bin=8)
    s.set_problem(prob, compute_fourpoint=4)
    r = s.solve()
    s.postprocessing(r)
    for key in ("gleg", "gtau", "occ", "giw-meas", "giw", "siw", "g4iw", "g4iw-pp", "g2iw", "sign"):
        res["z_%s_mean" % key] = np.asarray(r[key].mean()); res["z_%s_err" % key] = np.asarray(r[key].stderr()); res["z_%s_n" % key] = r[key].ntotal
    # 2. one worm component at a fixed eta (the eta search of a W-chain and a 2W-chain context round differently)
    s.qmc.update(PercentageWormInsert=0.2, PercentageWormReplace=0.2, WormEta=0.3, WormMeasG4iw=1, WormSearchEta=0, Nwarmups2Plus=2000, Nmeas=40)
    s.set_problem(prob, compute_fourpoint=8)
    rw = s.solve_component(0, 4, 16, [])
    g = rw["g4iw-worm/00016"]
    res["w_mean"] = np.asarray(g.mean()); res["w_err"] = np.asarray(g.stderr()); res["w_n"] = g.ntotal
    res["w_occ_mean"] = np.asarray(rw["occ"].mean()); res["w_occ_err"] = np.asarray(rw["occ"].stderr())
    # 3. eta search and window calibration: every rank finds its own value, all use the mean over ranks
    #    (impurity.py:739-740, 1161-1163); the rank's own values are kept in last_local
    s.qmc.update(WormSearchEta=1, WormEta=1e-6, TaudiffMax=-1.0, Nmeas=5)
    s.set_problem(prob, compute_fourpoint=8)
    rs = s.solve_component(0, 4, 16, [])
    loc = dict(s.last_local)
    res["eta_used"] = rs["worm-eta/00016"]
    for k in ("eta", "taudiffmax_phase1"):
        res[k + "_ranks"] = np.asarray(DistributedSample(np.array([[loc[k]]]), comm).local if world == 1 else _gather(comm, loc[k], world))
    res["taudiffmax_phase1_used"] = loc["taudiffmax_phase1_mean"]
    s.ctqmc.close()
    if rank == 0:
        np.savez(out, **res)
    if world > 1:
        dist.destroy_process_group()


def _gather(comm, value, world):
    buf = np.zeros(world); buf[comm.Get_rank()] = value
    return comm.allreduce_packed([buf])[0]


if __name__ == "__main__":
    main()
