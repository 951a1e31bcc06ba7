"""GPU (-m gpu): the N > 1 path with two HIP contexts as two ranks (one-GPU box: both ranks on device 0, gloo group).

Three fresh child processes are started before this process touches the GPU for them: rank 0 and rank 1 of a two-rank
job (W chains each, first_walker = rank W) and one single context with all 2 W chains.  Chain w's trajectory depends on
(seed, w) only, so the pooled two-rank results - through `DistributedSample` (one-particle results) and through
`reduce_pack().allreduce()` (two-particle arrays and the worm estimator: moments made on the GPU) - must equal those of
the single context.  Then `bench.py --gpus 2` in rehearsal mode as the driver launches it.  No scaling curve comes out
of this (two ranks share one GPU): it checks that the sharded path computes the right thing.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(args, timeout, env=None):
    return subprocess.Popen([sys.executable] + args, cwd=ROOT, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_two_ranks_equal_one_context_of_twice_the_chains(tmp_path):
    W = 64
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "two_rank_worker.py")
    outs = [str(tmp_path / "two.npz"), str(tmp_path / "r1.npz"), str(tmp_path / "one.npz")]
    procs = [_run([worker, "0", "2", str(port), outs[0], str(W)], 300), _run([worker, "1", "2", str(port), outs[1], str(W)], 300),
             _run([worker, "0", "1", "0", outs[2], str(W)], 300)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    two, one = np.load(outs[0]), np.load(outs[2])
    for key in ("gleg", "gtau", "occ", "giw-meas", "giw", "siw", "g4iw", "g4iw-pp", "g2iw", "sign"):
        m2, m1 = two["z_%s_mean" % key], one["z_%s_mean" % key]
        e2, e1 = two["z_%s_err" % key], one["z_%s_err" % key]
        assert int(two["z_%s_n" % key]) == (16 if key != "sign" else 2), key           # bins of both ranks; one sign per rank
        scale = max(np.abs(m1).max(), 1e-300)
        assert np.abs(m2 - m1).max() <= 1e-12 * scale, key
        if key == "sign":          # one observation per rank vs one per context: only the mean compares
            continue
        assert int(two["z_%s_n" % key]) == int(one["z_%s_n" % key]), key
        # device-made moments form the variance as a difference of sums: 1e-10 of the scale; host samples: 1e-12
        tol = 1e-10 if key in ("g4iw", "g4iw-pp", "g2iw") else 1e-12
        assert np.abs(e2 - e1).max() <= tol * max(np.abs(e1).max(), 1e-300), key
    assert int(two["w_n"]) == int(one["w_n"]) > 2
    sc = np.abs(one["w_mean"]).max()
    assert sc > 0 and np.abs(two["w_mean"] - one["w_mean"]).max() <= 1e-12 * sc
    assert np.abs(two["w_err"] - one["w_err"]).max() <= 1e-10 * np.abs(one["w_err"]).max()
    assert np.abs(two["w_occ_mean"] - one["w_occ_mean"]).max() <= 1e-12 and np.abs(two["w_occ_err"] - one["w_occ_err"]).max() <= 1e-12
    # eta and the provisional window of the calibration are the means over the ranks' own values
    assert two["eta_ranks"].shape == (2,) and two["eta_ranks"][0] != two["eta_ranks"][1]
    assert float(two["eta_used"]) == pytest.approx(two["eta_ranks"].mean(), rel=1e-14)
    assert float(two["taudiffmax_phase1_used"]) == pytest.approx(two["taudiffmax_phase1_ranks"].mean(), rel=1e-14)
    assert two["taudiffmax_phase1_ranks"][0] != two["taudiffmax_phase1_ranks"][1]


def test_bench_two_ranks_rehearsal():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), both ranks on GPU 0
    with the gloo backend (W2DYN_REHEARSE=1): one JSON line from rank 0 with n_gpus = 2 and the work of both ranks."""
    W, steps = 2048, 3
    cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--config", "C1", "--walkers", str(W),
           "--thermal", "2000", "--no-cpu-baseline"]
    p = _run(cmd, 400, env={"W2DYN_REHEARSE": "1"})
    try:
        o, _ = p.communicate(timeout=400)
    except subprocess.TimeoutExpired:
        p.kill(); o, _ = p.communicate()
    assert p.returncode == 0, o[-3000:]
    lines = [l for l in o.splitlines() if l.startswith("{")]
    assert len(lines) == 1, o[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak" and d["config"]["chains_per_gpu"] == W
    work = 2 * W * 100 * steps                       # Metropolis steps of both ranks in the timed region (NCorr = 100)
    assert d["value"] * d["ms_per_step"] * 1e-3 * steps == pytest.approx(work, rel=1e-9)
    assert d["gleg_err"]["bins"] == 2 * W // 8       # the reduction saw the accumulator bins of both ranks
