#!/usr/bin/env python3
"""Summarise the two SQ counter passes of profiles/instmix.sh into instructions per Monte-Carlo step of the sweep kernel.

usage: instmix_summary.py <dir pass A> <dir pass B> <config> <chains> <command>
The first sweep dispatch of a run is the thermalisation launch (different step count) and is dropped; the step count of a
launch is NCorr of the workload (bench.py WORKLOADS).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed
over waves; their ratios are fractions of wave lifetime.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
        acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        meta[k] = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size")}
    out = {}
    for k, d in acc.items():
        if not k.startswith("k_sweep"):
            continue
        out[k] = {}
        for c, v in d.items():
            v.sort(); vals = [x for _, x in v][1:] or [x for _, x in v]
            out[k][c] = sum(vals) / len(vals)
    return out, meta


def main():
    a, meta = load(sys.argv[1]); b, _ = load(sys.argv[2])
    cfg, chains, cmd = sys.argv[3], int(sys.argv[4]), sys.argv[5]
    import bench
    ncorr = bench.WORKLOADS[cfg]["overrides"]["NCorr"]
    name = max(a, key=lambda k: a[k].get("SQ_INSTS_VALU", 0.0))
    cnt = dict(a[name]); cnt.update(b.get(name, {}))
    steps = chains * ncorr
    # rocprofv3 of this ROCm release reports half the allocation for wave64 kernels on gfx950 (84 for the 168-register
    # variant of the sweep kernel, 128 for the 256-register one; the kernel descriptors of the build say 168 / 256)
    vg = 2 * (int(meta[name]["vgpr"] or 0) + int(meta[name]["accum_vgpr"] or 0))
    alloc = -(-vg // 8) * 8
    wps = min(8, 512 // alloc) if alloc else None
    res = {"config": cfg, "chains_per_gpu": chains, "kernel": name, "command": cmd + " (profiles/instmix.sh: two rocprofv3 --pmc passes, SQ block only)",
           "unit": "per launch (%d chains x %d steps), first launch dropped" % (chains, ncorr),
           "registers": meta[name], "vgprs_allocated": alloc, "waves_per_simd": wps, "counters": cnt,
           "per_mc_step": {k: cnt[k] / steps for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM") if k in cnt}}
    if "SQ_WAVE_CYCLES" in cnt and "SQ_WAIT_ANY" in cnt:
        wc = cnt["SQ_WAVE_CYCLES"]
        valu = cnt.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        res["wave_time_fraction"] = {"issuing": cnt.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "waiting": cnt["SQ_WAIT_ANY"] / wc, "valu": valu,
                                     "valu_busy_simd": valu * wps if wps else None}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
