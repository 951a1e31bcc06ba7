#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into per-launch HBM traffic.

usage: pmc_summary.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <chains> [config] [launches]

`launches` = steps + warmup of the profiled bench command: only the LAST that many dispatches of each kernel are averaged,
i.e. the launches of the measurement loop.  Everything before them (thermalisation; for worm workloads the eta search, whose
count_steps launches run a different number of steps - the bench is given --eta for these passes so that none happen) is
left out, so that the figure is comparable with the algorithmic bytes of ONE launch of the timed region.

Units and gfx950 correction as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B read requests at 64 B, so the read side is doubled.  The guide calibrates this for 16-B/lane
streams only; k_sweep loads 8 B/lane, so the corrected value is an upper estimate and the raw one is kept next to it.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter, last=0):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    rows = {}
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip().split("::")[-1]
            rows.setdefault(name, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for name, v in rows.items():
        v.sort()
        vals = [x for _, x in v]
        if last > 0:
            vals = vals[-last:]
        elif name.startswith("k_sweep") and len(vals) > 1:
            vals = vals[1:]
        out[name] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    last = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE", last)
    write = per_kernel(sys.argv[2], "WRITE_SIZE", last)
    chains = int(sys.argv[3]) if len(sys.argv) > 3 else None
    res = {"chains_per_gpu": chains, "config": sys.argv[4] if len(sys.argv) > 4 else "C2", "unit": "bytes per launch",
           "launches_averaged": "the last %d of each kernel (measurement loop)" % last if last else "all but the first sweep launch", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
        res["kernels"][k] = {"fetch_raw": f * 1024, "fetch_corrected": 2 * f * 1024, "write": w * 1024,
                             "traffic": (2 * f + w) * 1024, "launches_fetch_pass": nf, "launches_write_pass": nw}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
