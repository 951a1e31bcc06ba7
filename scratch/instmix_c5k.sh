#!/bin/bash
# instruction mix of the sweep kernel: two PMC passes (SQ block only), no trace domains
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="$ROOT/bench.py --steps 3 --warmup 1 --thermal 2000 --walkers 4096 --config C5K --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/pmcA -o a -- python3 $B > $OUT/pmcA.log 2>&1 || { tail -5 $OUT/pmcA.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmcB -o b -- python3 $B > $OUT/pmcB.log 2>&1 || { tail -5 $OUT/pmcB.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("OUT", "/tmp")
for tag in ("pmcA", "pmcB"):
    files = glob.glob(os.path.join(os.getcwd() if False else "%s/gpurun_out/%s" % (os.environ["GRAFT_REPO_ROOT"], tag), "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if not k.startswith("k_sweep"): continue
        print(tag, k, {c: sum(v[1:]) / max(len(v) - 1, 1) for c, v in d.items()})
PY
