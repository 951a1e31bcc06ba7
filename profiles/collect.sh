#!/bin/bash
# Run on the GPU box from the repo root: bash profiles/collect.sh <tag> [chains] [config]
# 1. plain bench line  2. rocprofv3 kernel trace + stats of the same command  3./4. PMC passes (FETCH_SIZE, WRITE_SIZE
# in runs of their own, no trace domains).  Results land in gpurun_out/<tag>_*; copy what is to be judged to profiles/.
# Worm workloads: the PMC passes (and the kernel trace) get the eta the plain run found (--eta), so that no count_steps
# launches - sweeps of another length - sit among the launches that are averaged.
set -u
TAG=${1:-r03}
CH=${2:-16384}
CFG=${3:-C2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --steps 10 --warmup 2 --walkers $CH --config $CFG"
timeout -k 10 420 python3 $BENCH > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || { echo "bench failed"; tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
tail -c 3000 "$OUT/${TAG}_bench.json"
ETA=$(python3 -c "import json,sys; d=json.loads(open('$OUT/${TAG}_bench.json').read().strip().splitlines()[-1]); print('--eta %r' % d['worm']['eta'] if 'worm' in d else '')")
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof" -o stats -- python3 $BENCH --no-cpu-baseline $ETA > "$OUT/${TAG}_prof.log" 2>&1 || { echo "rocprof stats failed"; tail -5 "$OUT/${TAG}_prof.log"; exit 1; }
find "$OUT/${TAG}_prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
cat "$OUT/${TAG}_kernel_stats.csv"
PMCB="$ROOT/bench.py --steps 3 --warmup 1 --thermal 2000 --walkers $CH --config $CFG --no-cpu-baseline $ETA"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -o pmc -- python3 $PMCB > "$OUT/${TAG}_pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/${TAG}_pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -o pmc -- python3 $PMCB > "$OUT/${TAG}_pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/${TAG}_pmc_write.log"; exit 1; }
python3 "$ROOT/profiles/pmc_summary.py" "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write" $CH $CFG 4 > "$OUT/${TAG}_pmc.json" && cat "$OUT/${TAG}_pmc.json"
