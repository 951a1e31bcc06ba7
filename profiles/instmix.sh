#!/bin/bash
# instruction mix of the sweep kernel: two PMC passes (SQ block only, 8 counters each), no trace domains
# usage (GPU box, repo root): bash profiles/instmix.sh <tag> [config] [chains]   -> gpurun_out/<tag>_sweep_instmix.json
set -u
TAG=${1:-r02}; CFG=${2:-C2}; CH=${3:-16384}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="$ROOT/bench.py --steps 3 --warmup 1 --thermal 4000 --walkers $CH --config $CFG --no-cpu-baseline"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_pmcA -o a -- python3 $B > $OUT/${TAG}_pmcA.log 2>&1 || { tail -5 $OUT/${TAG}_pmcA.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/${TAG}_pmcB -o b -- python3 $B > $OUT/${TAG}_pmcB.log 2>&1 || { tail -5 $OUT/${TAG}_pmcB.log; exit 1; }
python3 $ROOT/profiles/instmix_summary.py $OUT/${TAG}_pmcA $OUT/${TAG}_pmcB $CFG $CH "$B" > $OUT/${TAG}_sweep_instmix.json && cat $OUT/${TAG}_sweep_instmix.json
