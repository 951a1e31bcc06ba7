#!/bin/bash
# A/B probe: default library vs a variant (diagnostic)
set -u
V=${1:-}
run() { timeout -k 10 200 "$@" || exit 1; }
for lib in "" $V; do
  if [ -n "$lib" ]; then export W2DYN_HIP_LIB=$PWD/variants/$lib; echo "== $lib"; else echo "== default"; fi
  run python -u tests/perf_probe.py C2 16384
  run python -u tests/perf_probe.py C5Z 16384
  run python -u tests/perf_probe.py C4 8192
  run python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 worm', d['value'], d['kernel_ms'])"
done
