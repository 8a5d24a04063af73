#!/bin/bash
# Bench lines of the BASELINE configurations other than the default (c2).  Outputs under gpurun_out/<tag>/.
export TMPDIR=/tmp
TAG=${1:-cfg}
EXTRA=${BENCH_EXTRA---no-cpu-baseline}
shift
mkdir -p gpurun_out/$TAG
for c in "$@"; do
  timeout -k 10 500 python3 bench.py --config $c --steps 3 --warmup 1 $EXTRA > gpurun_out/$TAG/bench_$c.json 2> gpurun_out/$TAG/bench_$c.err || { tail -20 gpurun_out/$TAG/bench_$c.err; exit 1; }
  cat gpurun_out/$TAG/bench_$c.json
done
