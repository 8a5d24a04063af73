#!/bin/bash
# rocprofv3 kernel-trace stats of one bench configuration.  bash scripts/prof_config.sh <tag> <config> [steps]
export TMPDIR=/tmp
TAG=${1:-prof}
CFG=${2:-c2}
STEPS=${3:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$CFG -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_prof_$CFG.json 2> $OUT/bench_prof_$CFG.err || { tail -20 $OUT/bench_prof_$CFG.err; exit 1; }
f=$(find $OUT/prof_$CFG -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats_$CFG.csv
head -14 $OUT/kernel_stats_$CFG.csv | cut -c1-200
