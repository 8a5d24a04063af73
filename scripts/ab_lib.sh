#!/bin/bash
# usage (on the GPU box): AB_CONFIGS="c2 c3" bash scripts/ab_lib.sh v0 v1 ...  -> one bench summary line per (configuration, variants/<v>.so)
mkdir -p gpurun_out
for c in ${AB_CONFIGS:-c2}; do
  for v in "$@"; do
    echo "== $c $v"
    RTCORE_LIB=$PWD/variants/$v.so timeout -k 10 ${AB_TIMEOUT:-200} python3 bench.py --config $c --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/ab_${c}_$v.err | python3 -c "import sys,json; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],3), d['config']['kernel_ms_per_step'])" || { tail -5 gpurun_out/ab_${c}_$v.err; exit 1; }
  done
done
