#!/bin/bash
# usage (on the GPU box): bash scripts/ab.sh v0 v1 ...   -> one bench line summary per variant
for v in "$@"; do
  echo "== $v"
  RTCORE_LIB=$PWD/variants/$v.so timeout -k 10 200 python3 bench.py --config ${AB_CONFIG:-c2} --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline 2>gpurun_out/ab_$v.err | python3 -c "import sys,json; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['kernel_ms_per_step'])" || { tail -5 gpurun_out/ab_$v.err; exit 1; }
done
