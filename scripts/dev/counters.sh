#!/bin/bash
# developer: per-step work counters of the bench configurations
for c in ${AB_CONFIGS:-c3}; do
  timeout -k 10 300 python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/counters_$c.err | python3 -c "
import sys,json; d=json.load(sys.stdin); print('$c', round(d['ms_per_step'],2), d['config']['kernel_ms_per_step']); print(d['config']['counters_per_step'])" || { tail -5 gpurun_out/counters_$c.err; exit 1; }
done
