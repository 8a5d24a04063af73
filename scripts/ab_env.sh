#!/bin/bash
# usage (on the GPU box): AB_CONFIGS="c2 c5" bash scripts/ab_env.sh "RT_SHADOW_PACKET=0" "RT_SHADOW_PACKET=1" ...
# one bench summary line per (environment setting, configuration); the library reads its developer knobs from the environment
# once, in rt_context_create
mkdir -p gpurun_out
for c in ${AB_CONFIGS:-c2}; do
  for e in "$@"; do
    echo "== $c $e"
    tag=$(echo "$c_$e" | tr -c 'A-Za-z0-9_=' '_')
    env $e timeout -k 10 ${AB_TIMEOUT:-200} python3 bench.py --config $c --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/ab_$tag.err | python3 -c "import sys,json; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],3), d['config']['kernel_ms_per_step'])" || { tail -5 gpurun_out/ab_$tag.err; exit 1; }
  done
done
