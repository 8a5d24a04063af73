#!/bin/bash
# developer: per-outcome tallies of the second shadow phase (variant built with -DRT_SM_STATS: photon_visits / tex_evals = node steps /
# rays of the rays that end lit, gathers / shades = of the rays that end occluded; `shades` then also holds the real shade count)
for c in ${AB_CONFIGS:-c5}; do
  RTCORE_LIB=$PWD/variants/smstats.so timeout -k 10 300 python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/smstats_$c.err | python3 -c "
import sys,json; d=json.load(sys.stdin); c=d['config']['counters_per_step']; print('$c', d['ms_per_step'], {k:c[k] for k in ('rays_shadow','node_visits','closest_node_visits','photon_visits','tex_evals','gathers','shades','paths')})" || { tail -5 gpurun_out/smstats_$c.err; exit 1; }
done
