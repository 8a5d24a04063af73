#!/bin/bash
# developer: sm_stats.sh for several variants/<v>.so
for v in "$@"; do
  RTCORE_LIB=$PWD/variants/$v.so timeout -k 10 300 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/smstats_$v.err | python3 -c "
import sys,json; d=json.load(sys.stdin); c=d['config']['counters_per_step']; print('$v', round(d['ms_per_step'],1), 'occluded steps/ray', round(c['gathers']/max(c['shades'],1),1), 'lit', round(c['photon_visits']/max(c['tex_evals'],1),1), d['config']['kernel_ms_per_step']['shadow_mesh'])" || { tail -5 gpurun_out/smstats_$v.err; exit 1; }
done
