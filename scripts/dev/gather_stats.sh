#!/bin/bash
# developer: photon visits per lookup by kind of walk (variants gs1..gs5 built with -DRT_GATHER_STATS=1..5, see rt_photon.h)
for v in "$@"; do
  RTCORE_LIB=$PWD/variants/$v.so timeout -k 10 300 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/gstats_$v.err | python3 -c "
import sys,json; d=json.load(sys.stdin); c=d['config']['counters_per_step']; print('$v', d['ms_per_step'], {k:c[k] for k in ('photon_visits','gathers','paths')}, c['photon_visits']/max(c['gathers'],1))" || { tail -5 gpurun_out/gstats_$v.err; exit 1; }
done
