#!/bin/bash
# On the GPU box: C5 with the host-built (binned SAH) and the device-built (linear) BVH — load / commit seconds and step time.
for mode in "" "--device-bvh"; do
  echo "== C5 ${mode:-host-bvh}"
  timeout -k 10 400 python3 bench.py --config ${BVH_CONFIG:-c5} --steps ${BVH_STEPS:-2} --warmup 1 --no-cpu-baseline $mode 2>gpurun_out/bvh_cmp.err | python3 -c "import sys,json; d=json.load(sys.stdin); c=d['config']; print(round(d['ms_per_step'],2), 'ms/step  load', c['scene_load_s'], 's  commit', c['scene_commit_s'], 's ', c['mesh_bvh'], c['kernel_ms_per_step'])" || { tail -5 gpurun_out/bvh_cmp.err; exit 1; }
done
