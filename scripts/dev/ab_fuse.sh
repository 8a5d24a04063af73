#!/bin/bash
# A/B of the fused first-phase + shade kernels: RT_FUSE=0 / 1 on the bench configurations.  bash scripts/dev/ab_fuse.sh <configs...>
mkdir -p gpurun_out/ab_fuse
for c in "$@"; do
  for f in 0 1; do
    RT_FUSE=$f timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/ab_fuse/${c}_$f.json 2> gpurun_out/ab_fuse/${c}_$f.err || { echo "$c fuse=$f FAILED"; tail -5 gpurun_out/ab_fuse/${c}_$f.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_fuse/${c}_$f.json"))
print("$c fuse=$f", round(d["ms_per_step"], 3), "ms/step", round(d["value"]), "Mrays/s", d["config"]["kernel_ms_per_step"])
PY
  done
done
