#!/bin/bash
# One GPU-box call: PMC passes of one bench configuration, folded into profiles/<round>_pmc_summary_<config>.json (keyed by the
# kernel-source hash), then the bench line that reads it.   bash scripts/pmc_all.sh <round-tag> <config>
export TMPDIR=/tmp
R=${1:-r02}
export BENCH_CONFIG=${2:-c2}
bash scripts/pmc_traffic.sh ${R}_pmc_${BENCH_CONFIG} > /dev/null || exit 1
cp gpurun_out/${R}_pmc_${BENCH_CONFIG}/summary.json gpurun_out/${R}_pmc_summary_${BENCH_CONFIG}.json
mkdir -p profiles && cp gpurun_out/${R}_pmc_summary_${BENCH_CONFIG}.json profiles/${R}_pmc_summary_${BENCH_CONFIG}.json
timeout -k 10 400 python3 bench.py --config ${BENCH_CONFIG} --no-other-configs ${BENCH_ARGS:-} > gpurun_out/${R}_bench_${BENCH_CONFIG}.json 2> gpurun_out/${R}_bench_${BENCH_CONFIG}.err || { tail -20 gpurun_out/${R}_bench_${BENCH_CONFIG}.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/${R}_bench_${BENCH_CONFIG}.json"))
r = d["roofline"]
print(d["ms_per_step"], d["value"], r["bound"], r["kernel"], r["frac"], r["traffic"], r["pmc_summary"])
for k, v in r["per_kernel"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "hbm_algorithmic_frac", "hbm_traffic_frac", "valu_algorithmic_frac", "valu_issue_util", "active_lane_ratio", "occupancy_waves_per_simd", "l2_hit_rate")})
print(d.get("cpu_baseline"))
PY
