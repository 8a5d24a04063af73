#!/bin/bash
# developer: every kernel launch of the last step of a configuration, in order, with its duration (kernel trace)
CFG=${1:-c3}
bash scripts/prof_config.sh steptrace $CFG 2 > /dev/null || exit 1
f=$(find gpurun_out/steptrace/prof_$CFG -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/steptrace/launches_$CFG.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last step: from the last k_raygen* launch on
last=max(i for i,r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
t0=int(rows[last]["Start_Timestamp"])
prev_end=t0
for r in rows[last:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.3f gap %7.3f dur %8.3f  grid %9s %s"%((s-t0)/1e6,(s-prev_end)/1e6,(e-s)/1e6,r.get("Grid_Size_X", r.get("Grid_Size","?")),r["Kernel_Name"][:70]))
    prev_end=e
PY
wc -l gpurun_out/steptrace/launches_$CFG.txt
