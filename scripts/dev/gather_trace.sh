#!/bin/bash
# developer: per-launch durations of the photon gather stages of the last C4 step (kernel trace)
bash scripts/prof_config.sh gtrace c4 2 > /dev/null || exit 1
f=$(find gpurun_out/gtrace/prof_c4 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
out=[]
for r in rows:
    n=r["Kernel_Name"]
    if "gather" in n or "raygen" in n:
        out.append("%10.3f %8.3f %s"%((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,n[:40]))
print("\n".join(out[-9:]))
PY
