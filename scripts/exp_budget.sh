#!/bin/bash
for gb in 6 12 24 48 96; do
  echo "== RT_FRAME_GB=$gb"
  RT_FRAME_GB=$gb timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'], d['roofline']['launches_per_step'])" || exit 1
done
