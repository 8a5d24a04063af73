#!/bin/bash
# Memory-pipeline counters of the traversal kernels (TA / TCP / TD), separate passes.
export TMPDIR=/tmp
TAG=${1:-pmcmem}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }; }
run m1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
run m2 TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run m3 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
run m4 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum
cd $GRAFT_REPO_ROOT && python3 - $OUT <<'PY'
import csv, glob, os, sys, json, re
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f, newline='')):
        m = re.match(r'(?:void )?(k_\w+(?:<\w+>)?)', row['Kernel_Name'])
        if m: acc[m.group(1)][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, d in acc.items():
    if 'true' in k: continue
    res[k] = {c: {'avg': sum(v)/len(v), 'max': max(v), 'n': len(v)} for c, v in d.items()}
json.dump(res, open(os.path.join(out, 'mem_summary.json'), 'w'), indent=1, sort_keys=True)
for k in ('k_trace_closest<false>', 'k_shadow_hard<false>', 'k_shade<false>'):
    if k in res:
        print(k); [print('   ', c, round(v['avg']), round(v['max'])) for c, v in sorted(res[k].items())]
PY
