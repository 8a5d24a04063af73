#!/bin/bash
# HBM traffic of the hot kernels from PMC counters, one counter set per pass (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").
# Run on the GPU box from the repo root:  bash scripts/pmc_traffic.sh <tag>
export TMPDIR=/tmp
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/sq1.json 2> $OUT/sq1.err || { tail -5 $OUT/sq1.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/sq2.json 2> $OUT/sq2.err || { tail -5 $OUT/sq2.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/tcc -- python3 $GRAFT_REPO_ROOT/bench.py --config ${BENCH_CONFIG:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/tcc.json 2> $OUT/tcc.err || { tail -5 $OUT/tcc.err; exit 1; }
cd $GRAFT_REPO_ROOT && python3 scripts/pmc_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
