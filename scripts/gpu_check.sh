#!/bin/bash
# One GPU-box call: parity tests, bench line, kernel-trace stats.  Outputs under gpurun_out/.
export TMPDIR=/tmp
TAG=${1:-run}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$TAG/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/$TAG/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest_gpu.log
timeout -k 10 400 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 1; }
cat gpurun_out/$TAG/bench.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/$TAG/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG/bench_prof.err || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/$TAG/bench_prof.err; exit 1; }
find $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -3
