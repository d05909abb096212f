#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench.
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/bench_pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_SQ -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/bench_pmc_SQ.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/bench_trace.log
