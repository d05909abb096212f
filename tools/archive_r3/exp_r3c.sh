#!/bin/bash
# round 3, GPU call 3: where does the bit-position mode lose time?  per-kernel rocprofv3 stats of one prove() run in either
# table mode + the LDS-prefetch accumulation over the large tables.  Output: gpurun_out/r3c/
set -u
O=gpurun_out/r3c
rm -rf $O; mkdir -p $O
free -g > $O/host_mem.txt; nproc >> $O/host_mem.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, extra args, env...
  local name=$1; shift
  local extra=$1; shift
  env "$@" timeout 200 $B $extra > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'][:8])
except Exception as e:
    print('$name FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run window20 "" PLONK_MSM_TABLE=window
run bitpos20 "" PLONK_MSM_TABLE=bitpos
run bitpos20_lds "" PLONK_MSM_TABLE=bitpos PLONK_MSM_ACC=lds
run window20_lds "" PLONK_MSM_TABLE=window PLONK_MSM_ACC=lds
run bitpos20_noorder "" PLONK_MSM_TABLE=bitpos PLONK_MSM_ORDER=0
run window20b "" PLONK_MSM_TABLE=window
run bitpos20b "" PLONK_MSM_TABLE=bitpos
cd /tmp && export TMPDIR=/tmp
for mode in window bitpos; do
  PLONK_MSM_TABLE=$mode timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$mode -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof_$mode.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode $f"; head -25 "$f" | cut -d, -f1-5
  find $GRAFT_REPO_ROOT/$O/prof_$mode -name "*.db" -delete; find $GRAFT_REPO_ROOT/$O/prof_$mode -name "*kernel_trace.csv" -delete
done
cat $GRAFT_REPO_ROOT/$O/host_mem.txt
