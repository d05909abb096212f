#!/bin/bash
# round 3, GPU call 1: (a) issue rates of plain 32-bit ops, (b) 128-B gather rate vs table size (TLB reach),
# (c) PLONK_MSM_ORDER=1 at 2^20 with 32 / 64-entry slices vs the default, same box.  Output: gpurun_out/r3a/
set -u
O=gpurun_out/r3a
rm -rf $O; mkdir -p $O
timeout 120 build/ubench/rates32 > $O/rates32.txt 2>&1
timeout 300 build/ubench/gather > $O/gather.txt 2>&1
B="python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 2"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 150 $B > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])
except Exception as e:
    print('$name FAILED', e)
PY
}
run base X=1
run order32 PLONK_MSM_ORDER=1 PLONK_MSM_KSL=32
run order64 PLONK_MSM_ORDER=1 PLONK_MSM_KSL=64
run ksl64 PLONK_MSM_KSL=64
run base2 X=1
tail -60 $O/rates32.txt
cat $O/gather.txt
