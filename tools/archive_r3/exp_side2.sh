#!/bin/bash
set -u
O=gpurun_out/exp_side2; rm -rf $O; mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1"
for v in 2 0 2 0; do
  PLONK_SIDE_DEFER=$v timeout 40 $B > $O/b20_$v.json 2> $O/b20_$v.err
  python -c "
import json;j=json.loads(open('$O/b20_$v.json').read().strip().splitlines()[-1]);print('defer=$v 2^20', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
done
