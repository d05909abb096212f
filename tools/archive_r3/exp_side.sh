#!/bin/bash
# A/B of the side-stream schedule (PLONK_SIDE_DEFER=0|1) on one box; output under gpurun_out/exp_side/
set -u
O=gpurun_out/exp_side
rm -rf $O; mkdir -p $O
PLONK_SIDE_DEFER=1 timeout 300 python -m pytest tests/test_gpu_prover.py tests/test_gpu_soak.py tests/test_widget_semantics.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
for v in 1 0 1 0; do
  PLONK_SIDE_DEFER=$v timeout 120 $B > $O/b20_$v.json 2> $O/b20_$v.err
  python -c "
import json;j=json.loads(open('$O/b20_$v.json').read().strip().splitlines()[-1]);print('defer=$v 2^20', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
done
for v in 1 0; do
  PLONK_SIDE_DEFER=$v timeout 120 $B --log-gates 16 --steps 20 > $O/b16_$v.json 2> $O/b16_$v.err
  PLONK_SIDE_DEFER=$v timeout 120 $B --profile widgets > $O/wd_$v.json 2> $O/wd_$v.err
  python -c "
import json
for f in ('b16','wd'):
    j=json.loads(open('$O/'+f+'_$v.json').read().strip().splitlines()[-1]);print('defer=$v',f, j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
done
