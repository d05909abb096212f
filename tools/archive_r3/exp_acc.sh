#!/bin/bash
# A/B of the accumulation kernels on one box (PLONK_MSM_ACC=lds|regs); output under gpurun_out/exp_acc/
set -u
O=gpurun_out/exp_acc
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_soak.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
for v in lds regs lds regs; do
  PLONK_MSM_ACC=$v timeout 120 $B > $O/b20_$v.json 2> $O/b20_$v.err
  python -c "
import json;j=json.loads(open('$O/b20_$v.json').read().strip().splitlines()[-1]);print('$v 2^20', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
done
for v in lds regs; do
  PLONK_MSM_ACC=$v timeout 120 $B --log-gates 16 --steps 20 > $O/b16_$v.json 2> $O/b16_$v.err
  PLONK_MSM_ACC=$v timeout 120 $B --profile bench-like > $O/bl_$v.json 2> $O/bl_$v.err
  python -c "
import json
for f in ('b16','bl'):
    j=json.loads(open('$O/'+f+'_$v.json').read().strip().splitlines()[-1]);print('$v',f, j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
done
