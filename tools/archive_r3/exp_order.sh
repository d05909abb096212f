#!/bin/bash
# PLONK_MSM_ORDER=1 (lanes in order of slice length): parity of the MSM tests, then prove() at 2^16; output on stdout
export PLONK_MSM_ORDER=1
timeout 14 python -m pytest tests/test_gpu_msm.py -x -q -k "basic or edge or closed_form and 16 or skew or small_scalars" 2>&1 | tail -2
B="python bench.py --log-gates 16 --steps 10 --warmup 1 --no-extras --no-cpu-baseline"
PLONK_MSM_KSL=32 timeout 10 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order ksl32 2^16', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
timeout 10 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order ksl-rule 2^16', j['value'], j['kernel_ms_per_prove'], j['proof_blake2b'])"
