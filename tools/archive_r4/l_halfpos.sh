#!/bin/bash
# round 4, GPU call 12: half-density tables (a row for every second bit position): parity (edge cases, prover), then 2^20 forced and 2^22 default
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm_variants.py -x -q -m gpu -k "halfpos" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
cd /tmp
for V in "X=1" "PLONK_MSM_TABLE=halfpos"; do
  env $V python $R/bench.py --no-cpu-baseline --no-extras --log-gates 20 --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^20 $V', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12], d['roofline']['table_rows'])"
done
python $R/bench.py --no-cpu-baseline --no-extras --log-gates 22 --steps 4 --warmup 1 2>$O/b22.err | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('2^22', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:16], d['roofline']['table_rows'], d['config']['setup_s'])"
python $R/tools/msm_phases.py 22 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('phases 2^22', d['prove_ms'], 'g4', d['groups_of_4'], 'g12', d['groups_of_1_2'])"
