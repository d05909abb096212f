#!/bin/bash
# round 4, GPU call: 4 elements per lane as the DEFAULT of the NTT pass kernels, 8 under a busy MSM pipeline (prover.hip
# SideScope asks for them): parity of the transforms and of whole proofs, then same-box A/B against ELOG=3 everywhere
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_prover.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_prove_sizes.py -x -q -m gpu -k "test_proof_bytes_equal_c_oracle and not crossover and not 2p20" 2>&1 | tail -2
cd /tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12])"; }
for V in 3 hybrid 3 hybrid 2; do
  if [ $V = hybrid ]; then unset PLONK_NTT_ELOG; else export PLONK_NTT_ELOG=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates 20 --steps 8 --warmup 2 2>/dev/null | line "2^20 ELOG=$V"
done
for LG in 19 18 16 12; do
for V in 3 hybrid 3 hybrid; do
  if [ $V = hybrid ]; then unset PLONK_NTT_ELOG; else export PLONK_NTT_ELOG=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 20 --warmup 3 2>/dev/null | line "2^$LG ELOG=$V"
done
done
for V in 3 hybrid; do
  if [ $V = hybrid ]; then unset PLONK_NTT_ELOG; else export PLONK_NTT_ELOG=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates 22 --steps 3 --warmup 1 2>/dev/null | line "2^22 ELOG=$V"
done
unset PLONK_NTT_ELOG
echo "== rank alone (hybrid, then ELOG=3)"
python $R/tools/rank_alone.py 20 8 2,8 2>/dev/null | cut -c1-330
PLONK_NTT_ELOG=3 python $R/tools/rank_alone.py 20 8 2,8 2>/dev/null | cut -c1-330
