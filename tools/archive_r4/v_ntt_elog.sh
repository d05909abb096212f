#!/bin/bash
# round 4, GPU call: NTT pass kernels with 4 elements per lane (PLONK_NTT_ELOG=2: 1024-element tiles, <= 128 VGPRs, four waves
# per SIMD) against 8 (ELOG=3, two waves): parity, standalone transforms, whole proofs at 2^16 / 2^20 / 2^22 (same box)
set -u
R=$GRAFT_REPO_ROOT
cd $R
PLONK_NTT_ELOG=2 timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k "small_single or two_pass or three_pass or batch_of_five or ntt_batch_entry or full_size_properties or empty_input or truncates" 2>&1 | tail -3
cd /tmp
for E in 3 2 3 2; do
  echo "== standalone transforms ELOG=$E"
  PLONK_NTT_ELOG=$E python $R/tools/ntt_passes.py 20 20
done
for E in 3 2; do
  echo "== standalone transforms 2^16 / 2^22 gates ELOG=$E"
  PLONK_NTT_ELOG=$E python $R/tools/ntt_passes.py 16 50
  PLONK_NTT_ELOG=$E python $R/tools/ntt_passes.py 22 10
done
line() { python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12])"; }
for E in 3 2 3 2; do
  PLONK_NTT_ELOG=$E python $R/bench.py --no-cpu-baseline --no-extras --log-gates 20 --steps 8 --warmup 2 2>/dev/null | line "2^20 ELOG=$E"
done
for E in 3 2 3 2; do
  PLONK_NTT_ELOG=$E python $R/bench.py --no-cpu-baseline --no-extras --log-gates 16 --steps 30 --warmup 3 2>/dev/null | line "2^16 ELOG=$E"
done
for E in 3 2; do
  PLONK_NTT_ELOG=$E python $R/bench.py --no-cpu-baseline --no-extras --log-gates 12 --steps 50 --warmup 3 2>/dev/null | line "2^12 ELOG=$E"
done
for E in 3 2; do
  PLONK_NTT_ELOG=$E python $R/bench.py --no-cpu-baseline --no-extras --log-gates 22 --steps 3 --warmup 1 2>/dev/null | line "2^22 ELOG=$E"
done
