#!/bin/bash
# 4 concurrent children per round, 3 rounds, for TIDY=0 and TIDY=1
for t in 1 0; do
  fails=0
  for round in 1 2 3; do
    pids=()
    for j in 1 2 3 4; do
      ( PLONK_MSM_TIDY=$t PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=15 python -m pytest tests/test_gpu_prover.py tests/test_gpu_prove_sizes.py -x -q -m "gpu and not slow" -k "deterministic_v3 or random_arithmetic or (proof_bytes_equal_c_oracle and not 16 and not 2p20)" > /tmp/child_${t}_${round}_${j}.log 2>&1; echo $? > /tmp/child_${t}_${round}_${j}.rc ) &
      pids+=($!)
    done
    wait
    for j in 1 2 3 4; do rc=$(cat /tmp/child_${t}_${round}_${j}.rc); if [ "$rc" != "0" ]; then fails=$((fails+1)); grep -E "^FAILED|Error|Unsatisfied" /tmp/child_${t}_${round}_${j}.log | head -3; fi; done
  done
  echo "TIDY=$t: $fails failures of 12 children"
done
