#!/bin/bash
# Round 6: four provers on ONE GPU at the same time, each in its own process with a fresh context — the contention under which a
# context's FIRST proof lost the race against the fill of a twiddle table created on the starved side stream (ntt.hip ntt_tables;
# profiles/r06/SUMMARY.md section 4: 3 failures in 24 children before the fix, 0 in 24 after).
#   tools/r6_stress_children.sh [rounds = 6]      -> "N failures of M children"
rounds=${1:-6}
fails=0; total=0
for round in $(seq 1 $rounds); do
  for j in 1 2 3 4; do
    ( PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=15 python -m pytest tests/test_gpu_prover.py tests/test_gpu_prove_sizes.py -x -q -m "gpu and not slow" \
        -k "deterministic_v3 or random_arithmetic or (proof_bytes_equal_c_oracle and not 16 and not 2p20)" > /tmp/child_${round}_${j}.log 2>&1; echo $? > /tmp/child_${round}_${j}.rc ) &
  done
  wait
  for j in 1 2 3 4; do
    total=$((total+1))
    if [ "$(cat /tmp/child_${round}_${j}.rc)" != "0" ]; then fails=$((fails+1)); grep -E "^FAILED|Error|Unsatisfied" /tmp/child_${round}_${j}.log | head -3; fi
  done
done
echo "$fails failures of $total children"
