#!/bin/bash
# round 5, GPU call c: same-box A/B — round-4 tree vs this tree, 13-slot partition, side-stream CU mask, deferred-scope geometry
mkdir -p gpurun_out/r5c
exec > gpurun_out/r5c/log.txt 2>&1
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py
Q="python tools/archive_r5/quick.py"
O=gpurun_out/r5c/ab.jsonl
QUICK_BENCH=build/r4tree/bench.py $Q r4tree --steps 10 --warmup 2 >> $O
$Q r5 --steps 10 --warmup 2 >> $O
QUICK_BENCH=build/r4tree/bench.py $Q r4tree_again --steps 10 --warmup 2 >> $O
$Q r5_again --steps 10 --warmup 2 >> $O
PLONK_MSM_SORT13=1 $Q sort13 --steps 10 --warmup 2 >> $O
for k in 16 32 64; do PLONK_SIDE_CUS=$k $Q side_cus_$k --steps 10 --warmup 2 >> $O; done
PLONK_MSM_SORT13=1 PLONK_SIDE_CUS=32 $Q sort13_cus32 --steps 10 --warmup 2 >> $O
for lg in 16 18; do
  $Q base_2p$lg --log-gates $lg --steps 30 --warmup 3 >> $O
  PLONK_SIDE_AFTER_ELOG=3 $Q after_elog3_2p$lg --log-gates $lg --steps 30 --warmup 3 >> $O
  PLONK_SIDE_AFTER_ELOG=2 $Q after_elog2_2p$lg --log-gates $lg --steps 30 --warmup 3 >> $O
done
$Q base_2p22 --log-gates 22 --steps 3 --warmup 1 >> $O
PLONK_SIDE_CUS=32 PLONK_MSM_SORT13=1 $Q sort13_cus32_2p22 --log-gates 22 --steps 3 --warmup 1 >> $O
cat $O
PLONK_MSM_SORT13=0 timeout 200 python tools/msm_phases.py 20 > gpurun_out/r5c/phases_sort13_0.jsonl
PLONK_MSM_SORT13=1 timeout 200 python tools/msm_phases.py 20 > gpurun_out/r5c/phases_sort13_1.jsonl
cat gpurun_out/r5c/phases_sort13_*.jsonl
timeout 600 python -m pytest tests/test_gpu_msm_variants.py tests/test_gpu_config.py -q -m gpu -k "SORT13 or config or layout or budget" 2>&1 | tail -15
