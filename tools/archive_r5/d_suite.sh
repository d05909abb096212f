#!/bin/bash
# round 5, GPU call d: the whole GPU suite with durations, rank-alone A/B of the by-commitment wire split
mkdir -p gpurun_out/r5d
exec > gpurun_out/r5d/log.txt 2>&1
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py
for split in range commitment; do
  PLONK_BENCH_WIRE_SPLIT=$split timeout 400 python tools/rank_alone.py 20 5 2,4 >> gpurun_out/r5d/rank_alone_$split.jsonl
done
cat gpurun_out/r5d/rank_alone_*.jsonl
time timeout 1500 python -m pytest tests -q -m gpu --durations=45 -x 2>&1 | tail -90
