#!/bin/bash
# round 5, GPU call e: the whole GPU suite with durations
mkdir -p gpurun_out/r5e
exec > gpurun_out/r5e/log.txt 2>&1
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py
date +%s
time timeout 1700 python -m pytest tests -q -m gpu --durations=50 2>&1 | tail -110
date +%s
