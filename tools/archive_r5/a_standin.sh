#!/bin/bash
# round 5, GPU call a: partition probe (read-only), the stand-in transport tests, phase baseline at 2^20
mkdir -p gpurun_out/r5a
exec > gpurun_out/r5a/log.txt 2>&1
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
( rocm-smi --showcomputepartition; rocm-smi --showmemorypartition; amd-smi partition --current; amd-smi partition --accelerator; nproc; free -g | head -2 ) > gpurun_out/r5a/partition_probe.txt 2>&1
python __graft_entry__.py
timeout 900 python -m pytest tests/test_gpu_standin_transport.py -x -q -m "gpu and not slow" 2>&1 | tail -40
timeout 300 python tools/msm_phases.py 20 > gpurun_out/r5a/phases_2p20.jsonl
cat gpurun_out/r5a/phases_2p20.jsonl
