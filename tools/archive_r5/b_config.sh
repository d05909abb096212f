#!/bin/bash
# round 5, GPU call b: config struct / plan assertions / stand-in transport after the staging fix, a short bench line
mkdir -p gpurun_out/r5b
exec > gpurun_out/r5b/log.txt 2>&1
set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py
timeout 1200 python -m pytest tests/test_gpu_config.py tests/test_gpu_msm.py tests/test_gpu_msm_variants.py tests/test_gpu_standin_transport.py -q -m "gpu and not slow" --durations=25 2>&1 | tail -70
timeout 600 python bench.py --steps 10 --warmup 2 --no-2p22 --no-cpu-baseline > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
cat gpurun_out/r5b/bench.json; tail -5 gpurun_out/r5b/bench.err
