#!/bin/bash
# round 5, LAST GPU call: the final library (after the host-only edits that followed profiles/r05x) under rocprofv3 — stats, PMC
# passes, timeline — and the driver's default bench line
set -u
R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/prof_r05z
python __graft_entry__.py > gpurun_out/prof_r05z/build.log 2>&1
bash tools/profile_bench.sh r05z --no-extras > gpurun_out/prof_r05z.log 2>&1
python tools/timeline.py gpurun_out/prof_r05z/trace/bench_kernel_trace.csv > gpurun_out/prof_r05z/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r05z/timeline_2p20.txt; tail -1 gpurun_out/prof_r05z/timeline_2p20.txt
find gpurun_out/prof_r05z -name "*.db" -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r05z/bench_default_line.json 2> $R/gpurun_out/prof_r05z/bench_default.err ) 2>&1 | grep real
cut -c1-300 $R/gpurun_out/prof_r05z/bench_default_line.json
cd $R
timeout 900 python -m pytest tests/test_gpu_standin_transport.py tests/test_gpu_msm_variants.py tests/test_gpu_config.py -q -m "gpu and not slow" 2>&1 | tail -4
