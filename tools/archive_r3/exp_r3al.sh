#!/bin/bash
# round 3, GPU call 38: whole GPU suite on the final build (rowcol lanes rule), then the default bench line
set -u
O=gpurun_out/r3al
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -12 $O/gpu_suite.log
