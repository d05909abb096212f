#!/bin/bash
# round 3, GPU call 33: whole GPU suite on the build with direct NTT twiddles, the new slice rule, G2 validation, rank_alone tool
set -u
O=gpurun_out/r3ag
rm -rf $O; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -16 $O/gpu_suite.log
