#!/bin/bash
# round 3, GPU call 18: the whole GPU suite on the final MSM layout
set -u
O=gpurun_out/r3r
rm -rf $O; mkdir -p $O
timeout 2300 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -30 $O/gpu_suite.log
