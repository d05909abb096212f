#!/bin/bash
# round 3, GPU call 9: the whole GPU suite (with the new 2^20 / 2^22 parity tests) on the build with 5-byte intermediate words
set -u
O=gpurun_out/r3i
rm -rf $O; mkdir -p $O
timeout 2300 python -m pytest tests -m gpu -q --durations=30 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -50 $O/gpu_suite.log
