#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_gpu_prove_sizes.py tests/test_gpu_multirank.py -x -q -m gpu -k "crossover or (2p20 and 4)" --durations=5 2>&1 | tail -12 ) 2>&1 | cut -c1-200
