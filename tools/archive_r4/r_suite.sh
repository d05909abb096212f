#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
( time python -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/r4r/pytest.log 2>&1 ) 2>&1 | grep real
echo "pytest rc=$?"; tail -12 gpurun_out/r4r/pytest.log | cut -c1-160
