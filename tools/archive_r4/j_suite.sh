#!/bin/bash
# round 4, GPU call 10: the whole GPU suite as the driver runs it, with durations
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j
rm -rf $O; mkdir -p $O
cd $R
( time python -m pytest tests/ -x -q -m gpu --durations=25 > $O/pytest.log 2>&1 ) 2>&1 | grep real
echo "rc=$?"; tail -45 $O/pytest.log | cut -c1-200
