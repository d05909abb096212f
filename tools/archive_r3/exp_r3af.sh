#!/bin/bash
# round 3, GPU call 32: rank 0 of a W-rank sharded proof alone at 2^22 gates (BASELINE config 5's size)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3af
rm -rf $O; mkdir -p $O
timeout 900 python tools/rank_alone.py 22 3 2,4,8 > $O/rank_alone_2p22.jsonl 2> $O/rank_alone_2p22.err; echo "rc=$?"; cat $O/rank_alone_2p22.jsonl; tail -3 $O/rank_alone_2p22.err
