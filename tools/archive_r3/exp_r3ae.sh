#!/bin/bash
# round 3, GPU call 31: one rank of a W-rank sharded proof alone on the GPU (loop-back collectives) at 2^20 and 2^22
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ae
rm -rf $O; mkdir -p $O
timeout 600 python tools/rank_alone.py 20 5 > $O/rank_alone_2p20.jsonl 2> $O/rank_alone_2p20.err; echo "rc=$?"; cat $O/rank_alone_2p20.jsonl; tail -3 $O/rank_alone_2p20.err
timeout 600 python tools/rank_alone.py 16 10 > $O/rank_alone_2p16.jsonl 2> $O/rank_alone_2p16.err; echo "rc=$?"; cat $O/rank_alone_2p16.jsonl; tail -3 $O/rank_alone_2p16.err
