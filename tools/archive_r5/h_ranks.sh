#!/bin/bash
# round 5, GPU call h: one rank of W alone on the GPU (loop-back collectives), final build, one box: 2^20 W = 1 / 2 / 4 / 8, 2^22 W = 8, 2^16 W = 1 / 8
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/rank_alone.py 20 5 1,2,4,8 > $O/rank_alone_final_2p20.jsonl 2> $O/err20.txt
cat $O/rank_alone_final_2p20.jsonl
timeout 300 python tools/rank_alone.py 16 20 1,8 > $O/rank_alone_final_2p16.jsonl 2> $O/err16.txt
cat $O/rank_alone_final_2p16.jsonl
timeout 600 python tools/rank_alone.py 22 3 8 > $O/rank_alone_final_2p22.jsonl 2> $O/err22.txt
cat $O/rank_alone_final_2p22.jsonl
