#!/bin/bash
# round 6, second session: the final library under rocprofv3 — stats, PMC passes, timeline — and the driver's default bench
# line, then the rank-alone table; everything lands under gpurun_out/prof_r06z (tools/summarize_profile.py r06z turns it into
# profiles/r06z) and gpurun_out/r6b/final
set -u
R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf gpurun_out/prof_r06z; mkdir -p gpurun_out/prof_r06z gpurun_out/r6b/final
python __graft_entry__.py > gpurun_out/prof_r06z/build.log 2>&1
bash tools/profile_bench.sh r06z --no-extras > gpurun_out/prof_r06z.log 2>&1
python tools/timeline.py gpurun_out/prof_r06z/trace/bench_kernel_trace.csv > gpurun_out/prof_r06z/timeline_2p20.txt 2>&1; head -1 gpurun_out/prof_r06z/timeline_2p20.txt; tail -1 gpurun_out/prof_r06z/timeline_2p20.txt
find gpurun_out/prof_r06z -name "*.db" -delete
cd /tmp
( time python $R/bench.py > $R/gpurun_out/prof_r06z/bench_default_line.json 2> $R/gpurun_out/prof_r06z/bench_default.err ) 2>&1 | grep real
cut -c1-300 $R/gpurun_out/prof_r06z/bench_default_line.json
cd $R
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for w in 1 2 4 8; do python tools/rank_alone.py 20 10 $w 2>/dev/null; done > gpurun_out/r6b/final/rank_alone_final_2p20.jsonl
cut -c1-200 gpurun_out/r6b/final/rank_alone_final_2p20.jsonl
python tools/host_gaps.py 12 16 17 18 19 20 2>/dev/null > gpurun_out/r6b/final/host_gaps_final.jsonl
cut -c1-260 gpurun_out/r6b/final/host_gaps_final.jsonl
