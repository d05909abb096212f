#!/bin/bash
# GPU box: host helper threads of fetch_commitments (finish_pool.hpp) on / off, same library, same box, three repetitions;
# + one rank of 8 alone (where the partial sums of the ranks are added on the host as well)
out=${1:-gpurun_out/r6b/pool}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in 1 2 3; do
  for t in 0 3; do
    PLONK_HOST_THREADS=$t python tools/host_gaps.py 12 16 17 20 2>>$out/err.txt | sed "s/^{/{\"host_threads\": $t, /"
  done
done | tee $out/pool_ab.jsonl
for rep in 1 2; do
  for t in 0 3; do
    PLONK_HOST_THREADS=$t python tools/rank_alone.py 20 10 8 2>>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'host_threads': $t, 'rank_alone_W8_2p20': d['prove_ms_rank_alone']}))"
  done
  PLONK_HIP_LIB=$PWD/build/variants/libplonk_r6a.so python tools/rank_alone.py 20 10 8 2>>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': 'r6a', 'rank_alone_W8_2p20': d['prove_ms_rank_alone']}))"
done | tee $out/pool_rank8.jsonl
