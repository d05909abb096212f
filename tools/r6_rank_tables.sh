#!/bin/bash
# bit-position rows on the slices of a W = 4 / 8 rank — measured, not estimated.
# One rank alone (loop-back collectives), 2^20 gates: today's default (window rows over 2^15 buckets for slices of
# <= 2^18 + 64 points) against bit-position rows over 2^19 and 2^15 buckets, same box, same build.
out=${1:-gpurun_out/r06b}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6
for W in 4 8; do
  for cfg in "default" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=19" "PLONK_MSM_TABLE=bitpos PLONK_MSM_BUCKETS=15" "PLONK_MSM_TABLE=halfpos PLONK_MSM_BUCKETS=19"; do
    for split in range commitment; do
      [ $W = 8 ] && [ $split = commitment ] && continue
      if [ "$cfg" = default ]; then env PLONK_BENCH_WIRE_SPLIT=$split python tools/rank_alone.py 20 8 $W
      else env $cfg PLONK_BENCH_WIRE_SPLIT=$split python tools/rank_alone.py 20 8 $W; fi
    done
  done
done > $out/rank_alone_table_rows_2p20.jsonl 2> $out/rank_alone_table_rows_2p20.err
cat $out/rank_alone_table_rows_2p20.jsonl
