#!/bin/bash
# round 4, GPU call 1: same-box baselines of the r03 build (2^12 / 2^16 / 2^20, rank alone W = 2 / 8) and SQ / I-cache counters
# of every kernel of a 2^16 proof (why is msm_bucket_sum 3x slower than its addition count says?)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for LG in 16 12 20; do
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps $([ $LG -ge 20 ] && echo 10 || echo 30) --warmup 3 > $O/bench_$LG.json 2> $O/bench_$LG.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$LG.json').read().strip().splitlines()[-1])
print('2^$LG', d['value'], {k: round(v, 3) for k, v in d.get('kernel_ms_per_prove', {}).items()})
PY
done
python $R/tools/rank_alone.py 20 5 2,8 > $O/rank_alone_2p20.jsonl 2> $O/rank_alone.err; cat $O/rank_alone_2p20.jsonl
python $R/tools/rank_alone.py 16 20 1,8 > $O/rank_alone_2p16.jsonl 2>> $O/rank_alone.err; cat $O/rank_alone_2p16.jsonl
B="python $R/bench.py --no-cpu-baseline --no-extras --log-gates 16 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t16 -o bench -- $B > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ('pmc1', 'pmc2'):
    f = glob.glob('$O/%s/*counter_collection.csv' % tag)
    if not f: print(tag, 'no counters'); continue
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void plonk::', '').replace('plonk::', '')
        d[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
    for k, v in sorted(d.items()):
        w = max(v.get('SQ_WAVES', 1), 1)
        print(tag, k[:52], 'launches', n[k], 'waves/launch %.0f' % (w / max(n[k], 1)), {c: '%.4g' % (x / w) for c, x in sorted(v.items()) if c != 'SQ_WAVES'})
PY
T=$(find $O/t16 -name "*kernel_stats.csv" | head -1); head -40 $T | cut -c1-160
python $R/tools/timeline.py $(find $O/t16 -name "*kernel_trace.csv" | head -1) > $O/timeline_16.txt
find $O -name "*.db" -delete
