#!/bin/bash
# round 3, GPU call 28: instruction-cache counters — the straight-line NTT pass kernels (~190 KB of code) vs the looping msm_accumulate
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/ntt_a -o p -- python $GRAFT_REPO_ROOT/tools/ntt_passes.py 20 2 > $O/ntt_a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/ntt_b -o p -- python $GRAFT_REPO_ROOT/tools/ntt_passes.py 20 2 > $O/ntt_b.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/msm_a -o p -- python $GRAFT_REPO_ROOT/bench.py --log-gates 18 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/msm_a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/msm_b -o p -- python $GRAFT_REPO_ROOT/bench.py --log-gates 18 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/msm_b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ('ntt_a', 'ntt_b', 'msm_a', 'msm_b'):
    f = glob.glob('$O/%s/*counter_collection.csv' % tag)
    if not f: print(tag, 'no counters'); continue
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void plonk::', '')
        if 'ntt_pass' not in k and 'msm_accumulate' not in k and 'quotient' not in k: continue
        d[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in d.items():
        print(tag, k[:44], {c: '%.3g' % x for c, x in sorted(v.items())})
PY
find $O -name "*.db" -delete
