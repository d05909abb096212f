#!/bin/bash
# round 3, GPU call 26: NTT passes on their own — per-pass durations and SQ counters, two-level vs direct inter-pass twiddles
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" $O/avail.txt | sort -u | tr '\n' ' '; echo
for M in 0 1; do
  PLONK_NTT_DIRECT=$M python $GRAFT_REPO_ROOT/tools/ntt_passes.py 20 10 > $O/plain$M.txt 2>&1; cat $O/plain$M.txt
  PLONK_NTT_DIRECT=$M rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$M -o p -- python $GRAFT_REPO_ROOT/tools/ntt_passes.py 20 10 > $O/trace$M.log 2>&1
  python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('$O/t$M/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].replace('void plonk::', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'ntt_pass' in r['Kernel_Name']]
# 4 transforms x 11 launches x 3 passes, in order
for t in range(4):
    blk = seq[t * 33:(t + 1) * 33]
    for p in range(3):
        xs = [blk[i][1] for i in range(3 + p, 33, 3)]
        print('direct=$M transform', t, 'pass', p, blk[p][0], 'avg us %.1f' % (sum(xs) / len(xs)))
PY
  PLONK_NTT_DIRECT=$M rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc$M -o p -- python $GRAFT_REPO_ROOT/tools/ntt_passes.py 20 2 > $O/pmc$M.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$O/pmc$M/*counter_collection.csv')
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void plonk::', '')
    if 'ntt_pass' not in k: continue
    d[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k, v in d.items():
    print('direct=$M', k, 'launches', n[k], {c: '%.3g' % x for c, x in v.items()}, 'valu/wave %.0f' % (v['SQ_INSTS_VALU'] / max(v['SQ_WAVES'], 1)), 'wavecyc/wave %.0f' % (v['SQ_WAVE_CYCLES'] / max(v['SQ_WAVES'], 1)))
PY
  find $O -name "*.db" -delete
done
