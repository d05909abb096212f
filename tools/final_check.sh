#!/bin/bash
# End-of-round check on one box: GPU suite, rocprofv3 kernel stats of the bench, plain default bench line.
# Output under gpurun_out/final/ (tools/summarize_profile.py-style inputs: trace/bench_kernel_stats.csv, bench_kernel_trace.csv)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -m pytest tests -m gpu -x -q --durations=6 --deselect tests/test_gpu_prove_sizes.py::test_proof_bytes_equal_c_oracle_2p20 -k "not three_pass" > $O/gpu_suite.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite.log
tail -12 $O/gpu_suite.log
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/bench_trace.log 2>&1
T=$(find $O/trace -name "bench_kernel_trace.csv" | head -1)
S=$(find $O/trace -name "bench_kernel_stats.csv" | head -1)
python $R/tools/timeline.py $T > $O/timeline_20.txt 2>&1
cp $S $O/bench_kernel_stats.csv
python - "$T" > $O/acc_launches.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
acc = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "msm_accumulate" in r["Kernel_Name"]]
n = (len(acc) - 4) // 4
d = acc[-4 * n:]
print(f"{len(d)} launches in {n} proofs, average {sum(d)/len(d)/1e6:.3f} ms per launch, {4*sum(d)/len(d)/1e6:.2f} ms per proof")
PY
rm -rf $O/trace
grep -h "metric" $O/bench_trace.log | cut -c1-300
cd $R
timeout 150 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default_line.json
