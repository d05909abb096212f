#!/bin/bash
# round 4, GPU call 4: the 2^17-bucket variant (nbm) — parity (edge cases + prover), then 2^19 single GPU and rank-alone W=2 @2^20 / W=8 @2^22
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm_variants.py -x -q -m gpu -k "17 or BSUM" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['value'], {k: round(v, 3) for k, v in d.get('kernel_ms_per_prove', {}).items()}, d.get('proof_blake2b', '')[:16], 'rows', d['roofline'].get('table_rows'))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for V in default window default window; do
  E=""; [ $V = window ] && E="PLONK_MSM_TABLE=window"
  env $E python $R/bench.py --no-cpu-baseline --no-extras --log-gates 19 --steps 10 --warmup 2 > $O/b19_$V.json 2> $O/b19_$V.err
  line $O/b19_$V.json "2^19 $V"
done
for V in default window; do
  E=""; [ $V = window ] && E="PLONK_MSM_TABLE=window"
  env $E python $R/tools/rank_alone.py 20 5 2 2> $O/ra20_$V.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^20 W=2 $V', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
  env $E python $R/tools/rank_alone.py 22 3 8 2> $O/ra22_$V.err | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('rank_alone 2^22 W=8 $V', d['prove_ms_rank_alone'], d['kernel_ms'], d['table_rows'])"
done
python $R/tools/msm_phases.py 19 > $O/phases_19.jsonl 2> $O/phases.err; cat $O/phases_19.jsonl
