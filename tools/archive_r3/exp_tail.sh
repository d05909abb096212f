#!/bin/bash
# A/B of the MSM tail variants on one box (PLONK_MSM_TAIL=quad|serial); output under gpurun_out/exp_tail/
set -u
O=gpurun_out/exp_tail
rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_soak.py tests/test_widget_semantics.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -12 $O/tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
for v in quad serial; do
  for p in bench-like widgets; do
    PLONK_MSM_TAIL=$v timeout 120 $B --profile $p > $O/b20_${p}_$v.json 2> $O/b20_${p}_$v.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp_tail/b*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["kernel_ms_per_prove"], j["proof_blake2b"])
    except Exception as e:
        print(f, "FAILED", e)
PY
