#!/bin/bash
# A/B of the MSM tail variants on one box (PLONK_MSM_TAIL, PLONK_MSM_KSL); output under gpurun_out/exp_tail/
set -u
O=gpurun_out/exp_tail
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_soak.py -m gpu -x -q --durations=8 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
B="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
for v in quad serial; do
  PLONK_MSM_TAIL=$v timeout 120 $B > $O/b20_$v.json 2> $O/b20_$v.err
  PLONK_MSM_TAIL=$v timeout 120 $B --log-gates 16 --steps 20 > $O/b16_$v.json 2> $O/b16_$v.err
done
for k in 64 128; do
  PLONK_MSM_KSL=$k timeout 120 $B > $O/b20_ksl$k.json 2> $O/b20_ksl$k.err
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
