#!/bin/bash
# round 4, GPU call: the side-transform schedule again now that deferred transforms run the four-wave NTT kernels:
# PLONK_SIDE_DEFER unset (by size) / 0 / 1 / 2 at 2^19 and 2^20 gates (dense) and at 2^20 with every widget
set -u
R=$GRAFT_REPO_ROOT
cd /tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['kernel_ms_per_prove'], d['proof_blake2b'][:12])"; }
for LG in 19 20; do
for V in default 1 2 default 1 2; do
  if [ $V = default ]; then unset PLONK_SIDE_DEFER; else export PLONK_SIDE_DEFER=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --log-gates $LG --steps 10 --warmup 2 2>/dev/null | line "dense 2^$LG DEFER=$V"
done
done
for V in default 1 2 default 1 2; do
  if [ $V = default ]; then unset PLONK_SIDE_DEFER; else export PLONK_SIDE_DEFER=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --profile widgets --log-gates 20 --steps 8 --warmup 2 2>/dev/null | line "widgets 2^20 DEFER=$V"
done
for V in default 1; do
  if [ $V = default ]; then unset PLONK_SIDE_DEFER; else export PLONK_SIDE_DEFER=$V; fi
  python $R/bench.py --no-cpu-baseline --no-extras --profile bench-like --log-gates 20 --steps 8 --warmup 2 2>/dev/null | line "bench-like 2^20 DEFER=$V"
done
