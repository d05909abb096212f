#!/bin/bash
# the gap between `value` (wire columns resident) and a proof from pinned HOST wire columns, same box:
#   PLONK_WIRE_BY_COLUMN=0  one grouped commitment launch after the last copy (round 5)
#   PLONK_WIRE_BY_COLUMN=2  column by column on the main stream
#   default                 column by column on two streams
out=${1:-gpurun_out/r06d}
mkdir -p $out
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6
for lg in 20 19; do
for v in 0 2 1 0 1; do
  PLONK_WIRE_BY_COLUMN=$v python bench.py --log-gates $lg --no-cpu-baseline --host-wires-only --steps 10 2>>$out/ab_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'log_gates': $lg, 'PLONK_WIRE_BY_COLUMN': $v, 'value': d['value'], 'prove_ms_host_wires_pinned': d.get('prove_ms_host_wires_pinned'), 'gap_ms': round(d.get('prove_ms_host_wires_pinned', 0) - d['value'], 3), 'error': d.get('host_wires_error'), 'proof': d['proof_blake2b']}))"
done
done | tee $out/host_wires_ab.jsonl
