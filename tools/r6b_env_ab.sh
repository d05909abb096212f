#!/bin/bash
# GPU box, round 6 second session — same-box A/B of environment knobs on the tree's library: prove() per size with no knob
# and under each given setting ("NAME=v" or "NAME=v,NAME2=w"), interleaved, three repetitions (tools/host_gaps.py).
#   usage: bash tools/r6b_env_ab.sh OUT_FILE "log_gates ..." SETTING [SETTING ...]
out=$1; sizes=$2; shift 2
mkdir -p $(dirname $out)
export PLONK_CIRCUIT_CACHE=/tmp/plonk_circuits_r6b
for rep in $(seq 1 ${AB_REPS:-3}); do
  for cfg in none "$@"; do
    if [ $cfg = none ]; then envs=""; else envs=$(echo $cfg | tr ',' ' '); fi
    env $envs python tools/host_gaps.py $sizes 2>>$out.err | sed "s/^{/{\"env\": \"$cfg\", /"
  done
done | tee $out
