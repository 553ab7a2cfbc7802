#!/bin/bash
# usage: ab_env_model.sh <reps> "<ENV=a>" "<ENV=b>" -- <bench args>
REPS=$1; shift
ENVS=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
for rep in $(seq 1 $REPS); do
  for e in "${ENVS[@]}"; do
    ( export $e; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']
print('$e', d['value'], d['ms_per_step'], {n:v['ms'] for n,v in sorted(k.items(), key=lambda x:-x[1]['ms'])[:5]}, d.get('score_sample',[None])[:2])" )
  done
done
