#!/bin/bash
# Development aid: same-box A/B of one build under different environment switches.
# usage: gpurun -- bash tools/ab_env.sh <reps> "<VAR=a VAR2=b>" "<VAR=c>" ... -- fam1 fam2 ...
REPS=$1; shift
ENVS=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
FAMS=${@:-layernorm_bf16 groupnorm_bf16}
for rep in $(seq 1 $REPS); do
  for e in "${ENVS[@]}"; do
    ( export $e; python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']
print('$e', d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (f, k[f]['ms']) for f in '$FAMS'.split() if f in k), d['score_sample'][:2])" )
  done
done
