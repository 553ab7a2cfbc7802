#!/bin/bash
# Development aid: same-box A/B of the builds under ab_libs/ on a secondary bench line.  usage: ab_model.sh <reps> <bench args...>
R=$(pwd); REPS=$1; shift
cp $R/diffsim_amd/libdiffsim_amd.so /tmp/lib_orig.so
for rep in $(seq 1 $REPS); do
  for f in $R/ab_libs/lib_*.so; do
    v=$(basename $f .so); v=${v#lib_}
    cp $f $R/diffsim_amd/libdiffsim_amd.so
    python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']
print('$v', d['value'], d['ms_per_step'], {n:v['ms'] for n,v in sorted(k.items(), key=lambda x:-x[1]['ms'])[:5]}, d.get('score_sample',[None])[:2])"
  done
done
cp /tmp/lib_orig.so $R/diffsim_amd/libdiffsim_amd.so
