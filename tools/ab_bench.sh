#!/bin/bash
# Development aid: same-box A/B of several builds of libdiffsim_amd.so.  Put the builds at ab_libs/lib_<name>.so
# (ab_libs/ is git-ignored but travels with gpurun), then: gpurun -- ./tools/ab_bench.sh [reps] [kernel families to print ...]
R=$(pwd)
REPS=${1:-2}
shift
FAMS=${@:-gemm_bf16_256x256_linear_geglu}
cp $R/diffsim_amd/libdiffsim_amd.so /tmp/lib_orig.so
for rep in $(seq 1 $REPS); do
  for f in $R/ab_libs/lib_*.so; do
    v=$(basename $f .so); v=${v#lib_}
    cp $f $R/diffsim_amd/libdiffsim_amd.so
    python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-product-default --no-pixels-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']
print('$v', d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (f.replace('gemm_bf16_',''), k[f]['ms']) for f in '$FAMS'.split() if f in k), d['score_sample'][:2])"
  done
done
cp /tmp/lib_orig.so $R/diffsim_amd/libdiffsim_amd.so
