#!/bin/bash
# Development aid: same-box A/B of two builds of libdiffsim_amd.so.  Put the two builds at ab_libs/lib_base.so and
# ab_libs/lib_nt.so (any variant), then: gpurun -- ./tools/ab_bench.sh
R=$(pwd)
for rep in 1 2; do
  for v in base nt; do
    cp $R/ab_libs/lib_$v.so $R/diffsim_amd/libdiffsim_amd.so
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp $R/ab_libs/lib_base.so $R/diffsim_amd/libdiffsim_amd.so
