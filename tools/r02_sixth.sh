#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_vae.py tests/test_gpu_sdxl.py -q -x > gpurun_out/r02_tests_g.txt 2>&1; echo tests rc=$?; tail -5 gpurun_out/r02_tests_g.txt
for v in 0 1 0 1; do
  DSIM_GN_ONEPASS=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/r02g_launches_gn$v.jsonl > gpurun_out/r02g_bench_gn$v.json 2> gpurun_out/r02g_bench_gn$v.log; echo gn $v rc=$?
  python3 -c "import json;d=json.loads(open('gpurun_out/r02g_bench_gn$v.json').read().splitlines()[-1]);k=d['kernel_breakdown_ms_per_step'];print('gn',$v,d['value'],d['ms_per_step'],d['score_sample'],k.get('groupnorm_bf16'))"
done
