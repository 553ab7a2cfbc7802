#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "attention" > gpurun_out/r02_tests_c.txt 2>&1; echo attn tests rc=$?; tail -3 gpurun_out/r02_tests_c.txt
for sp in 0 1 0 1; do
  DSIM_ATTN_SP=$sp python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/r02c_launches_sp$sp.jsonl > gpurun_out/r02c_bench_sp$sp.json 2> gpurun_out/r02c_bench_sp$sp.log; echo sp $sp rc=$?
  python3 -c "import json;d=json.loads(open('gpurun_out/r02c_bench_sp$sp.json').read().splitlines()[-1]);print('sp',$sp,d['value'],d['ms_per_step'],d['score_sample'], d['kernel_breakdown_ms_per_step'].get('attention_bf16_d40'), d['kernel_breakdown_ms_per_step'].get('attention_bf16_d80'))"
done
