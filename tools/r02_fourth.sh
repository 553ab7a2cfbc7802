#!/bin/bash
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --dump-launches gpurun_out/r02d_launches_$name.jsonl > gpurun_out/r02d_bench_$name.json 2> gpurun_out/r02d_bench_$name.log; echo $name rc=$?
  python3 -c "import json;d=json.loads(open('gpurun_out/r02d_bench_$name.json').read().splitlines()[-1]);k=d['kernel_breakdown_ms_per_step'];print('$name',d['value'],d['ms_per_step'], 'd40',k.get('attention_bf16_d40'),'d80',k.get('attention_bf16_d80'),'lin_res',k.get('gemm_bf16_256x320_linear_res'),'lin',k.get('gemm_bf16_256x320_linear'), k.get('gemm_bf16_128x160_linear_res'), k.get('gemm_bf16_128x160_linear'))"
}
run base DSIM_ATTN_SP=0
run sp3 DSIM_ATTN_SP=1
run sp2 DSIM_ATTN_SP=1 DSIM_ATTN_SP_LDS=81920
run sp1 DSIM_ATTN_SP=1 DSIM_ATTN_SP_LDS=163840
run k320 DSIM_ATTN_SP=0 DSIM_SMALLK_BM128=320
run k640 DSIM_ATTN_SP=0 DSIM_SMALLK_BM128=640
run k1280 DSIM_ATTN_SP=0 DSIM_SMALLK_BM128=1280
