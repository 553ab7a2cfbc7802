#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02_tests_e.txt 2>&1; echo tests rc=$?; tail -15 gpurun_out/r02_tests_e.txt
python3 bench.py --steps 20 --warmup 5 --dump-launches gpurun_out/r02e_launches.jsonl > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.log; echo bench rc=$?
python3 -c "import json;d=json.loads(open('gpurun_out/r02e_bench.json').read().splitlines()[-1]);k=d['kernel_breakdown_ms_per_step'];print(d['value'],d['ms_per_step'],d['parity_vs_cpu_oracle'],k.get('gemm_bf16_256x256_linear_geglu'))"
