#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py::test_vae_sd15_512px tests/test_gpu_round2.py::test_cli_end_to_end -q > gpurun_out/r02_tests_b.txt 2>&1; echo tests rc=$?
for mb in 0 24 48 96; do
  DSIM_CHUNK_MB=$mb python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-launches gpurun_out/r02b_launches_chunk$mb.jsonl > gpurun_out/r02b_bench_chunk$mb.json 2> gpurun_out/r02b_bench_chunk$mb.log; echo chunk $mb rc=$?
  python3 -c "import json;d=json.loads(open('gpurun_out/r02b_bench_chunk$mb.json').read().splitlines()[-1]);print('chunk',$mb,d['value'],d['ms_per_step'],d['score_sample'])"
done
python tools/sdxl_f64_probe.py > gpurun_out/r02_sdxl_f64_probe.txt 2>&1; tail -4 gpurun_out/r02_sdxl_f64_probe.txt
grep -E "assert|Error|passed|failed" gpurun_out/r02_tests_b.txt | head -20
