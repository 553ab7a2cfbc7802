#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round2.py tests/test_gpu_e2e.py -q > gpurun_out/r02_tests_h.txt 2>&1; echo tests rc=$?; tail -4 gpurun_out/r02_tests_h.txt
for ns in 1 2 3 1 2; do
  python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams $ns > gpurun_out/r02h_bench_s$ns.json 2> gpurun_out/r02h_bench_s$ns.log; echo streams $ns rc=$?
  python3 -c "import json;d=json.loads(open('gpurun_out/r02h_bench_s$ns.json').read().splitlines()[-1]);print('streams',$ns,d['value'],d['ms_per_step'],d['score_sample'])"
done
python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams 2 --batch-pairs 24 | python3 -c "import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('2x24',d['value'],d['ms_per_step'])"
python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams 2 --batch-pairs 40 | python3 -c "import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('2x40',d['value'],d['ms_per_step'])"
