#!/bin/bash
# round-2 first GPU call: overlap micro-benchmark + baseline bench lines (evidence for the README/DESIGN numbers)
R=$(pwd); mkdir -p gpurun_out
./tools/ubench_overlap > gpurun_out/r02_ubench_overlap.txt 2>&1; echo ubench rc=$?
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench_sd15.json 2> gpurun_out/r02a_bench_sd15.log; echo sd15 rc=$?
python3 bench.py --model sdxl --batch-pairs 8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_bench_sdxl.json 2> gpurun_out/r02a_bench_sdxl.log; echo sdxl rc=$?
python3 bench.py --model dit --batch-pairs 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02a_bench_dit.json 2> gpurun_out/r02a_bench_dit.log; echo dit rc=$?
python3 bench.py --model dit --fp8-attention --batch-pairs 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02a_bench_dit_fp8.json 2> gpurun_out/r02a_bench_dit_fp8.log; echo ditfp8 rc=$?
python3 bench.py --pixels-in --batch-pairs 16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_bench_pixels_in.json 2> gpurun_out/r02a_bench_pixels_in.log; echo pixels rc=$?
tail -c 400 gpurun_out/r02a_bench_sd15.json; cat gpurun_out/r02_ubench_overlap.txt
