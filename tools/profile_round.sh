#!/bin/bash
# Run on the GPU box (gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/profile_round.sh <tag>   e.g. r02f -> gpurun_out/prof_<tag>_{trace,fetch,write,mfma,tap1,tap2}, gpurun_out/bench_<tag>*.json
# PMC passes are separate runs with --pmc only (no trace domains), as the pool requires.
TAG=$1
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-product-default --no-pixels-leg > $R/gpurun_out/bench_${TAG}_under_rocprof.json 2> $R/gpurun_out/rocprof_${TAG}_trace.log
echo trace rc=$?
find $R/gpurun_out/prof_${TAG}_trace -type f ! -name "*kernel_stats.csv" -delete
for c in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${c%%:*}; ctr=${c#*:}
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/prof_${TAG}_${name} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-product-default --no-pixels-leg > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_${name}.log
  echo $name rc=$?
  # keep only the counter csv (the merged-back directory is size-limited)
  find $R/gpurun_out/prof_${TAG}_${name} -type f ! -name "*counter_collection.csv" -delete
done
# the tap layer alone (north_star MFMA-utilisation target): two SQ passes
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_${TAG}_tap1 -- python3 $R/tools/tap_probe.py > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_tap1.log; echo tap1 rc=$?
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/prof_${TAG}_tap2 -- python3 $R/tools/tap_probe.py > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_tap2.log; echo tap2 rc=$?
for t in tap1 tap2; do find $R/gpurun_out/prof_${TAG}_$t -type f ! -name "*counter_collection.csv" -delete; done
# kernel stats of the secondary configs
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace_sdxl -- python3 $R/bench.py --model sdxl --batch-pairs 8 --steps 3 --warmup 1 --no-cpu-baseline --no-product-default --no-pixels-leg > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_trace_sdxl.log; echo trace_sdxl rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace_pixels -- python3 $R/bench.py --pixels-in --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_trace_pixels.log; echo trace_pixels rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace_dit -- python3 $R/bench.py --model dit --batch-pairs 64 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_trace_dit.log; echo trace_dit rc=$?
for t in trace_sdxl trace_pixels trace_dit; do find $R/gpurun_out/prof_${TAG}_$t -type f ! -name "*kernel_stats.csv" -delete; done
# PMC passes of the secondary models (their roofline.traffic / mfma_util_pmc fields): --pmc only, one counter set per run
for m in dit sdxl; do
  bpm=64; [ $m = sdxl ] && bpm=8
  for c in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=${c%%:*}; ctr=${c#*:}
    rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/prof_${TAG}_${m}_${name} -- python3 $R/bench.py --model $m --batch-pairs $bpm --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-product-default --no-pixels-leg > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_${m}_${name}.log
    echo ${m}_$name rc=$?
    find $R/gpurun_out/prof_${TAG}_${m}_${name} -type f ! -name "*counter_collection.csv" -delete
  done
done
# PMC passes of the pixels-in line (the VAE encoder's kernels: roofline.traffic / mfma_util_pmc of `--pixels-in` and of the default run's pixels_in leg)
for c in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${c%%:*}; ctr=${c#*:}
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/prof_${TAG}_pixels_in_${name} -- python3 $R/bench.py --pixels-in --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_pixels_in_${name}.log
  echo pixels_in_$name rc=$?
  find $R/gpurun_out/prof_${TAG}_pixels_in_${name} -type f ! -name "*counter_collection.csv" -delete
done
# sustained pass: the MFMA-busy / clock counters over 40 back-to-back steps (the held clock of a long run, not of the first seconds)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma40 -- python3 $R/bench.py --steps 40 --warmup 0 --no-cpu-baseline --no-profile --no-product-default --no-pixels-leg > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_mfma40.log; echo mfma40 rc=$?
find $R/gpurun_out/prof_${TAG}_mfma40 -type f ! -name "*counter_collection.csv" -delete
cd $R
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.log; echo bench rc=$?
# BASELINE config 2's size: 157 steps x 64 pairs = 10 048 pairs in one timed region (sustained clock)
python3 bench.py --steps 157 --warmup 5 --no-cpu-baseline --no-product-default --no-pixels-leg > gpurun_out/bench_${TAG}_10k.json 2> gpurun_out/bench_${TAG}_10k.log; echo bench10k rc=$?
python3 bench.py --steps 20 --warmup 5 --streams 2 --no-cpu-baseline --no-product-default --no-pixels-leg > gpurun_out/bench_${TAG}_two_streams.json 2> gpurun_out/bench_${TAG}_two_streams.log; echo two_streams rc=$?
python3 bench.py --model sdxl --steps 6 --warmup 2 > gpurun_out/bench_${TAG}_sdxl.json 2> gpurun_out/bench_${TAG}_sdxl.log; echo sdxl rc=$?
python3 bench.py --model dit --steps 10 --warmup 3 > gpurun_out/bench_${TAG}_dit.json 2> gpurun_out/bench_${TAG}_dit.log; echo dit rc=$?
python3 bench.py --model dit --fp8-attention --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_dit_fp8.json 2> gpurun_out/bench_${TAG}_dit_fp8.log; echo ditfp8 rc=$?
python3 bench.py --pixels-in --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_pixels_in.json 2> gpurun_out/bench_${TAG}_pixels_in.log; echo pixels rc=$?
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --fusion 0 --batch-pairs 32 --no-product-default > gpurun_out/bench_${TAG}_unfused.json 2> gpurun_out/bench_${TAG}_unfused.log; echo unfused rc=$?
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --dedup-cfg --no-product-default > gpurun_out/bench_${TAG}_dedup_cfg.json 2> gpurun_out/bench_${TAG}_dedup_cfg.log; echo dedup rc=$?
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-product-default --dump-launches gpurun_out/launches_${TAG}.jsonl > /dev/null 2>&1; echo launches rc=$?
python3 tools/sweep_batch.py > gpurun_out/batch_sweep_${TAG}.txt 2> gpurun_out/batch_sweep_${TAG}.log; echo sweep rc=$?
DSIM_DECODE_PROCS=auto python3 tools/files_in_bench.py > gpurun_out/bench_${TAG}_files_in.json 2> gpurun_out/bench_${TAG}_files_in.log; echo files_in rc=$?
tail -c 400 gpurun_out/bench_${TAG}.json
