#!/bin/bash
# Run on the GPU box (gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/profile_round.sh <tag>      e.g. r01c  -> gpurun_out/prof_<tag>_{trace,fetch,write,mfma}, gpurun_out/bench_<tag>.json
TAG=$1
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_${TAG}_under_rocprof.json 2> $R/gpurun_out/rocprof_${TAG}_trace.log
echo trace rc=$?
for c in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${c%%:*}; ctr=${c#*:}
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/prof_${TAG}_${name} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/rocprof_${TAG}_${name}.log
  echo $name rc=$?
  # keep only the counter csv (the merged-back directory is size-limited)
  find $R/gpurun_out/prof_${TAG}_${name} -type f ! -name "*counter_collection.csv" -delete
done
cd $R
python3 bench.py --steps 6 --warmup 2 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.log
echo bench rc=$?
tail -c 600 gpurun_out/bench_${TAG}.json
