#!/bin/bash
# Turn what `tools/profile_round.sh <tag>` left under gpurun_out/ (merged back by gpurun) into the tracked summaries under profiles/.
#   profiles/collect_round.sh r04b
T=$1
G=gpurun_out
python3 profiles/summarize.py $T $G/prof_${T}_trace $G/prof_${T}_fetch $G/prof_${T}_write $G/prof_${T}_mfma
python3 profiles/summarize.py ${T}_sdxl $G/prof_${T}_trace_sdxl $G/prof_${T}_sdxl_fetch $G/prof_${T}_sdxl_write $G/prof_${T}_sdxl_mfma
python3 profiles/summarize.py ${T}_dit $G/prof_${T}_trace_dit $G/prof_${T}_dit_fetch $G/prof_${T}_dit_write $G/prof_${T}_dit_mfma
python3 profiles/summarize.py --stats-only ${T}_pixels_in $G/prof_${T}_trace_pixels
python3 profiles/summarize.py --mfma-only ${T}_pmc_mfma_sustained $G/prof_${T}_mfma40          # 40 back-to-back steps: MFMA busy + held clock
python3 profiles/summarize_shapes.py $T $G/launches_${T}.jsonl $G/prof_${T}_fetch $G/prof_${T}_write
python3 profiles/summarize_tap.py $T $G/prof_${T}_tap1 $G/prof_${T}_tap2
python3 profiles/launch_table.py $T $G/launches_${T}.jsonl 40
for f in "" _10k _two_streams _sdxl _dit _dit_fp8 _pixels_in _unfused _dedup_cfg _files_in; do
  [ -s $G/bench_${T}$f.json ] && tail -n 1 $G/bench_${T}$f.json | python3 -c "import json,sys; json.dump(json.loads(sys.stdin.read()), open('profiles/${T}_bench$f.json','w'), indent=1)"
done
cp $G/batch_sweep_${T}.txt profiles/${T}_batch_sweep.txt
ls profiles/${T}_*
