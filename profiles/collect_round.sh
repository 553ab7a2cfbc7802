#!/bin/bash
# Turn what `tools/profile_round.sh <tag>` left under gpurun_out/ (merged back by gpurun) into the tracked summaries under profiles/.
#   profiles/collect_round.sh r04b
T=$1
G=gpurun_out
python3 profiles/summarize.py $T $G/prof_${T}_trace $G/prof_${T}_fetch $G/prof_${T}_write $G/prof_${T}_mfma
python3 profiles/summarize.py ${T}_sdxl $G/prof_${T}_trace_sdxl $G/prof_${T}_sdxl_fetch $G/prof_${T}_sdxl_write $G/prof_${T}_sdxl_mfma
python3 profiles/summarize.py ${T}_dit $G/prof_${T}_trace_dit $G/prof_${T}_dit_fetch $G/prof_${T}_dit_write $G/prof_${T}_dit_mfma
python3 profiles/summarize.py ${T}_pixels_in $G/prof_${T}_trace_pixels $G/prof_${T}_pixels_in_fetch $G/prof_${T}_pixels_in_write $G/prof_${T}_pixels_in_mfma
python3 profiles/summarize.py --mfma-only ${T}_pmc_mfma_sustained $G/prof_${T}_mfma40          # 40 back-to-back steps: MFMA busy + held clock
python3 profiles/summarize_shapes.py $T $G/launches_${T}.jsonl $G/prof_${T}_fetch $G/prof_${T}_write
python3 profiles/summarize_tap.py $T $G/prof_${T}_tap1 $G/prof_${T}_tap2
python3 profiles/launch_table.py $T $G/launches_${T}.jsonl 40
for f in "" _10k _two_streams _sdxl _dit _dit_fp8 _pixels_in _unfused _dedup_cfg _files_in; do
  [ -s $G/bench_${T}$f.json ] && tail -n 1 $G/bench_${T}$f.json | python3 -c "import json,sys; json.dump(json.loads(sys.stdin.read()), open('profiles/${T}_bench$f.json','w'), indent=1)"
done
# the lines were printed before this snapshot's PMC passes were summarised: point their PMC fields at the snapshot's own files
python3 - $T <<'PY'
import json, os, sys
tag = sys.argv[1]
for suffix, model in (("", ""), ("_10k", ""), ("_two_streams", ""), ("_unfused", ""), ("_dedup_cfg", ""), ("_pixels_in", "pixels_in_"), ("_sdxl", "sdxl_"), ("_dit", "dit_"), ("_dit_fp8", "dit_")):
    p = f"profiles/{tag}_bench{suffix}.json"
    hb, mf = f"profiles/{tag}_{model}pmc_hbm.json", f"profiles/{tag}_{model}pmc_mfma.json"
    if not (os.path.exists(p) and os.path.exists(hb) and os.path.exists(mf)):
        continue
    d, H, M = json.load(open(p)), json.load(open(hb)), json.load(open(mf))
    for e in [d.get("roofline", {})] + d.get("roofline_top_kernels", []):
        k = e.get("kernel")
        if k in M and "mfma_util_pmc" in e:
            e["mfma_util_pmc"] = M[k]["mfma_util"]
    r = d.get("roofline", {})
    if r.get("kernel") in H:
        r["traffic"] = H[r["kernel"]]["hbm_bytes_per_launch"]
        r["pmc_source"] = {"traffic": os.path.basename(hb), "mfma_util_pmc": os.path.basename(mf)}
    json.dump(d, open(p, "w"), indent=1)
PY
cp $G/batch_sweep_${T}.txt profiles/${T}_batch_sweep.txt
ls profiles/${T}_*
