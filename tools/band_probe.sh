#!/bin/bash
# Development aid (GPU box): fabric traffic (FETCH_SIZE, gfx950 x2 correction) and duration of one kbench GEMM shape per
# tile-order band width gn (KB_ONEEXP = gn << 16; 255 = the row-major order).   tools/band_probe.sh <kbench filter> <B2> gn...
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
F=$1; B2=$2; shift 2
for gn in "$@"; do
  export KB_ONEEXP=$((gn << 16))
  rm -rf /tmp/bp_$gn
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/bp_$gn -- $R/tools/kbench $B2 3 $F > /tmp/bp_$gn.log 2>&1
  python3 - $gn <<'PY'
import csv, glob, sys, collections
gn = sys.argv[1]
fs = glob.glob(f"/tmp/bp_{gn}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_kernel" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][-40:], r["Grid_Size"])].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print(f"gn={gn:>3s} {k[0]} grid={k[1]:>7s} n={len(v)} fetch={2 * sum(x[0] for x in v) / len(v) / 1e6:8.3f} GB(x2 corrected; counter in KB) us={sum(x[1] for x in v) / len(v):8.1f}")
PY
done
