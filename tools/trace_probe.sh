#!/bin/bash
# Development aid (GPU box): kernel-trace stats of a python probe.  tools/trace_probe.sh tools/vae_probe.py [args]
export TMPDIR=/tmp
R=$(pwd)
cd /tmp && rm -rf /tmp/tp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python3 $R/"$@" > /tmp/tp.log 2>&1
tail -2 /tmp/tp.log | cut -c1-300
f=$(find /tmp/tp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
