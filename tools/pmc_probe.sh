#!/bin/bash
# Development aid: SQ counter passes over tools/attn_probe.py (run on the GPU box): tools/pmc_probe.sh <kernel substring> [probe args]
cd /tmp && export TMPDIR=/tmp
KN=$1; shift
rm -rf /tmp/p1 /tmp/p2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/p1 -- python3 /root/repo/tools/${PROBE:-attn_probe.py} "$@" > /tmp/o1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p2 -- python3 /root/repo/tools/${PROBE:-attn_probe.py} "$@" > /tmp/o2.log 2>&1
python3 - "$KN" <<'PY'
import csv, glob, collections, sys
kn = sys.argv[1]
for d in ("/tmp/p1", "/tmp/p2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no csv", d); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if kn in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in acc.items():
        print(f"{k:32s} {v / n[k]:16.0f}  launches={n[k]}")
PY
