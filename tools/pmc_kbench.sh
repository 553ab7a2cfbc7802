#!/bin/bash
# Development aid: SQ counter passes over one kbench shape (run on the GPU box from the repo root):
#   tools/pmc_kbench.sh <kernel substring> <kbench args...>     e.g.  tools/pmc_kbench.sh pair_tail160 256 5 tail_256_d160
# Two --pmc passes (8 SQ slots each); prints per-launch averages of every counter and the derived MFMA-busy / wave-cycle split.
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
KN=$1; shift
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/p1 -- $ROOT/tools/${KBENCH:-kbench} "$@" > /tmp/o1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p2 -- $ROOT/tools/${KBENCH:-kbench} "$@" > /tmp/o2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d /tmp/p3 -- $ROOT/tools/${KBENCH:-kbench} "$@" > /tmp/o3.log 2>&1
python3 - "$KN" <<'PY'
import csv, glob, collections, sys
kn = sys.argv[1]
tot = {}
for d in ("/tmp/p1", "/tmp/p2", "/tmp/p3"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no csv", d); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if kn in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in acc.items():
        tot[k] = v / n[k]
        print(f"{k:32s} {v / n[k]:16.0f}  launches={n[k]}")
if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
    cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
    wc = tot.get("SQ_WAVE_CYCLES", 0)
    print(f"cycles per launch {cyc:.0f}   MFMA-busy {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.4f}")
    if wc:
        print(f"wave cycles: issue-stall {tot['SQ_WAIT_INST_ANY'] / wc:.3f}  waitcnt/barrier {tot['SQ_WAIT_ANY'] / wc:.3f}  active {tot.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}"
              f"   VALU-busy / MFMA-busy {tot['SQ_ACTIVE_INST_VALU'] * 4 / tot['SQ_VALU_MFMA_BUSY_CYCLES']:.3f}")
PY
