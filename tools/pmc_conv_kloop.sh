#!/bin/bash
# Evidence task (round 6): per-stall-reason SQ / TCP / TCC counters of the two top 3x3-conv shapes of the headline step, gemm_kernel<bf16,256,320,2,...>
# (run on the GPU box from the repo root): tools/pmc_conv_kloop.sh <out file>
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/r06_conv_kloop_pmc.txt}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /tmp/counters.txt 2>&1
echo "== counters offered on this gfx950 box that the task names (grep of rocprofv3 -L)" > $OUT
for c in SQ_INSTS_VALU_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum; do
  if grep -q "$c" /tmp/counters.txt; then echo "  offered  $c" >> $OUT; else echo "  MISSING  $c" >> $OUT; fi
done
run() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pk_$name
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pk_$name -- $ROOT/tools/kbench 256 3 $SHAPE > /tmp/pk_$name.log 2>&1
  python3 - "$name" "$KSUB" >> $OUT <<'PY'
import csv, glob, collections, sys
name, kn = sys.argv[1], sys.argv[2]
fs = glob.glob(f"/tmp/pk_{name}/**/*counter_collection.csv", recursive=True)
if not fs:
    print(f"  pass {name}: no csv (counter set rejected?)"); print(open(f"/tmp/pk_{name}.log").read()[-600:]); sys.exit(0)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if kn in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
if not acc:
    names = sorted({r["Kernel_Name"][:110] for r in csv.DictReader(open(fs[0]))})
    print(f"  pass {name}: no kernel matching '{kn}' among {len(names)} kernels:", names[:6])
for k, v in sorted(acc.items()):
    print(f"  {k:36s} {v / n[k]:18.0f}  per launch (launches={n[k]})")
PY
}
for SHAPE in conv3_64_320_320 conv3_16_2560_1280; do
  KSUB="Li256ELi320ELi2ELb0ELi4ELi2ELi0E"
  echo "== $SHAPE  (kernel gemm_kernel<__bf16, 256, 320, 2, false, 4, 2, 0>; kbench 256 elements = 64 pairs)" >> $OUT
  run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
  run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
  run c SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
  run d SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH GRBM_GUI_ACTIVE
  run e TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  run f TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
grep -c . /tmp/counters.txt >> /dev/null
cat $OUT
