#!/usr/bin/env python3
"""profiles/<tag>_tap_pmc.json from the counter csv files of tools/tap_probe.py's --pmc passes:
  python profiles/summarize_tap.py r02 gpurun_out/prof_r02_tap1 gpurun_out/prof_r02_tap2
Per kernel (the tapped q/k/v GEMM, pair_tail_kernel): MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x
1024 SIMDs), and the wave-cycle split issue-stall / waitcnt-or-barrier / active (SQ_WAIT_INST_ANY, SQ_WAIT_ANY,
SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES), VALU-busy and MFMA/VALU co-execution."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import demangle, short          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rr = list(csv.DictReader(open(f)))
            dm = demangle(sorted({r["Kernel_Name"] for r in rr}))
            for r in rr:
                k = short(dm[r["Kernel_Name"]])
                if not (k.startswith("gemm_kernel") or k.startswith("pair_tail")):
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                acc[k]["n_" + r["Counter_Name"]] += 1
    out = {}
    for k, e in acc.items():
        def avg(c):
            return e[c] / max(e["n_" + c], 1)
        cyc = avg("GRBM_GUI_ACTIVE") / 8.0
        wc = avg("SQ_WAVE_CYCLES")
        out[k] = {"launches": int(e["n_GRBM_GUI_ACTIVE"]), "cycles_per_launch": int(cyc),
                  "mfma_util": round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0), 4) if cyc else None,
                  "wave_cycles_issue_stall_frac": round(avg("SQ_WAIT_INST_ANY") / wc, 4) if wc else None,
                  "wave_cycles_waitcnt_barrier_frac": round(avg("SQ_WAIT_ANY") / wc, 4) if wc else None,
                  "wave_cycles_active_frac": round(avg("SQ_ACTIVE_INST_ANY") / wc, 4) if wc else None,
                  "valu_busy_over_mfma_busy": round(avg("SQ_ACTIVE_INST_VALU") * 4 / max(avg("SQ_VALU_MFMA_BUSY_CYCLES"), 1), 4),
                  "mfma_valu_coexec_cycles_per_launch": int(avg("SQ_VALU_MFMA_COEXEC_CYCLES"))}
    json.dump(out, open(os.path.join(HERE, f"{tag}_tap_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
