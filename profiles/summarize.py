#!/usr/bin/env python3
"""Turn the rocprofv3 output that a gpurun call merged into gpurun_out/ into the small, tracked
summaries under profiles/ (gpurun_out/ is scratch).

  python profiles/summarize.py r01 gpurun_out/prof_r1_trace gpurun_out/prof_r1_fetch gpurun_out/prof_r1_write [gpurun_out/prof_r1_mfma]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, one row per kernel) and
profiles/<tag>_pmc_hbm.json (per kernel: launches, FETCH_SIZE and WRITE_SIZE per launch in KB as the
counters report them, and HBM bytes per launch with the gfx950 correction of
MI355X_MICROARCH.md section HBM: FETCH_SIZE counts half the bytes of wide coalesced reads -> x2).
With the optional fifth argument (a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass) it also writes
profiles/<tag>_pmc_mfma.json: per kernel, MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), with
cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), i.e. at the clock the kernel actually ran at.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True,
                             text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(n):
    """llvm-cxxfilt of this ROCm does not know the DF16b (__bf16) mangling: prettify by hand."""
    m = re.match(r"_ZN4dsim12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)EEv", n)
    if m:
        args = []
        for tok in re.findall(r"DF16b|f|Li\d+E|Lb[01]E", m.group(2)):
            if tok == "DF16b":
                args.append("__bf16")
            elif tok == "f":
                args.append("float")
            elif tok.startswith("Li"):
                args.append(tok[2:-1])
            else:
                args.append("true" if tok[2] == "1" else "false")
        return f"{m.group(1)}<{', '.join(args)}>"
    n = n.replace("dsim::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)


def write_stats(tag, trace):
    stats = glob.glob(os.path.join(trace, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    dm = demangle([r["Name"] for r in rows])
    with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "percent", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(dm[r["Name"]]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])


def write_mfma(outname, d):
    """profiles/<outname>.json: per kernel MFMA utilisation and held clock from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass"""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rr = list(csv.DictReader(open(f)))
    dm3 = demangle(sorted({r["Kernel_Name"] for r in rr}))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rr:
        e = acc[short(dm3[r["Kernel_Name"]])]
        e[r["Counter_Name"]] += float(r["Counter_Value"])
        e["n_" + r["Counter_Name"]] += 1
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp") and r.get("Start_Timestamp"):
            e["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    mf = {}
    for k, e in acc.items():
        n = max(e["n_GRBM_GUI_ACTIVE"], 1)
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0 / n
        busy = e["SQ_VALU_MFMA_BUSY_CYCLES"] / max(e["n_SQ_VALU_MFMA_BUSY_CYCLES"], 1)
        mf[k] = {"launches": int(n), "cycles_per_launch": int(cyc), "mfma_busy_cycles_per_launch": int(busy),
                 "mfma_util": round(busy / (cyc * 1024.0), 4) if cyc else None}
        if e.get("ns"):
            # clock the chip held during this kernel IN THIS (profiled, serialised) pass: GRBM_GUI_ACTIVE / 8 XCDs over the
            # dispatch's own begin/end timestamps (reads high on dispatches shorter than ~0.3 ms: MI355X_MICROARCH.md DVFS)
            mf[k]["held_clock_ghz"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / e["ns"], 3)
            mf[k]["avg_launch_us"] = round(e["ns"] / n / 1e3, 1)
    json.dump(mf, open(os.path.join(HERE, f"{outname}.json"), "w"), indent=1, sort_keys=True)


def main():
    if sys.argv[1] == "--mfma-only":           # python profiles/summarize.py --mfma-only <output name> <mfma pass dir>
        write_mfma(sys.argv[2], sys.argv[3])
        print("wrote", sys.argv[2])
        return
    if sys.argv[1] == "--stats-only":          # python profiles/summarize.py --stats-only <tag> <trace dir>
        write_stats(sys.argv[2], sys.argv[3])
        print("wrote", sys.argv[2])
        return
    tag, trace, fetch, write = sys.argv[1:5]
    write_stats(tag, trace)
    pmc = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
    for d, key in ((fetch, "FETCH_SIZE_KB"), (write, "WRITE_SIZE_KB")):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        rr = list(csv.DictReader(open(f)))
        dm2 = demangle(sorted({r["Kernel_Name"] for r in rr}))
        for r in rr:
            e = pmc[short(dm2[r["Kernel_Name"]])]
            e[key] += float(r["Counter_Value"])
            if key == "FETCH_SIZE_KB":
                e["launches"] += 1
    out = {}
    for k, e in pmc.items():
        n = max(e["launches"], 1)
        fk, wk = e["FETCH_SIZE_KB"] / n, e["WRITE_SIZE_KB"] / n
        out[k] = {"launches": e["launches"], "fetch_size_kb_per_launch": round(fk, 1),
                  "write_size_kb_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
    json.dump(out, open(os.path.join(HERE, f"{tag}_pmc_hbm.json"), "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 5:
        write_mfma(f"{tag}_pmc_mfma", sys.argv[5])
    print("wrote", tag)


if __name__ == "__main__":
    main()
