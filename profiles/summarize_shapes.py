#!/usr/bin/env python3
"""Per-SHAPE fabric traffic of the GEMM / attention families (the per-kernel summaries of summarize.py are family averages).

  python profiles/summarize_shapes.py <tag> <launches.jsonl> <fetch pass dir> <write pass dir>

The --pmc passes run `bench.py --steps 1 --warmup 0 --no-profile`, i.e. the same deterministic launch sequence several times
(the untimed first step + one timed step); `bench.py --dump-launches` of the same build lists the launches of ONE step in
order with their shapes.  Dispatch i of kernel symbol S in a pass is therefore launch (i mod n_S) of S's per-step list.
Writes profiles/<tag>_pmc_shapes.json: per kernel symbol and shape -- launches per step, HBM-side bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md section HBM), the algorithmic bytes of the
launch record, their ratio, and the HIP-event duration of the launch record.
"""
import collections
import csv
import glob
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from summarize import demangle, short  # noqa: E402


def per_dispatch(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rr = list(csv.DictReader(open(f)))
    dm = demangle(sorted({r["Kernel_Name"] for r in rr}))
    seq = collections.defaultdict(dict)          # symbol -> dispatch id -> summed counter value
    for r in rr:
        s = seq[short(dm[r["Kernel_Name"]])]
        k = int(r["Dispatch_Id"])
        s[k] = s.get(k, 0.0) + float(r["Counter_Value"])
    return {k: [v[i] for i in sorted(v)] for k, v in seq.items()}


def main():
    tag, launches, fetch, write = sys.argv[1:5]
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(HERE, "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_step = collections.defaultdict(list)      # symbol -> [(shape, algorithmic MB, ms)] in launch order
    for ln in open(launches):
        r = json.loads(ln)
        if r["kernel"].startswith(("gemm_", "attention_", "ff_", "ln_linear")):
            per_step[bench.rocprof_name(r["kernel"])].append((r["shape"], r["mb"], r["ms"]))
    fe, wr = per_dispatch(fetch), per_dispatch(write)
    out = {}
    for sym, lst in per_step.items():
        if sym not in fe or sym not in wr:
            continue
        n = len(lst)
        if len(fe[sym]) % n or len(wr[sym]) % n:
            out[sym] = {"error": "dispatch count %d / %d is not a multiple of the %d launches per step" % (len(fe[sym]), len(wr[sym]), n)}
            continue
        acc = collections.OrderedDict()
        for i, (shape, mb, ms) in enumerate(lst):
            e = acc.setdefault(shape, {"launches_per_step": 0, "hbm_bytes": 0.0, "samples": 0, "algorithmic_mb_per_launch": mb, "ms": 0.0})
            e["launches_per_step"] += 1
            e["ms"] += ms
            for rep in range(len(fe[sym]) // n):
                e["hbm_bytes"] += (2.0 * fe[sym][rep * n + i] + wr[sym][rep * n + i]) * 1024.0
                e["samples"] += 1
        out[sym] = {}
        for shape, e in acc.items():
            b = e["hbm_bytes"] / e["samples"]
            out[sym][shape] = {"launches_per_step": e["launches_per_step"], "hbm_bytes_per_launch": int(b),
                               "algorithmic_bytes_per_launch": int(e["algorithmic_mb_per_launch"] * 1e6),
                               "traffic_over_algorithmic": round(b / (e["algorithmic_mb_per_launch"] * 1e6), 2) if e["algorithmic_mb_per_launch"] else None,
                               "avg_launch_ms": round(e["ms"] / e["launches_per_step"], 4)}
    json.dump(out, open(os.path.join(HERE, f"{tag}_pmc_shapes.json"), "w"), indent=1)
    print("wrote", f"{tag}_pmc_shapes.json", sum(len(v) for v in out.values()), "shapes")


if __name__ == "__main__":
    main()
