#!/usr/bin/env python3
"""profiles/<tag>_launches.txt from the per-launch dump of one profiled step:
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --dump-launches gpurun_out/launches_<tag>.jsonl
  python3 profiles/launch_table.py <tag> gpurun_out/launches_<tag>.jsonl [rows]
One line per (kernel family, shape): launches, summed ms, average us, TFLOP/s and TB/s of the ALGORITHMIC flops / bytes."""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tag, path = sys.argv[1], sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 36
    acc = collections.OrderedDict()
    for line in open(path):
        r = json.loads(line)
        e = acc.setdefault((r["kernel"], r["shape"]), [0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += r["ms"]
        e[2] += r["gflop"]
        e[3] += r["mb"]
    total = sum(e[1] for e in acc.values())
    out = [f"per-shape launches of one profiled step ({len(acc)} shapes, {sum(e[0] for e in acc.values())} launches, "
           f"{total:.2f} ms of kernel time; the {rows} largest):"]
    for (k, shape), (n, ms, gf, mb) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:rows]:
        out.append(f"{k:34s} {shape:30s} n={n:2d} ms={ms:7.3f} avg={ms / n * 1e3:8.1f}us TF/s={gf / ms:7.1f} TB/s={mb / ms / 1e3:5.2f}")
    open(os.path.join(HERE, f"{tag}_launches.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
