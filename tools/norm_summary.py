"""Development aid: per-shape time / bandwidth of the norm launches in a `bench.py --dump-launches` file."""
import collections
import json
import sys

agg = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    if "norm" not in r["kernel"]:
        continue
    a = agg.setdefault((r["kernel"], r["shape"]), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += r["ms"]
    a[2] += r["mb"]
for (k, sh), (c, ms, mb) in agg.items():
    print(f"{k:18s} {sh:28s} n={c:3d} ms={ms:7.3f} avg_us={ms / c * 1e3:7.1f} TB/s={mb / ms / 1e3:6.2f}")
