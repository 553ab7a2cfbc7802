#!/bin/bash
# Development aid: alternate kbench binaries on the same box, R rounds, and print the median time per shape and binary.
#   gpurun -- ./tools/ab_kbench.sh "<bin> <bin> ..." <rounds> <B2> <iters> <filter> [<filter> ...]     (binaries under tools/)
BINS=$1; R=$2; B2=$3; IT=$4; shift 4
OUT=/tmp/ab_kbench.$$
: > $OUT
for f in "$@"; do
  for r in $(seq 1 $R); do
    for b in $BINS; do
      ./tools/$b $B2 $IT $f | grep "^$f " | sed "s/^/$b /" >> $OUT
    done
  done
done
python3 - $OUT "$BINS" <<'PY'
import re, sys, collections
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+) .*auto (\S+)\s+([\d.]+) ms\s+([\d.]+) TF st=(-?\d+) sum=(\w+)", l)
    if not m: continue
    e = d.setdefault(m.group(2), collections.OrderedDict()).setdefault(m.group(1), {"ms": [], "sum": set(), "tile": m.group(3)})
    e["ms"].append(float(m.group(4))); e["sum"].add(m.group(7))
bins = sys.argv[2].split()
for shape, v in d.items():
    base = sorted(v[bins[0]]["ms"])[len(v[bins[0]]["ms"]) // 2]
    cells = []
    for b in bins:
        if b not in v: continue
        ms = sorted(v[b]["ms"]); med = ms[len(ms) // 2]
        same = v[b]["sum"] == v[bins[0]]["sum"] and len(v[b]["sum"]) == 1
        cells.append("%s %.4f (%+.1f%%)%s" % (b.replace("kbench", "kb"), med, 100 * (base / med - 1), "" if same else " SUM-DIFFERS"))
    print("%-26s %s  %s" % (shape, v[bins[0]]["tile"], " | ".join(cells)))
PY
