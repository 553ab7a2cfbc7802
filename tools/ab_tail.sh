#!/bin/bash
# Development aid: the score tail of the default tap in alternating kbench binaries on one box.
#   gpurun -- ./tools/ab_tail.sh "<bin> <bin> ..." <rounds>
BINS=$1; R=${2:-3}
for r in $(seq 1 $R); do
  for b in $BINS; do
    echo "== $b round $r"
    KB_ROUNDS=5 timeout 120 ./tools/$b 256 20 tail_256_d160 2>&1 | grep "^tail_256_d160 .*cosine"
  done
done
