#!/bin/bash
# Same box, alternating runs: the bench's kernel scope (cfg 3, no scopes, no parity pass) with several builds of libfqtk_match.so.
# usage: tools/ab_match_libs.sh <reps> <libdir or ""> ...     ("" = the product library; a libdir under fqtk_amd/lib holds a libfqtk_match.so)
REPS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp fqtk_amd/lib/libfqtk_match.so /tmp/product_libfqtk_match.so
for rep in $(seq 1 $REPS); do
  for L in "$@"; do
    if [ -n "$L" ]; then cp fqtk_amd/lib/$L/libfqtk_match.so fqtk_amd/lib/libfqtk_match.so; else cp /tmp/product_libfqtk_match.so fqtk_amd/lib/libfqtk_match.so; fi
    python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --no-verify --no-scopes ${BENCH_ARGS} > /tmp/ab_line.json 2>/dev/null
    python -c "
import json; d=json.load(open('/tmp/ab_line.json')); print('lib=${L:-product}', 'G_reads_s', round(d['value']/1000,1), 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done
cp /tmp/product_libfqtk_match.so fqtk_amd/lib/libfqtk_match.so
