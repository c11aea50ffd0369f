#!/bin/bash
# Developer tool: the table-form kernels under two prebuilt libraries on ONE box: fqtk_amd/lib/old/libfqtk_match.so
# (build it from the commit to compare against: git archive <commit> fqtk_amd/csrc include | tar -x -C /tmp/oldsrc; hipcc ...)
# and the product build.  cfg 5 takes the direct form by default; cfg 3 / 2 / 4 with the hash-table form pinned.
cd "$(dirname "$0")/.."
line() { grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1', round(d['value']/1000,1), 'G reads/s', r['kernel_ms'], 'ms', r['frac'])" || { echo "$1 failed"; tail -3 /tmp/ab_err.txt; }; }
run() { python bench.py --config $CFG $MODE --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/tmp/ab_err.txt | line "cfg$CFG $MODE $1"; }
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for rep in 1 2; do for lib in old prod; do
if [ $lib = old ]; then cp fqtk_amd/lib/old/libfqtk_match.so fqtk_amd/lib/libfqtk_match.so; else cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so; fi
CFG=5 MODE="" run $lib
for c in 3 2 4; do CFG=$c MODE="--memo-table" run $lib; done
if [ $rep = 1 ]; then CFG=5 MODE="--lens" run $lib; CFG=3 MODE="--memo-table --lens" run $lib; fi
done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
