#!/bin/bash
# Developer tool: A/B non-temporal streaming loads/stores (default) against plain ones (-DFQTK_NO_NT).
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
run() { python bench.py --config $1 --steps 5 --warmup 1 --cpu-seconds 0 $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$2', d['value'], d['roofline']['kernel_ms'], d['config']['parity'])" || echo "$2 failed"; }
for flag in "" "${FLAG:--DFQTK_NO_NT}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $flag -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
  for c in ${CONFIGS:-3 2 5}; do run $c "nt[$flag] cfg$c"; done
  run 3 "nt[$flag] cfg3 scan" --no-cache
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
