#!/bin/bash
# Developer tool: builds the library with the memo-kernel ablation variants and times each on cfg3.
#   1 = skip global probes, 2 = skip LDS code lookups, 4 = skip histogram, 8 = skip result store, 16 = skip LDS hot table
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
run() { python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes --reads 200000000 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'])" || echo "$1 failed"; }
for a in ${ABLATIONS:-0 1 4 16 3 7}; do FQTK_MEMO_R=${R:-1} FQTK_MEMO_ABLATE=$a run "memo R=${R:-1} ablate=$a"; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
