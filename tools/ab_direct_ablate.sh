#!/bin/bash
# Developer tool: ablations of the direct-indexed table form on cfg 5 (dev build).
#   1 = no global probes at all, 4 = no histogram, 8 = no result store, 16 = no LDS cache, 256 = no cuckoo table (N reads)
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
run() { python bench.py --config 5 --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || echo "$1 failed"; }
for a in 0 1 4 16 256 17 272 2 8; do FQTK_MEMO_ABLATE=$a run "cfg5 direct (R=1, pipelined) ablate=$a"; done
FQTK_MEMO_NOPF=1 run "cfg5 direct R=1 unpipelined"
for r in 2 4; do FQTK_MEMO_R=$r FQTK_MEMO_NOPF=1 run "cfg5 direct R=$r unpipelined"; done
FQTK_MEMO_R=2 run "cfg5 direct R=2 pipelined (spills)"
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
