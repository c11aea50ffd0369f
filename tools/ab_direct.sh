#!/bin/bash
# Developer tool: table-form memo with and without the direct-indexed variant (dev build: FQTK_NO_DIRECT=1
# keeps every entry in the cuckoo table, as in round 1), on cfg 5, cfg 2 (table pinned) and two plain plates.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
for nd in "" 1; do
echo "== FQTK_NO_DIRECT=${nd:-0}"
for c in 5 2; do FQTK_NO_DIRECT=$nd python bench.py --config $c --memo-table --steps 10 --warmup 2 --cpu-seconds 0 --parity windows --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['memo_kind'])"; done
for a in "1536 10" "1536 8" "96 8 2"; do FQTK_NO_DIRECT=$nd python tools/bench_custom.py $a 2>/dev/null | grep "memo_kind=1"; done
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
