#!/bin/bash
# Developer tool: reads per lane of the TABLE-form memo kernel on table-form workloads (dev build).
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
for r in ${RS:-1 2 4}; do
echo "== R=$r"
for a in "768 16" "1536 10" "384 16 2"; do FQTK_MEMO_R=$r python tools/bench_custom.py $a 2>/dev/null | grep "memo_kind=1"; done
for c in 5 3; do FQTK_MEMO_R=$r python bench.py --config $c --memo-table --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline']['kernel_ms'])"; done
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
