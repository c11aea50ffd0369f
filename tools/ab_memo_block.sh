#!/bin/bash
# Developer tool: workgroup size / hot-table budget of the table-form memo kernel.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for v in ${VARIANTS:-"256 16384" "512 32768" "1024 65536"}; do set -- $v
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_MEMO_BLOCK=$1 -DFQTK_HOT_BYTES=$2 -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || continue
echo "== block $1 hot $2"
for a in "768 16" "1536 10" "1536 8" "384 16 2"; do python tools/bench_custom.py $a 2>/dev/null | grep "memo_kind=1"; done
for c in 5 3; do python bench.py --config $c --memo-table --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline']['kernel_ms'])"; done
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
