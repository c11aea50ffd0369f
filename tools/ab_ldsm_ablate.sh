#!/bin/bash
# Developer tool: what bounds the LDS-form kernel -- the stream or the look-up?  Compile-time ablations of
# lds_memo_kernel (FQTK_LDSM_ABL: 1 no look-up, 2 no histogram, 4 no verification), cfg 3 and cfg 2, next to
# tools/hbm_stream.hip's figure for the bare stream of the same shape.  Results are WRONG by construction.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for abl in ${VARIANTS:-0 2 4 6 1 3}; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_LDSM_ABL=$abl -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || continue
for c in ${CONFIGS:-3 2}; do
FQTK_MEMO_ABLATE=1 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --parity none --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('abl=$abl cfg$c', round(d['value']/1000,1), 'G reads/s', d['roofline']['kernel_ms'], 'ms', d['roofline']['frac'])"
done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
