#!/bin/bash
# Developer tool: sensitivity of the LDS memo kernel to the hash salt (= placement of the hot entries).
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
for c in ${CONFIGS:-3 2 4}; do for s in ${SALTS:-0 1 2 3 4 5 6 7}; do
FQTK_LDSM_TRIALS=1 FQTK_LDSM_SALT=$s python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/tmp/err.txt | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c salt=$s', d['value'], d['roofline']['kernel_ms'])"; grep ldsm /tmp/err.txt | tail -1
done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
