#!/bin/bash
# Developer tool: A/B the reads-per-lane factor R of the LDS-resident memo kernel (dev build).
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE ${EXTRA} -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
for c in ${CONFIGS:-3 2 4}; do for r in ${RS:-1 2 4 8}; do
FQTK_MEMO_R=$r python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c R=$r', d['value'], d['roofline']['kernel_ms'])"
done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
