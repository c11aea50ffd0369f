#!/bin/bash
# Developer tool: builds the library with the memo-kernel ablation variants and times each on cfg3.
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip
for a in 0 16 1 2 4 8 3 7 15; do
  FQTK_MEMO_ABLATE=$a python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --reads 200000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ablate=$a', d['value'], d['roofline']['kernel_ms'])" || echo "ablate=$a failed"
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
