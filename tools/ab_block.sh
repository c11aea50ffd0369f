#!/bin/bash
# Developer tool: workgroup size of the LDS memo kernel (small tables can afford more, smaller workgroups).
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for blk in ${BLOCKS:-1024 512 256}; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -DFQTK_LDS_BLOCK=$blk -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || continue
for c in ${CONFIGS:-2 4 3}; do for r in ${RS:-4 2}; do
FQTK_MEMO_R=$r python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c block=$blk R=$r', d['value'], d['roofline']['kernel_ms'])"
done; done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
