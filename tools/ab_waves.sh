#!/bin/bash
# Developer tool: A/B waves-per-EU (register caps) x reads-per-lane of the memo kernel.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for wv in ${WAVES:-8}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -DFQTK_MEMO_WAVES=$wv -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
  for c in ${CONFIGS:-3 2}; do for r in ${RS:-1 2}; do
    FQTK_MEMO_R=$r python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes --reads 200000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('waves=$wv cfg$c R=$r', d['value'], d['roofline']['kernel_ms'])"
  done; done
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
