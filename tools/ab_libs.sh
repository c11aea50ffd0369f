#!/bin/bash
# Developer tool: A/B prebuilt matcher libraries (tools/_ab/*.so) on the bench configs, interleaved.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for rep in 1 2; do for c in ${CONFIGS:-3 2 4}; do for v in ${VARIANTS:-old new}; do
cp tools/_ab/$v.so fqtk_amd/lib/libfqtk_match.so
python bench.py --config $c --steps ${STEPS:-5} --warmup 1 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c $v', d['value'], d['roofline']['kernel_ms'])"
done; done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
