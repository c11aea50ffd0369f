#!/bin/bash
# The presence filter of the hash-table memo form on / off (FQTK_MEMO_NO_FILTER=1: single-choice hot table, no filter), one box:
# cfg 3 / 2 / 4 with the table form pinned, and 12+12 / 16+16 dual indices.   usage: tools/ab_filter.sh   (on the GPU box)
cd "$(dirname "$0")/.."
for c in 3 2 4; do for f in "" 1; do
FQTK_MEMO_NO_FILTER=$f python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --no-scopes --parity windows --memo-table >/dev/null 2>&1 && python -c "
import json
d=json.load(open('gpurun_out/bench_detail.json')); r=d['roofline']
print(json.dumps({'config': d['config']['workload'][:5].strip(), 'filter': 'off' if '$f' else 'on', 'G_reads_s': round(d['value']/1000,1), 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'parity': d['config']['parity'][:40]}))"
done; done
for a in "384 24 1 2" "384 32 1 2"; do for f in "" 1; do echo "filter $([ -n "$f" ] && echo off || echo on): $(FQTK_MEMO_NO_FILTER=$f timeout 300 python tools/bench_custom.py $a 2>/dev/null | tail -1)"; done; done
