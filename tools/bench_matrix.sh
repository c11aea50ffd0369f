#!/bin/bash
# All five configs x {default memo form, table form pinned, scan}: one line each (results table of DESIGN.md section 7).
cd "$(dirname "$0")/.."
for c in 3 2 4 5 1; do for mode in "" "--memo-table" "--no-cache"; do
python bench.py --config $c --steps ${STEPS:-5} --warmup 1 --cpu-seconds ${CPU_SECONDS:-0} --no-scopes --parity windows $mode 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'config': d['config']['workload'][:5].strip(), 'mode': '$mode' or 'default', 'memo_kind': d['config'].get('memo_kind'), 'G_reads_s': round(d['value']/1000,1), 'GBps': r['achieved'], 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'kernel': r['kernel']}))"
done; done
