#!/bin/bash
# All five configs x {default memo form, table form pinned, scan}: one line each (results table of DESIGN.md section 7).
cd "$(dirname "$0")/.."
for c in 3 2 4 5 1; do for mode in "" "--memo-table" "--no-cache"; do
python bench.py --config $c --steps ${STEPS:-50} --warmup ${WARMUP:-10} --cpu-seconds ${CPU_SECONDS:-0} --no-scopes --parity windows $mode >/dev/null 2>&1 && python -c "
import json
d=json.load(open('gpurun_out/bench_detail.json')); r=d['roofline']   # (the whole record of the run; the printed line is its short form)
row={'config': d['config']['workload'][:5].strip(), 'mode': '$mode' or 'default', 'memo_kind': d['config'].get('memo_kind'), 'G_reads_s': round(d['value']/1000,1), 'GBps': r['achieved'], 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'kernel': r['kernel']}
if row['config'] == 'cfg1': row['note'] = 'launch-bound: the config is ONE launch of 1 M reads (12 MB), %.0f us -- its fraction says nothing about the kernel (the same kernel on cfg 2: 100 M reads)' % (r['kernel_ms'] * 1e3)
print(json.dumps(row))"
done; done
