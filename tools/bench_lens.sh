#!/bin/bash
# Variable-length batches (obs_len, '+B' read structures): the LENS instantiations of the memo kernels, one line per config.
cd "$(dirname "$0")/.."
for c in 3 2 5 4; do for mode in "--lens" ""; do
python bench.py --config $c --steps ${STEPS:-5} --warmup 1 --cpu-seconds 0 --no-scopes --parity windows $mode >/dev/null 2>&1 && python -c "
import json
d=json.load(open('gpurun_out/bench_detail.json')); r=d['roofline']
print(json.dumps({'config': d['config']['workload'][:5].strip(), 'obs_len': d['config']['obs_len'], 'memo_kind': d['config'].get('memo_kind'), 'G_reads_s': round(d['value']/1000,1), 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch'], 'kernel': r['kernel'], 'parity': d['config']['parity']}))"
done; done
