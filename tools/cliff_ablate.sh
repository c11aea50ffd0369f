#!/bin/bash
# Developer study: where the first pass loses time when reads carry IUPAC bytes (cfg 3).  Builds a variant of the library with
# FQTK_EXTRA_DEFS, runs the 0 / 1 % / 10 % rows without the parity gate, restores the product build.
cd "$(dirname "$0")/.."
row() { FQTK_SYNTH_PIUPAC=$2 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --parity none --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'row': '$1', 'G_reads_s': round(d['value']/1000,1), 'kernel_ms': r['kernel_ms'], 'ms_per_step': d['ms_per_step']}))"; }
for v in "$@"; do
    echo "=== variant: [$v]"
    FQTK_EXTRA_DEFS="$v" python -m fqtk_amd.build >/dev/null 2>&1 || { echo build failed; continue; }
    row "none" 0; row "1% IUPAC" 0.00063; row "10% IUPAC" 0.0066
done
python -m fqtk_amd.build >/dev/null 2>&1
