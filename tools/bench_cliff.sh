#!/bin/bash
# Non-canonical reads on cfg 3: throughput with '.' no-calls (served by the memo under N's
# key) and with IUPAC bytes in the READ (each such read takes a wave-cooperative scan of all samples).
# Rates are per READ; the generator's knobs are per base (16 bases).
cd "$(dirname "$0")/.."
row() { FQTK_SYNTH_PDOT=$2 FQTK_SYNTH_PIUPAC=$3 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --parity windows --no-scopes $4 >/dev/null 2>&1 && python -c "
import json
d=json.load(open('gpurun_out/bench_detail.json')); r=d['roofline']
print(json.dumps({'row': '$1', 'G_reads_s': round(d['value']/1000,1), 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'kernel': r['kernel'], 'parity': d['config']['parity']}))"; }
row "cfg3, no non-canonical reads" 0 0
row "cfg3, 1% of reads carry a dot" 0.00063 0
row "cfg3, 10% of reads carry a dot" 0.0066 0
row "cfg3, 0.1% of reads carry an IUPAC byte" 0 0.0000625
row "cfg3, 1% of reads carry an IUPAC byte" 0 0.00063
row "cfg3, 10% of reads carry an IUPAC byte" 0 0.0066
row "cfg3 table form, no non-canonical reads" 0 0 --memo-table
row "cfg3 table form, 1% dot" 0.00063 0 --memo-table
row "cfg3 table form, 1% IUPAC" 0 0.00063 --memo-table
row "cfg3 table form, 10% IUPAC" 0 0.0066 --memo-table
