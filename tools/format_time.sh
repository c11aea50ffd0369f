#!/bin/bash
# k_format's stage seconds of a scope-E run (cfg 3's shape, plain inputs, N templates): the `format` entry of the device's stage clock,
# and microseconds per chunk of 262 144 templates.   usage: tools/format_time.sh [templates]   (on the GPU box)
cd "$(dirname "$0")/.."
N=${1:-32000000}
timeout 600 python tools/scope_bench.py --skip-b --templates $N --repeat-block --threads 16 2>&1 | tail -1 | python -c "
import sys, json, re
d = json.loads(sys.stdin.read())['E']
st = [s for s in d['stages'] if 'stage seconds' in s][0]
fmt = float(re.search(r'format ([0-9.]+)', st).group(1))
chunks = d['templates'] / 262144
print(json.dumps({'templates': d['templates'], 'wall_M_per_s': d['M_templates_per_s'], 'steady_M_per_s': d['M_templates_per_s_steady'], 'format_s': fmt, 'format_us_per_chunk': round(fmt / chunks * 1e6, 1), 'stages': st}))"
