#!/bin/bash
# Developer A/B of k_format build variants: scope E's `format` stage seconds (64 M templates), one box.
cd "$(dirname "$0")/.."
for v in "$@"; do
    echo "=== variant: [$v]"
    FQTK_EXTRA_DEFS="$v" python -m fqtk_amd.build >/dev/null 2>&1 || { echo build failed; continue; }
    timeout 300 python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads 16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['E']; print(d['seconds'], d['M_templates_per_s_steady'], d['stages'][0])"
done
python -m fqtk_amd.build >/dev/null 2>&1
