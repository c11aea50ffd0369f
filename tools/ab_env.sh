#!/bin/bash
# Developer tool: bench a config under several values of one dev-build environment knob.
# usage: VAR=FQTK_MEMO_SLOT_FACTOR VALUES="4 2 1" CONFIGS="5" tools/ab_env.sh [extra bench args]
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
for rep in 1 2; do for c in ${CONFIGS:-5}; do for v in $VALUES; do
env $VAR=$v python bench.py --config $c --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c $VAR=$v', d['value'], d['roofline']['kernel_ms'])"
done; done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
