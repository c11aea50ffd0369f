#!/bin/bash
# Developer tool: A/B of the LDS-form kernel's full-tile loop -- reads per lane R x software pipeline (PF) -- dev build.
# The product launches R = 2 pipelined; FQTK_LDSM_NOPF=1 selects the plain loop, FQTK_LDSM_PF=1 pipelines R = 1 / 4.
set -e
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip
for c in ${CONFIGS:-3 2 4}; do for r in 1 2 4; do for pf in 0 1; do
FQTK_MEMO_R=$r FQTK_LDSM_PF=$pf FQTK_LDSM_NOPF=$((1-pf)) python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c R=$r PF=$pf', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
