#!/bin/bash
# Developer tool: LDS-form kernel, copies of the LDS histogram (FQTK_LDSM_HIST_SHIFT, dev build): 1 copy vs the
# product's choice, on the configs with few samples (cfg 4: 24, cfg 2: 96, cfg 1: 16) and on cfg 3 (384).
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
for rep in 1 2; do for c in 4 2 3; do for hs in 0 1 2 3 auto; do
if [ $hs = auto ]; then unset FQTK_LDSM_HIST_SHIFT; else export FQTK_LDSM_HIST_SHIFT=$hs; fi
python bench.py --config $c --steps 20 --warmup 3 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c hist copies 2^$hs', round(d['value']/1000,1), d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done; done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
