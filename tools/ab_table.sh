#!/bin/bash
# Developer tool: the table-form kernel's loop shape (reads per lane R, pipelined or not) on the configs it
# serves -- cfg 5 (direct-indexed, IUPAC table), cfg 3 and cfg 2 with the table form pinned.  Dev build.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
run() { python bench.py --config $CFG --memo-table --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$CFG $1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || echo "cfg$CFG $1 failed"; }
for CFG in ${CONFIGS:-5 3 2}; do
run "R=1 pipelined (product)"
FQTK_MEMO_NOPF=1 run "R=1 plain"
FQTK_MEMO_R=2 run "R=2 pipelined"
FQTK_MEMO_R=2 FQTK_MEMO_NOPF=1 run "R=2 plain"
FQTK_MEMO_R=4 run "R=4 plain"
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
