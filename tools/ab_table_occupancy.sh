#!/bin/bash
# Developer tool: table-form kernel (cfg 5, direct-indexed), registers against occupancy -- the two-reads-per-lane
# software pipeline needs more than the 64 VGPRs that 8 waves per SIMD allow.  Dev builds.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
run() { python bench.py --config ${CFG:-5} --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1000,1), d['roofline']['kernel_ms'])" || echo "$1 failed"; }
for v in ${VARIANTS:-"1024 8 2" "512 6 3" "256 6 6" "256 5 5" "512 4 2" "256 4 4"}; do set -- $v
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -DFQTK_MEMO_BLOCK=$1 -DFQTK_MEMO_WAVES=$2 -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || continue
export FQTK_MEMO_PER_CU=$3
FQTK_MEMO_R=2 run "block $1 waves/SIMD $2 per CU $3: R=2 plain"
FQTK_MEMO_R=2 FQTK_MEMO_PF=1 run "block $1 waves/SIMD $2 per CU $3: R=2 pipelined"
FQTK_MEMO_R=1 FQTK_MEMO_PF=1 run "block $1 waves/SIMD $2 per CU $3: R=1 pipelined"
FQTK_MEMO_R=4 run "block $1 waves/SIMD $2 per CU $3: R=4 plain"
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
