#!/bin/bash
# Developer tool: A/B builds of the BGZF compressor kernel on the GPU box.  Every argument is one variant: a string of
# -D defines ("" = the product).  Only fqtk_bgzf.hip is recompiled; the other objects of libfqtk_match.so are reused.
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for defs in "$@"; do
  echo "=== variant: [$defs]"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c $defs -o /tmp/bgzf_ab.o fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/lib/obj/fqtk_match.hip.o fqtk_amd/lib/obj/fqtk_demux.hip.o fqtk_amd/lib/obj/fqtk_inflate.hip.o /tmp/bgzf_ab.o || exit 1
  python tools/bgzf_bench.py --blocks ${BLOCKS:-4096} $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:d[k] for k in ('hbm',)})"
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
