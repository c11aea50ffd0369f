#!/bin/bash
# Developer tool: A/B builds of the BGZF member decoder on the GPU box.  Every argument is one variant: a string of
# -D defines ("" = the product).  Only fqtk_inflate.hip is recompiled; the other objects of libfqtk_match.so are reused.
#   FQTK_INFLATE_ABL_NOCOPY   matches are not copied (wrong bytes, same control flow): what the match path costs
#   FQTK_INFLATE_ABL_NOSTORE  literals are not stored either
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
for defs in "$@"; do
  echo "=== variant: [$defs]"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c $defs -o /tmp/inflate_ab.o fqtk_amd/csrc/fqtk_inflate.hip || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/lib/obj/fqtk_match.hip.o fqtk_amd/lib/obj/fqtk_bgzf.hip.o fqtk_amd/lib/obj/fqtk_demux.hip.o /tmp/inflate_ab.o || exit 1
  chk=""; [ -n "$defs" ] && [[ "$defs" == *ABL* ]] && chk="--no-check"
  for m in ${MEMBERS:-8192 1024}; do
    python tools/inflate_bench.py --members $m $chk 2>/dev/null | tail -1
    python tools/inflate_bench.py --members $m --const-qual $chk 2>/dev/null | tail -1
  done
done
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
