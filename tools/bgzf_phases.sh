#!/bin/bash
# Developer tool: where the BGZF compressor kernel spends its time, phase by phase (dev build with timestamps).
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c -DFQTK_BGZF_PHASE_TIMES ${LZ_TIMES:+-DFQTK_BGZF_LZ_TIMES} ${LZ_COUNTS:+-DFQTK_BGZF_LZ_COUNTS} $EXTRA_DEFS -o /tmp/bgzf_ph.o fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/lib/obj/fqtk_match.hip.o fqtk_amd/lib/obj/fqtk_demux.hip.o fqtk_amd/lib/obj/fqtk_inflate.hip.o /tmp/bgzf_ph.o || exit 1
python - <<'PY'
import ctypes as C, json, subprocess, sys
sys.path.insert(0, ".")
import runpy
from fqtk_amd import _lib
lib = _lib.load()
import os
sys.argv = ["bgzf_bench.py", "--hbm-only"] + os.environ.get("BENCH_ARGS", "").split()
runpy.run_path("tools/bgzf_bench.py", run_name="__main__")
t = (C.c_ulonglong * 16)()
assert lib.fqtk_bgzf_dev_phase_ticks(t) == 0
names = ["load", "index", "count + literal costs", "reach", "rank", "code lengths (two lanes)", "19-symbol code || count bits", "offsets", "emit", "store", "header bits", "canonical codes", "lz", "clear (inside clear + rank, lane 0)", "code-length runs", "code-length symbols"]
tot = sum(t[:16])
for k, nme in enumerate(names):
    print(f"{nme:28s} {100.0 * t[k] / tot:5.1f} %")
z = (C.c_ulonglong * 10)()
assert lib.fqtk_bgzf_dev_lz_cycles(z) == 0
if os.environ.get("LZ_COUNTS"):   # events per wavefront (the first active lane counts), over every launch of the run above
    steps = max(z[0], 1)
    print(f"lz, wavefront events per step: match passes {z[1] / steps:.3f}, second passes {z[2] / steps:.3f}, extension rounds {z[3] / steps:.3f}, "
          f"matches taken {z[4] / steps:.3f}, pending inserts {z[5] / steps:.3f}; steps per wavefront: {z[0] / (16.0 * 4096 * 6):.1f} (six launches of 4096 blocks)")
elif os.environ.get("LZ_TIMES"):
    waves, steps = max(z[9], 1), max(z[8], 1)
    print(f"lz, per wave: {steps / waves:.0f} steps; history preload {z[4] / waves:.0f} cycles; cycles per step: table reads {z[0] / steps:.0f}, "
          f"candidate reads + literal costs {z[1] / steps:.0f}, match extension {z[2] / steps:.0f}, match token {z[5] / steps:.0f}, "
          f"history inserts {z[6] / steps:.0f}, literal token {z[7] / steps:.0f}, rest {z[3] / steps:.0f}")
PY
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
