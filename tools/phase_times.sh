#!/bin/bash
# Developer tool: tools/phase_times.py under the -DFQTK_DEV_TIMING build kept in fqtk_amd/lib/variants/timing/ (built here:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DFQTK_DEV_TIMING -DFQTK_DEV_ABLATE -c fqtk_amd/csrc/fqtk_match.hip ...)
cd "$(dirname "$0")/.."
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
cp fqtk_amd/lib/variants/timing/libfqtk_match.so fqtk_amd/lib/libfqtk_match.so
python tools/phase_times.py 5
python tools/phase_times.py 3 --memo-table
python tools/phase_times.py 2 --memo-table
FQTK_MEMO_R=1 python tools/phase_times.py 5
cp /tmp/libfqtk_match.prod.so fqtk_amd/lib/libfqtk_match.so
