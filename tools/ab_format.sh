#!/bin/bash
# Developer A/B of k_format build variants (FQTK_EXTRA_DEFS): the kernel's average duration under rocprofv3 in a 16 M-template run, one box.
# usage: tools/ab_format.sh "" "-DFQTK_FORMAT_OCC=5" ...   (on the GPU box; builds there)
cd "$(dirname "$0")/.."
for v in "$@"; do
    echo "=== variant: [$v]"
    FQTK_EXTRA_DEFS="$v" python -m fqtk_amd.build >/dev/null 2>&1 || { echo build failed; continue; }
    bash tools/kernel_stats_e.sh ab_format_tmp 16000000 "k_format" | tail -1
done
