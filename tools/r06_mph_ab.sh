#!/bin/bash
# Round 6: the LDS form behind a minimal perfect hash (three-byte entries) against what served the same tables before.
# usage (GPU box): bash tools/r06_mph_ab.sh > gpurun_out/r06_mph_ab.txt
# FQTK_LDS_MPH: 0 = never, 1 = where the cuckoo form has no room (default), 2 = wherever it can be planned.
cd "$(dirname "$0")/.."
for shape in "384 24" "440 24" "384 20" "320 24" "200 24"; do
    for mode in 0 2; do
        echo "== S L = $shape  FQTK_LDS_MPH=$mode"
        FQTK_LDS_MPH=$mode timeout 600 python tools/bench_custom.py $shape 1 2 2>&1 | grep "G reads/s"
    done
done
