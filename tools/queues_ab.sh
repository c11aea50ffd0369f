#!/bin/bash
# How many hardware queues the HIP runtime may open (GPU_MAX_HW_QUEUES) vs `fqtk demux` on serial gzip inputs: resident memory, steady rate, wall clock.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_q_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
paths, meta, _ = scope_bench.make_inputs("$D", 16000000, False, repeat_first_block=True)
scope_bench.gzip_single_stream(paths)
PY
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq.gz $D/I1.fastq.gz $D/I2.fastq.gz $D/R2.fastq.gz -r 150T 8B 8B 150T -s $D/meta.tsv -t 16"
for q in "" 1 2 3 8 ""; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    if [ -n "$q" ]; then GPU_MAX_HW_QUEUES=$q FQTK_TIMING=1 $CMD -o $D/out 2> $D/err; else FQTK_TIMING=1 $CMD -o $D/out 2> $D/err; fi
    t1=$(date +%s.%N)
    echo "GPU_MAX_HW_QUEUES=[$q] wall $(awk -v a=$t0 -v b=$t1 'BEGIN{printf "%.3f", b-a}') steady $(grep -o "([0-9.]* M templates/s)" $D/err | tr -d '()' | cut -d' ' -f1) $(grep -o "RssAnon: *[0-9]* kB" $D/err | tail -1) main $(grep "at the end" $D/err | sed 's/^\[ *\([0-9.]*\) .*/\1/')"
    rm -rf $D/out
  done
done
rm -rf $D
