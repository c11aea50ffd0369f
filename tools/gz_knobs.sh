#!/bin/bash
# Serial gzip inputs (cfg 3's shape) through `fqtk demux` under several settings of the device decoder's knobs, same box, same files.
# usage: tools/gz_knobs.sh <tag> [templates] "<ENV=.. ENV=..>" ...      (each quoted argument is one run's environment; "" = defaults)
TAG=${1:-gz_knobs}
N=${2:-64000000}
shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
D=$(mktemp -d /dev/shm/fqtk_gzk_XXXX)
python - <<PY
import sys, os
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
paths, meta, _ = scope_bench.make_inputs("$D", 1000000, False)
scope_bench.gzip_single_stream(paths, reps=$N // 1000000)
for p in paths:
    os.unlink(p)
PY
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq.gz $D/I1.fastq.gz $D/I2.fastq.gz $D/R2.fastq.gz -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
i=0
for E in "$@"; do
  for rep in 1 2; do
    rm -rf $D/out
    t0=$(date +%s.%N)
    env $E FQTK_TIMING=1 $CMD 2> $O/run_$i.err
    t1=$(date +%s.%N)
    echo "[$E] $(grep -o 'from the first chunk.*devices ready at [0-9.]* s' $O/run_$i.err); first chunk cut at $(grep 'first chunk cut' $O/run_$i.err | grep -o '^\[ *[0-9.]*')" | tee -a $O/summary.txt
  done
  i=$((i+1))
done
rm -rf $D
