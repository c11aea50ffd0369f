#!/bin/bash
# Developer A/B on ONE box: `fqtk demux` (cfg 3's shape, plain inputs on RAM-backed scratch) with 1 / 2 / 4 writer threads
# and 8 / 16 --threads; prints wall-clock and steady rates.  usage: tools/ab_writers.sh [templates]
N=${1:-64000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d /dev/shm/fqtk_ab_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
scope_bench.make_inputs("$D", $N, False, repeat_first_block=True)
PY
for rep in 1 2; do for cfg in "2 16" "4 16" "1 16" "2 8" "2 32"; do
  set -- $cfg
  rm -rf $D/out
  s=$(date +%s.%N)
  FQTK_WRITERS=$1 $R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq $D/I1.fastq $D/I2.fastq $D/R2.fastq -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t $2 2> $D/err.txt
  e=$(date +%s.%N)
  echo "writers $1 threads $2: wall $(python3 -c "print(round($e - $s, 3))") s; $(grep -o '[0-9.]* s from the first chunk[^)]*)' $D/err.txt)"
done; done
rm -rf $D
