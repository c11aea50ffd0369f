#!/bin/bash
# Same box, same inputs, alternating runs: `fqtk demux` (cfg 3's shape) on bgzf | gz | plain inputs with several builds of libfqtk_match.so.
# usage: tools/ab_pipeline_libs.sh <tag> <kind> <templates> <reps> <libdir or ""> ...     ("" = the product library; a libdir holds a libfqtk_match.so)
#        ENVS="A=1 B=2" adds environment to every run
TAG=$1; KIND=$2; N=$3; REPS=$4; shift 4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
D=$(mktemp -d /dev/shm/fqtk_abl_XXXX)
python - <<PY
import sys, os
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
kind, n = "$KIND", $N
if kind == "plain":
    scope_bench.make_inputs("$D", n, False, repeat_first_block=True)
else:
    paths, meta, _ = scope_bench.make_inputs("$D", 1000000, False)
    (scope_bench.bgzf_repeated if kind == "bgzf" else scope_bench.gzip_single_stream)(paths, reps=n // 1000000)
    for p in paths:
        os.unlink(p)
PY
case $KIND in plain) X="";; bgzf) X=".bgz";; gz) X=".gz";; esac
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq$X $D/I1.fastq$X $D/I2.fastq$X $D/R2.fastq$X -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
for rep in $(seq 1 $REPS); do
  for L in "$@"; do
    rm -rf $D/out
    if [ -n "$L" ]; then export LD_LIBRARY_PATH=$R/$L; else unset LD_LIBRARY_PATH; fi
    t0=$(date +%s%N)
    env $ENVS FQTK_TIMING=1 $CMD 2> $O/run_$KIND.err
    t1=$(date +%s%N)
    echo "[$KIND lib=${L:-product}] wall $(( (t1 - t0) / 1000000 )) ms; $(grep -o 'from the first chunk.*M templates/s)' $O/run_$KIND.err); inflating $(grep -o 'gzip chunks: [0-9.]*' $O/run.err | head -1)" | tee -a $O/summary_$KIND.txt
  done
done
unset LD_LIBRARY_PATH
rm -rf $D
