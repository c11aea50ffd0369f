#!/bin/bash
# The record pipeline on BGZF inputs (members inflated on the device) under rocprofv3: kernel stats of one `fqtk demux`
# run (cfg 3's shape), next to the same run's own stage clock.
# usage: tools/profile_pipeline_bgzf.sh <tag> [templates]     (on the GPU box via gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pipe_bgzf}
N=${2:-16000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
python - <<PY
import sys, os
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
paths, meta, _ = scope_bench.make_inputs("$D", 1000000, False)
scope_bench.bgzf_repeated(paths, reps=$N // 1000000)
PY
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq.bgz $D/I1.fastq.bgz $D/I2.fastq.bgz $D/R2.fastq.bgz -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
FQTK_TIMING=1 $CMD 2> $O/run.err; grep -E "record pipeline|stage seconds|inflating|thread-seconds" $O/run.err
rm -rf $D/out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1
rm -rf $D
f=$(find $O/stats -name "run_kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 $f | head -24
