#!/bin/bash
# `fqtk demux` on serial gzip inputs (decoded on the device in chunks) under rocprofv3: kernel stats of one run (cfg 3's shape).
# usage: tools/profile_pipeline_gz.sh <tag> [templates]     (on the GPU box via gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pipe_gz}
N=${2:-16000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
paths, meta, _ = scope_bench.make_inputs("$D", $N, False, repeat_first_block=True)
scope_bench.gzip_single_stream(paths)
PY
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq.gz $D/I1.fastq.gz $D/I2.fastq.gz $D/R2.fastq.gz -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
FQTK_TIMING=1 $CMD 2> $O/run.err; grep -E "record pipeline|stage seconds|inflating|thread-seconds|gzip input . *: [0-9]* chunks" $O/run.err
rm -rf $D/out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1
rm -rf $D
f=$(find $O/stats -name "run_kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 $f | head -12
