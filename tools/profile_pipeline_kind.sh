#!/bin/bash
# One `fqtk demux` run (cfg 3's shape) on plain | bgzf | gz inputs under rocprofv3 --kernel-trace --stats, next to the same run's own clocks.
# usage: tools/profile_pipeline_kind.sh <tag> <plain|bgzf|gz> [templates] [extra fqtk arguments...]   (on the GPU box; writes gpurun_out/<tag>/)
TAG=${1:-pipe}
KIND=${2:-gz}
N=${3:-64000000}
shift 3 2>/dev/null
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
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
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq$X $D/I1.fastq$X $D/I2.fastq$X $D/R2.fastq$X -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16 $EXTRA"
FQTK_TIMING=1 $CMD 2> $O/run_$KIND.err; grep -E "record pipeline|stage seconds|inflating|thread-seconds" $O/run_$KIND.err
rm -rf $D/out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats_$KIND.log 2>&1
rm -rf $D
f=$(find $O/stats -name "run_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/${KIND}_kernel_stats.csv && cut -c1-150 $f | head -28
rm -rf $O/stats
