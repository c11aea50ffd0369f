#!/bin/bash
# Kernel stats (rocprofv3 --kernel-trace --stats) of one `fqtk demux` run on plain inputs of cfg 3's shape.
# usage: tools/kernel_stats_e.sh <tag> [templates] [pattern]   (on the GPU box; writes gpurun_out/<tag>/run_kernel_stats.csv, prints the kernels that match pattern)
TAG=${1:-kstats}
N=${2:-16000000}
PAT=${3:-.}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
scope_bench.make_inputs("$D", $N, False, repeat_first_block=True)
PY
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq $D/I1.fastq $D/I2.fastq $D/R2.fastq -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1
rm -rf $D
f=$(find $O/stats -name "run_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/run_kernel_stats.csv && python - <<PY
import csv
for r in csv.DictReader(open("$O/run_kernel_stats.csv")):
    import re
    if re.search(r"$PAT", r["Name"]):
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "min_us", round(float(r["MinNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
rm -rf $O/stats
