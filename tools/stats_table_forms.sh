#!/bin/bash
# rocprofv3 --kernel-trace --stats of the table-form memo kernels (their fractions rested on HIP events only):
# cfg 5 (direct-indexed form), cfg 3 / 2 / 4 with the hash-table form pinned, a 12+12 dual index.  usage (gpurun): tools/stats_table_forms.sh <tag>
TAG=${1:-table_forms_stats}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- "$@" > $O/$name.log 2>&1
    local f=$(find $O/$name -name "s_kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "== $name: $*" | sed "s#$R/##g"; grep -E "Name|memo_kernel|match_kernel" $f; } >> $O/summary.csv
    grep -h "G reads/s\|\"value\"" $O/$name.log | tail -2 | cut -c1-400 >> $O/bench_lines.txt
}
B="python $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes"
run cfg5 $B --config 5
run cfg3_table $B --config 3 --memo-table
run cfg2_table $B --config 2 --memo-table
run cfg4_table $B --config 4 --memo-table
run dual_12_12 python $R/tools/bench_custom.py 384 24 1 2
cat $O/summary.csv
