#!/bin/bash
# Serial gzip inputs through `fqtk demux`, decoded on the device in chunks vs by the host's decoders: same box, same files, alternating runs.
# usage: tools/gz_ab.sh <tag> [templates] [reps] [real]   (on the GPU box via gpurun; writes gpurun_out/<tag>/)
#   real: the inputs are compressed by `gzip -6` / `gzip -1` as whole files (no repeated deflate blocks), 4 M templates at most
TAG=${1:-gz_ab}
N=${2:-16000000}
REPS=${3:-2}
REAL=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_gzab_XXXX)
python - <<PY
import sys, subprocess
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
if "$REAL":
    paths, meta, _ = scope_bench.make_inputs("$D", $N, False)
    procs = [subprocess.Popen(["gzip", "-$REAL", "-f", p]) for p in paths]
    assert all(p.wait() == 0 for p in procs)
else:
    paths, meta, _ = scope_bench.make_inputs("$D", $N, False, repeat_first_block=True)
    scope_bench.gzip_single_stream(paths)
PY
ls -l $D > $O/inputs.txt
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq.gz $D/I1.fastq.gz $D/I2.fastq.gz $D/R2.fastq.gz -r 150T 8B 8B 150T -s $D/meta.tsv -t 16"
run() {   # name, extra args
    local name=$1; shift
    local t0=$(date +%s.%N)
    FQTK_TIMING=1 $CMD -o $D/out_$name "$@" 2> $O/$name.err
    local rc=$?
    local t1=$(date +%s.%N)
    local steady=$(grep -o "([0-9.]* M templates/s)" $O/$name.err | tr -d '()' | cut -d' ' -f1)
    local t_end=$(grep -o "epoch [0-9.]*" $O/$name.err | tail -1 | cut -d' ' -f2)
    local t_log=$(grep "at the end" $O/$name.err | tail -1 | sed 's/^\[ *\([0-9.]*\) .*/\1/')
    [ -n "$t_end" ] && echo "$name: before main $(awk -v a=$t0 -v e=$t_end -v l=$t_log 'BEGIN { printf "%.3f", e - l - a }') s, main $t_log s, after the last line $(awk -v b=$t1 -v e=$t_end 'BEGIN { printf "%.3f", b - e }') s" | tee -a $O/summary.txt
    echo "$name rc=$rc $(awk -v a=$t0 -v b=$t1 -v n=$N 'BEGIN { printf "wall_s=%.3f M_templates_per_s_wall=%.2f", b - a, n / (b - a) / 1e6 }') steady=$steady" | tee -a $O/summary.txt
    md5sum $D/out_$name/demux-metrics.txt | cut -c1-32 >> $O/summary.txt
    rm -rf $D/out_$name
}
for r in $(seq 1 $REPS); do
    run device$r
    run host$r --host-inflate
done
grep -h "gzip input" $O/device1.err | head -40 > $O/stretches.txt
grep -hE "stage seconds|thread-seconds|inflating" $O/device1.err >> $O/stretches.txt
rm -rf $D
cat $O/summary.txt
