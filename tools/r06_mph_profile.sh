#!/bin/bash
# Round 6: the perfect-hash LDS form on a 384 x 24 table (12+12 dual index) under the profiler: kernel stats by rocprofv3's clock, then
# counter passes (counters only, --kernel-trace), per launch and per read; the same for the cuckoo LDS form on 384 x 20 beside it.
# usage (gpurun): tools/r06_mph_profile.sh <tag>      writes gpurun_out/<tag>/{summary.csv,pmc.txt}
TAG=${1:-r06_mph}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[mph_384x24]="python $R/tools/bench_custom.py 384 24 1 2"
CMD[cuckoo_384x20]="python $R/tools/bench_custom.py 384 20 1 2"
for name in mph_384x24 cuckoo_384x20; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -o s -- ${CMD[$name]} > $O/st_$name.log 2>&1
    f=$(find $O/st_$name -name "s_kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "== $name: ${CMD[$name]}" | sed "s#$R/##g"; grep -E "Name|memo_kernel" $f; } >> $O/summary.csv
    rm -rf $O/st_$name
    pass() { n=${name}_$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- ${CMD[$name]} > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
    pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR
    pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
    pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
done
python - > $O/pmc.txt <<PY
import csv,collections,os,glob
O="$O"
n=100_000_000
for d in sorted(glob.glob(f"{O}/*_*/")):
    f=d+"p_counter_collection.csv"
    name=os.path.basename(d[:-1])
    if not os.path.exists(f): print(name,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "lds_memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(name,k,"launches",len(v),"per_launch %.5g"%(sum(v)/len(v)),"per_read %.4g"%(sum(v)/len(v)/n))
PY
cat $O/summary.csv; cat $O/pmc.txt
rm -rf $O/*_sq1 $O/*_sq2 $O/*_tcp1
