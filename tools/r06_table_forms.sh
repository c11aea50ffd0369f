#!/bin/bash
# Round 6, VERDICT r05 item 5(b): the table-form memo kernels closed WITH counters.  For cfg 5 (direct-indexed form), cfg 3 with the hash-table
# form pinned and a 12+12 dual index (both with the presence filter, the product build): rocprofv3 kernel stats, then counter passes
# (counters only, --kernel-trace): L1 -> L2 read requests, L1 pending-stall / gated cycles, address-unit busy, L2 hits / requests.
# usage (gpurun): tools/r06_table_forms.sh <tag>      writes gpurun_out/<tag>/{summary.csv,pmc.txt}
TAG=${1:-r06_table_forms}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes"
declare -A CMD
CMD[cfg5]="$B --config 5"
CMD[cfg3_table]="$B --config 3 --memo-table"
CMD[dual_12_12]="python $R/tools/bench_custom.py 384 24 1 2"
for name in cfg5 cfg3_table dual_12_12; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -o s -- ${CMD[$name]} > $O/st_$name.log 2>&1
    f=$(find $O/st_$name -name "s_kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "== $name: ${CMD[$name]}" | sed "s#$R/##g"; grep -E "Name|memo_kernel|match_kernel" $f; } >> $O/summary.csv
    rm -rf $O/st_$name
    pass() { n=${name}_$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- ${CMD[$name]} > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
    pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
    pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
    pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
    pass ta TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
    pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
done
python - > $O/pmc.txt <<PY
import csv,collections,os,glob
O="$O"
reads={"cfg5":50_000_000,"cfg3_table":400_000_000,"dual_12_12":100_000_000}
for d in sorted(glob.glob(f"{O}/*_*/")):
    f=d+"p_counter_collection.csv"
    name=os.path.basename(d[:-1])
    if not os.path.exists(f): print(name,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "memo_kernel" in r["Kernel_Name"] and "lds_memo" not in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n=[v for k,v in reads.items() if name.startswith(k)][0]
    for k,v in agg.items(): print(name,k,"launches",len(v),"per_launch %.5g"%(sum(v)/len(v)),"per_read %.4g"%(sum(v)/len(v)/n))
PY
cat $O/summary.csv; cat $O/pmc.txt
rm -rf $O/*_tcp1 $O/*_tcp2 $O/*_tcc $O/*_ta $O/*_sq
