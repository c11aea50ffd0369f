#!/bin/bash
# Developer tool: counters of the record pipeline's kernels in a `fqtk demux` run without compression (--compression-level 0:
# stream A's kernels then run without the DEFLATE kernel beside them).  usage: tools/pmc_format.sh <tag>   (gpurun)
TAG=${1:-pmc_format}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
scope_bench.make_inputs("$D", 8000000, False, repeat_first_block=True)
PY
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq $D/I1.fastq $D/I2.fastq $D/R2.fastq -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16 --compression-level 0"
pass() { n=$1; shift; timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $CMD > $O/$n.log 2>&1 || echo "pass $n failed"; rm -rf $D/out; }
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1; rm -rf $D/out
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS
# (round 5: the memory-counter passes of this command hung on the pool's boxes -- three timeouts, 15 GPU-minutes; they run only when asked for: MEM_PASSES=1)
if [ -n "$MEM_PASSES" ]; then
pass mem FETCH_SIZE WRITE_SIZE
pass mem2 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
fi
pass vm SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
rm -rf $D
python - <<PY
import csv,collections,os,glob
O="$O"
f=glob.glob(f"{O}/stats/**/run_kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if float(r["Percentage"])>1: print("stats", r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), r["Percentage"])
for d in sorted(glob.glob(f"{O}/*/")):
    f=d+"p_counter_collection.csv"
    if not os.path.exists(f): continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_format" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("k_format", os.path.basename(d[:-1]),k,len(v),"%.5g"%(sum(v)/len(v)))
PY
