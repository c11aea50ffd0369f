#!/bin/bash
# Developer tool: SQ counters of the BGZF member decoder (tools/inflate_bench.py, 8192 members), counters only (--kernel-trace).
# usage: tools/pmc_inflate.sh <tag>      (gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pmc_inflate}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/inflate_bench.py --members 8192 --reps 2 $BENCH_ARGS"
pass() { n=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH
python - <<PY
import csv,collections,os,glob
O="$O"
for d in sorted(glob.glob(f"{O}/sq*/")):
    f=d+"p_counter_collection.csv"
    if not os.path.exists(f): print(d,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "inflate_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(os.path.basename(d[:-1]),k,len(v),"%.5g"%(sum(v)/len(v)))
PY
