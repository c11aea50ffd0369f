#!/bin/bash
# Developer tool: PMC passes over the memo kernel (200 M-read cfg3 batch); every pass is time-boxed.
# usage: tools/pmc_memo.sh <tag>
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes --reads 200000000"
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_FLAT
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
pass tcc2 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_READ_sum TCC_CYCLE_sum
python - <<PY
import csv,collections,os
O="$O"
for d in ("sq1","sq2","sq3","tcp1","tcp2","tcc","tcc2"):
    f=f"{O}/{d}/p_counter_collection.csv"
    if not os.path.exists(f): print(d,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "memo" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,len(v),"%.5g"%(sum(v)/len(v)))
PY
