#!/bin/bash
# Developer tool: SQ / TCP / TCC counters of the table-form memo kernels (cfg 5: direct-indexed form; cfg 3 with the
# hash-table form pinned), product build; every pass is time-boxed and collects counters only (--kernel-trace).
# usage: tools/pmc_table.sh <tag>      (gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pmc_table}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 5 3; do
B="python $R/bench.py --config $cfg --memo-table --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes"
pass() { n=cfg${cfg}_$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
pass ta TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
done
python - <<PY
import csv,collections,os,glob
O="$O"
for d in sorted(glob.glob(f"{O}/cfg*_*/")):
    f=d+"p_counter_collection.csv"
    if not os.path.exists(f): print(d,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(os.path.basename(d[:-1]),k,len(v),"%.5g"%(sum(v)/len(v)))
PY
