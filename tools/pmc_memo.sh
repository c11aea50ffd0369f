#!/bin/bash
# Developer tool: PMC passes over the memo kernel (200 M-read cfg3 batch).  usage: tools/pmc_memo.sh <tag> [env...]
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --reads 200000000"
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum
python - <<PY
import csv,collections,glob,os
O="$O"
for d in ("sq1","sq2","sq3","ta","tcp","tcp2"):
    f=f"{O}/{d}/p_counter_collection.csv"
    if not os.path.exists(f): print(d,"missing", open(f"{O}/{d}.log").read()[-300:]); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "memo" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,len(v),"%.4g"%(sum(v)/len(v)))
PY
