#!/bin/bash
# Developer tool: memory-side counters (L1 = TCP, L2 = TCC, address unit = TA) of a memo kernel, to see WHICH stage the
# waves' waiting (SQ_WAIT_ANY) is spent behind: request latencies, busy / stall cycles, queue levels.
# usage: tools/pmc_table2.sh <tag> "<bench.py flags>" [ENV=..]     (gpurun; writes gpurun_out/<tag>/; counters only)
TAG=${1:-pmc2}; FLAGS=${2:---config 5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py $FLAGS --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes"
pass() { n=$1; shift; timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tccbusy TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum
pass tccq TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass tcpq TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcpt TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
pass ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
python - <<PY
import csv,collections,os,glob
O="$O"
for d in sorted(glob.glob(f"{O}/*/")):
    f=d+"p_counter_collection.csv"
    if not os.path.exists(f): print(d,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("$TAG", os.path.basename(d[:-1]),k,len(v),"%.5g"%(sum(v)/len(v)))
PY
