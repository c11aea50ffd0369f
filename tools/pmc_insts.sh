#!/bin/bash
# Developer tool: instructions per launch of the memo kernels (SQ_INSTS_* over one bench run).
# usage: tools/pmc_insts.sh [bench.py args, default cfg 3 with 200 M reads]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
A=${*:---reads 200000000}
B="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes $A"
rm -rf $R/gpurun_out/insts
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/insts -o p -- $B > $R/gpurun_out/insts.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/insts2 -o p -- $B > $R/gpurun_out/insts2.log 2>&1
python - <<PY
import csv,collections
for d in ("insts","insts2"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open("$R/gpurun_out/%s/p_counter_collection.csv" % d)):
        if "memo" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(k[0],k[1],len(v),"%.6g"%(sum(v)/len(v)))
PY
