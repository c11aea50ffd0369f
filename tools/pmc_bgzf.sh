#!/bin/bash
# Developer tool: instruction and LDS counters of the BGZF kernel (tools/bgzf_bench.py), separate time-boxed --pmc passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_bgzf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bgzf_bench.py --blocks 4096 --reps 2 $BENCH_ARGS"
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass lds SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
python - <<PY
import csv, glob, collections
for n in ("insts", "lds"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % n, recursive=True):
        acc = collections.defaultdict(float); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "deflate_kernel" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
        for k in acc: print(n, k, "per launch: %.4g" % (acc[k] / max(calls[k], 1)), "launches", calls[k])
PY
