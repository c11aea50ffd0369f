#!/bin/bash
# Developer tool: L2 request counters of the table-form kernel on cfg 5, with the direct-indexed form (product)
# and without it (FQTK_NO_DIRECT=1: every entry in the cuckoo table, the round-1 structure).  Dev build.
# usage: tools/pmc_cfg5.sh <tag>      (gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pmc_cfg5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
cp fqtk_amd/lib/libfqtk_match.so /tmp/libfqtk_match.prod.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DFQTK_DEV_ABLATE -o fqtk_amd/lib/libfqtk_match.so fqtk_amd/csrc/fqtk_match.hip fqtk_amd/csrc/fqtk_bgzf.hip || exit 1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config 5 --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes"
for nd in 0 1; do
  export FQTK_NO_DIRECT=$nd
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/tcp_nd$nd -o p -- $B > $O/tcp_nd$nd.log 2>&1 || echo "pass tcp nd=$nd failed"
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tcc_nd$nd -o p -- $B > $O/tcc_nd$nd.log 2>&1 || echo "pass tcc nd=$nd failed"
done
cp /tmp/libfqtk_match.prod.so $R/fqtk_amd/lib/libfqtk_match.so
python - <<PY
import csv, collections, os, json
O="$O"; out={}
for nd in (0,1):
    for grp in ("tcp","tcc"):
        f=f"{O}/{grp}_nd{nd}/p_counter_collection.csv"
        if not os.path.exists(f): continue
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): out[f"{'direct' if nd==0 else 'hash_table_only'}.{k}"]=sum(v)/len(v)
out["reads_per_launch"]=50000000
json.dump(out, open(f"{O}/pmc_cfg5.json","w"), indent=1); print(json.dumps(out, indent=1))
PY
