#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, and the HBM-traffic
# PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, --kernel-trace only; MI355X_MICROARCH.md HBM).
# usage: tools/profile_bench.sh <tag>     (run on the GPU box via gpurun; writes gpurun_out/<tag>/)
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-200
B="python $R/bench.py --cpu-seconds 0 --no-verify --no-scopes"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
python - <<PY
import csv, json, collections
O="$O"
out={}
for d,c in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    rows=[r for r in csv.DictReader(open(f"{O}/{d}/p_counter_collection.csv")) if "memo_kernel" in r["Kernel_Name"] or "match_kernel" in r["Kernel_Name"]]
    vals=[float(r["Counter_Value"]) for r in rows if r["Counter_Name"]==c]
    vals=vals[-10:]
    out[c+"_KB_per_launch"]=sum(vals)/max(len(vals),1)
    out["kernel"]=rows[-1]["Kernel_Name"] if rows else None
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read
out["hbm_read_bytes_corrected"]=out["FETCH_SIZE_KB_per_launch"]*1024*2
out["hbm_write_bytes"]=out["WRITE_SIZE_KB_per_launch"]*1024
out["traffic_bytes_per_launch"]=out["hbm_read_bytes_corrected"]+out["hbm_write_bytes"]
import sys
sys.path.insert(0, "$R")
import bench
out["kernel_sources_sha1"]=bench.kernel_sources_digest()   # bench.py reports this traffic only for the kernels it was measured on
json.dump(out, open(f"{O}/pmc_traffic.json","w"), indent=1)
# the form bench.py reads as roofline.traffic (copy to profiles/pmc_traffic.json)
json.dump({"cfg3": {"reads_per_launch": 400000000, "kernel_prefix": "fqtk::lds_memo_kernel",
                    "traffic_bytes_per_launch": out["traffic_bytes_per_launch"], "kernel_sources_sha1": out["kernel_sources_sha1"],
                    "source": "tools/profile_bench.sh"}}, open(f"{O}/pmc_traffic_for_bench.json","w"), indent=1)
print(out)
PY
cat $O/stats/bench_kernel_stats.csv | head -4 | cut -c1-200
