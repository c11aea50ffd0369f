#!/bin/bash
# The record pipeline under rocprofv3: kernel stats of one `fqtk demux` run (cfg 3's shape, plain inputs), and the HBM
# traffic / instruction counters of its kernels in SEPARATE --pmc passes (--kernel-trace only, MI355X_MICROARCH.md HBM).
# usage: tools/profile_pipeline.sh <tag> [templates]     (on the GPU box via gpurun; writes gpurun_out/<tag>/)
TAG=${1:-pipe}
N=${2:-16000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/fqtk_prof_XXXX)
python - <<PY
import sys
sys.path.insert(0, "$R/tools"); sys.path.insert(0, "$R")
import scope_bench
scope_bench.make_inputs("$D", $N, False, repeat_first_block=True)
PY
export FQTK_CLEAN_EXIT=1
CMD="$R/fqtk_amd/bin/fqtk demux -i $D/R1.fastq $D/I1.fastq $D/I2.fastq $D/R2.fastq -r 150T 8B 8B 150T -s $D/meta.tsv -o $D/out -t 16"
FQTK_TIMING=1 $CMD 2> $O/plain_run.err; grep -E "record pipeline|stage seconds" $O/plain_run.err
rm -rf $D/out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1
rm -rf $D/out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- $CMD > $O/$c.log 2>&1; rm -rf $D/out
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/insts -o p -- $CMD > $O/insts.log 2>&1; rm -rf $D/out
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/lds -o p -- $CMD > $O/lds.log 2>&1
rm -rf $D
python - <<PY
import csv, collections, json, glob
O = "$O"
def short(n):
    n = n.split("(")[0]
    return n[:70]
stats = {}
f = glob.glob(f"{O}/stats/**/run_kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("FETCH_SIZE", "WRITE_SIZE", "insts", "lds"):
    for f in glob.glob(f"{O}/{d}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k in sorted(set(stats) | set(cnt), key=lambda k: -stats.get(k, {}).get("total_ms", 0)):
    e = dict(stats.get(k, {}))
    for c, v in cnt.get(k, {}).items():
        e[c + "_per_launch"] = round(sum(v) / len(v), 1)
    if "FETCH_SIZE_per_launch" in e:   # KB; gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads (guide)
        e["hbm_read_MB_per_launch_x2"] = round(e["FETCH_SIZE_per_launch"] * 2 * 1024 / 1e6, 3)
    if "WRITE_SIZE_per_launch" in e:
        e["hbm_write_MB_per_launch"] = round(e["WRITE_SIZE_per_launch"] * 1024 / 1e6, 3)
    out[k] = e
json.dump({"templates": $N, "chunk_templates": 262144, "kernels": out}, open(f"{O}/pipeline_profile.json", "w"), indent=1)
for k, e in list(out.items())[:14]:
    print(k, e.get("calls"), e.get("avg_us"), e.get("pct"), e.get("hbm_read_MB_per_launch_x2"), e.get("hbm_write_MB_per_launch"))
PY
