cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes --reads 200000000"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/insts -o p -- $B > $R/gpurun_out/insts.log 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/insts/p_counter_collection.csv")):
    if "lds_memo" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k,len(v),"%.6g"%(sum(v)/len(v)))
PY
