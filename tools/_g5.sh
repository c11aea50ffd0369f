run() { python bench.py --config $3 --steps 10 --warmup 2 --cpu-seconds 0 $2 --no-scopes 2>/tmp/err.txt | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('parity'))" || { echo "$1 failed"; tail -5 /tmp/err.txt; }; }
FQTK_DIRECT_SHAPE=thin run "cfg5 thin" "--parity full" 5
FQTK_DIRECT_SHAPE=thin run "cfg5 thin rep2" "--no-verify" 5
FQTK_DIRECT_SHAPE=fat4 run "cfg5 fat4" "--no-verify" 5
FQTK_DIRECT_SHAPE=thin run "cfg5 thin lens" "--no-verify --lens" 5
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
cd /tmp; export TMPDIR=/tmp
for sh in thin; do
FQTK_DIRECT_SHAPE=$sh timeout 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/p_$sh -o p -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes > /dev/null 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/p_$sh/p_counter_collection.csv")):
    if "memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$sh", {k: "%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
done
