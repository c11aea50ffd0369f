mkdir -p gpurun_out/r04c
run() { python bench.py --config 5 --steps 10 --warmup 2 --cpu-seconds 0 $2 --no-scopes 2>/tmp/err.txt | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('parity'))" || { echo "$1 failed"; tail -5 /tmp/err.txt; }; }
for sh in thin fat2 fat4; do FQTK_DIRECT_SHAPE=$sh run "cfg5 $sh" "--parity full"; done
for sh in thin thinpf fat2 fat4; do FQTK_DIRECT_SHAPE=$sh run "cfg5 $sh rep2" "--no-verify"; done
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
for sh in thin fat4; do
FQTK_DIRECT_SHAPE=$sh timeout 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/p_$sh -o p -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-scopes > /dev/null 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/p_$sh/p_counter_collection.csv")):
    if "memo_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$sh", {k: "%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
done
