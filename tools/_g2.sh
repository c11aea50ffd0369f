mkdir -p gpurun_out/r04b
run() { python bench.py --config 5 --steps 10 --warmup 2 --cpu-seconds 0 $2 --no-scopes 2>/tmp/err.txt | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('parity'))" || { echo "$1 failed"; tail -5 /tmp/err.txt; }; }
for sh in thin thinpf fat2 fat4; do FQTK_DIRECT_SHAPE=$sh run "cfg5 $sh" "--parity full"; done
for sh in thin thinpf fat2 fat4; do FQTK_DIRECT_SHAPE=$sh run "cfg5 $sh rep2" "--no-verify"; done
FQTK_DIRECT_SHAPE=fat4 python bench.py --config 5 --reads 400000000 --steps 5 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg5 fat4 400M', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
