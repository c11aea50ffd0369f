python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
python tools/bench_custom.py 384 24 2>&1 | grep "S="
python tools/bench_custom.py 96 24 2>&1 | grep "S="
python tools/bench_custom.py 384 32 2>&1 | grep "S="
python tools/bench_custom.py 384 24 2 2>&1 | grep "S="
for c in 5 3; do python bench.py --config $c --memo-table --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline']['frac'], d['create_ms'])"; done
python bench.py --config 3 --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg3 lds', d['value'], d['roofline']['frac'], d['create_ms'])"
