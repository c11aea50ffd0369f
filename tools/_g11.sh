for c in 3 2 4; do python bench.py --config $c --memo-table --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline']['frac'])"; done
python bench.py --config 3 --memo-table --lens --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg3 table lens', d['value'], d['roofline']['frac'])"
python tools/bench_custom.py 384 24 2>&1 | grep "S="
python tools/bench_custom.py 384 32 2>&1 | grep "S="
python tools/bench_custom.py 384 24 2 2>&1 | grep "S="
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
