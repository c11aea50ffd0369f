set -x
mkdir -p gpurun_out/r04a
python -m pytest tests/test_demuxer_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r04a/demuxer_tests.txt
tools/phase_times.sh > gpurun_out/r04a/phase_times.txt 2>&1
for c in 5 3; do python bench.py --config $c --memo-table --steps 10 --warmup 2 --cpu-seconds 0 --no-verify --no-scopes 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg$c table', d['value'], d['roofline'])"; done > gpurun_out/r04a/base.txt 2>&1
cat gpurun_out/r04a/*.txt
