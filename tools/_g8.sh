mkdir -p gpurun_out/r04f
python -m pytest tests/test_demuxer_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
python tools/scope_bench.py --skip-b --templates 64000000 --threads 16 --repeat-block 2>/dev/null | tail -1 > gpurun_out/r04f/scope_E.json; python -c "
import json; d=json.load(open('gpurun_out/r04f/scope_E.json'))['E']; print('E', d['seconds'], d['M_templates_per_s'], d['M_templates_per_s_steady'], d['stages'][0])"
