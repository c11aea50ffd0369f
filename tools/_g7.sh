mkdir -p gpurun_out/r04e
python -m pytest tests/test_cli_gpu.py tests/test_demuxer_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
(time python tools/soak_cli.py --iters 10 --exe fqtk_amd/bin/fqtk.thread) > gpurun_out/r04e/tsan_soak.log 2>&1; tail -15 gpurun_out/r04e/tsan_soak.log
python tools/scope_bench.py --skip-b --templates 64000000 --threads 16 --repeat-block 2>/dev/null | tail -1 > gpurun_out/r04e/scope_E.json; python -c "
import json; d=json.load(open('gpurun_out/r04e/scope_E.json'))['E']; print('E', d['seconds'], d['M_templates_per_s'], d['M_templates_per_s_steady'], d['metrics_vs_oracle'], d['stages'])"
python tools/scope_bench.py --skip-b --templates 16000000 --threads 16 --repeat-block --extra "--devices 0,0" 2>/dev/null | tail -1 > gpurun_out/r04e/scope_E_dev00.json; python -c "
import json; d=json.load(open('gpurun_out/r04e/scope_E_dev00.json'))['E']; print('E devices 0,0', d['seconds'], d['M_templates_per_s'], d['M_templates_per_s_steady'], d['metrics_vs_oracle'])"
