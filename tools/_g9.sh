mkdir -p gpurun_out/r04f
cp fqtk_amd/lib/libfqtk_match.so /tmp/prod.so
for w in 8 6 5 4; do
cp fqtk_amd/lib/variants/fmt$w/libfqtk_match.so fqtk_amd/lib/libfqtk_match.so
python tools/scope_bench.py --skip-b --templates 32000000 --threads 16 --repeat-block 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())['E']; print('waves $w: E', d['seconds'], d['M_templates_per_s'], d['M_templates_per_s_steady'], d['stages'][0])"
done
cp /tmp/prod.so fqtk_amd/lib/libfqtk_match.so
python -m pytest tests/test_demuxer_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
