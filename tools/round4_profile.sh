#!/bin/bash
# Round 4's measured artefacts in one go (run through gpurun): writes gpurun_out/<tag>/, from where the summaries are copied
# into profiles/.   usage: tools/round4_profile.sh r04_final
TAG=${1:-r04_final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
bash tools/profile_bench.sh ${TAG}_prof > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-300
STEPS=10 bash tools/bench_matrix.sh > $O/bench_matrix.jsonl 2> /dev/null
for c in 3 2 5; do python bench.py --config $c --lens --steps 10 --warmup 2 --cpu-seconds 0 --parity windows --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'config': d['config']['workload'][:5].strip(), 'obs_len': d['config']['obs_len'], 'memo_kind': d['config']['memo_kind'], 'G_reads_s': round(d['value']/1000,1), 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch'], 'parity': d['config']['parity']}))" >> $O/bench_lens.jsonl; done
for a in "384 24 1 2" "384 24 2 2" "384 32 1 2"; do echo "== bench_custom $a" >> $O/bench_custom.txt; timeout 300 python tools/bench_custom.py $a >> $O/bench_custom.txt 2>&1; done
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
for l in 1 6 9; do python tools/inflate_bench.py --level $l 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl; done
python tools/inflate_bench.py --const-qual 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl
python tools/inflate_bench.py --members 1024 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl
for v in "" "--bgzf" "--bgzf --extra=--host-inflate"; do FQTK_TIMING=1 timeout 300 python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads 16 $v >> $O/scope_E.jsonl 2>> $O/scope_E.err; done
FQTK_TIMING=1 timeout 300 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err
bash tools/profile_pipeline_bgzf.sh ${TAG}_pipe 16000000 > $O/profile_pipeline_bgzf.log 2>&1
python tools/soak_cli.py --iters ${SOAK_CLI:-80} --seed 23 > $O/soak_cli.log 2>&1; tail -1 $O/soak_cli.log
ls $O
