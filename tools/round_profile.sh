#!/bin/bash
# Every measured artefact of a round in one go (run through gpurun): writes gpurun_out/<tag>/, from where the
# summaries are copied into profiles/.   usage: tools/round_profile.sh r02_final
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
bash tools/profile_bench.sh ${TAG}_prof > $O/profile.log 2>&1; tail -4 $O/profile.log
STEPS=10 bash tools/bench_matrix.sh > $O/bench_matrix.jsonl 2> /dev/null
bash tools/bench_cliff.sh > $O/cliff.jsonl 2>&1
for c in 3 2 5; do python bench.py --config $c --lens --steps 10 --warmup 2 --cpu-seconds 0 --parity windows --no-scopes 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'config': d['config']['workload'][:5].strip(), 'obs_len': d['config']['obs_len'], 'memo_kind': d['config']['memo_kind'], 'G_reads_s': round(d['value']/1000,1), 'frac': r['frac'], 'kernel_ms': r['kernel_ms'], 'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_launch'], 'parity': d['config']['parity']}))" >> $O/bench_lens.jsonl; done
for a in "--config 3" "--config 2" "--config 4" "--config 5" "--config 1" "--config 3 --table" "--config 2 --table"; do python tools/full_parity.py $a 2>/dev/null | tail -1 >> $O/full_parity.jsonl; done
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
for v in "--threads 16" "--threads 16 --extra=--gpu-bgzf" "--threads 8" "--threads 8 --extra=--gpu-bgzf" "--threads 32" "--threads 32 --extra=--gpu-bgzf"; do python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block $v >> $O/scope_E.jsonl 2>> $O/scope_E.err; done
for v in "" "--extra=--gpu-bgzf"; do python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz $v >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err; done
FQTK_ZLIB_INFLATE=1 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz >> $O/scope_E_gz_zlib.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 8000000 --repeat-block --threads 16 --bgzf >> $O/scope_E.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 8000000 --repeat-block --threads 16 --bgzf --extra=--gpu-bgzf >> $O/scope_E.jsonl 2>> $O/scope_E.err
FQTK_SOAK_SEEDS=${SOAK:-300} python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random" > $O/soak.log 2>&1; tail -2 $O/soak.log
python tools/soak_cli.py --iters ${SOAK_CLI:-60} --seed 11 > $O/soak_cli.log 2>&1; tail -1 $O/soak_cli.log
hipcc --offload-arch=gfx950 -O3 tools/stream_skeleton.hip -o /tmp/stream_skeleton 2>/dev/null && /tmp/stream_skeleton > $O/stream_skeleton.txt 2>&1
FQTK_SYNTH_PIUPAC=0.00063 python tools/kernel_times.py $O/kt_iupac1 -- --steps 10 --warmup 2 --cpu-seconds 0 --parity windows --no-scopes > $O/kernel_times_iupac1pct.csv 2>&1
bash tools/ab_ldsm_ablate.sh > $O/ab_ldsm_ablate.txt 2>&1
bash tools/pmc_insts.sh > $O/pmc_insts_cfg3.txt 2>&1
bash tools/pmc_cfg5.sh ${TAG}_pmc5 > $O/pmc5.log 2>&1
bash tools/pmc_memo.sh ${TAG}_pmc > $O/pmc_memo.log 2>&1; tail -30 $O/pmc_memo.log
ls $O
