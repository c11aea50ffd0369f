#!/bin/bash
# Round-3 artefacts in one go (run through gpurun): writes gpurun_out/r03/, from where the summaries go to profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
STEPS=10 bash tools/bench_matrix.sh > $O/bench_matrix.jsonl 2> /dev/null
STEPS=10 bash tools/bench_lens.sh > $O/bench_lens.jsonl 2>/dev/null
bash tools/bench_cliff.sh > $O/cliff.jsonl 2>&1
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
bash tools/bgzf_phases.sh > $O/bgzf_phases.txt 2>&1
for t in 16 8 32; do python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads $t >> $O/scope_E.jsonl 2>> $O/scope_E.err; done
python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --extra=--host-output >> $O/scope_E_host.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --bgzf >> $O/scope_E_bgzf_inputs.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --extra="--compression-level 0" >> $O/scope_E_level0.jsonl 2>> $O/scope_E.err
python tools/soak_cli.py --iters ${SOAK_CLI:-60} --seed 11 > $O/soak_cli.log 2>&1; tail -1 $O/soak_cli.log
for m in seq par both; do python tools/startup_probe.py $m; done > $O/startup_probe.txt 2>&1
ls $O
