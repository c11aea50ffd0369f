#!/bin/bash
# Round 5's measured artefacts in one go (run through gpurun): writes gpurun_out/<tag>/, from where the summaries are copied
# into profiles/.   usage: tools/round5_profile.sh r05_final
TAG=${1:-r05_final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
bash tools/profile_bench.sh ${TAG}_prof > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-300
cp gpurun_out/bench_detail.json $O/bench_detail_default.json 2>/dev/null
STEPS=10 bash tools/bench_matrix.sh > $O/bench_matrix.jsonl 2> /dev/null
STEPS=10 bash tools/bench_lens.sh > $O/bench_lens.jsonl 2> /dev/null
bash tools/bench_cliff.sh > $O/cliff.jsonl 2> /dev/null
for a in "384 24 1 2" "384 24 2 2" "384 32 1 2"; do echo "== bench_custom $a" >> $O/bench_custom.txt; timeout 300 python tools/bench_custom.py $a >> $O/bench_custom.txt 2>&1; done
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
for l in 1 6; do python tools/inflate_bench.py --level $l 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl; done
python tools/inflate_bench.py --const-qual 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl
for v in "" "--bgzf"; do FQTK_TIMING=1 timeout 300 python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads 16 $v >> $O/scope_E_64M.jsonl 2>> $O/scope_E.err; done
bash tools/gz_ab.sh ${TAG}_gz_ab 16000000 2 > /dev/null 2>&1; cp gpurun_out/${TAG}_gz_ab/summary.txt $O/gz_ab_summary.txt; cp gpurun_out/${TAG}_gz_ab/stretches.txt $O/gz_ab_stretches.txt
bash tools/gz_ab.sh ${TAG}_gz_ab_real 4000000 1 6 > /dev/null 2>&1; cp gpurun_out/${TAG}_gz_ab_real/summary.txt $O/gz_ab_real_gzip6_summary.txt
bash tools/kernel_stats_e.sh ${TAG}_kstats 16000000 . > $O/pipeline_plain_kernel_stats.txt 2>&1
ls $O
