#!/bin/bash
# Round 6's measured artefacts in one go (run through gpurun, one box): writes gpurun_out/<tag>/, from where the summaries are copied into profiles/.
# usage: tools/round6_profile.sh r06_final [part ...]     parts: bench matrix decoder pipeline soak (default: all)
TAG=${1:-r06_final}; shift
PARTS=${@:-bench matrix decoder pipeline soak}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
has() { [[ " $PARTS " == *" $1 "* ]]; }
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
if has bench; then
  bash tools/profile_bench.sh ${TAG}_prof > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-300
  cp gpurun_out/${TAG}_prof/bench.json $O/bench_line.json; cp gpurun_out/bench_detail.json $O/bench_detail_default.json 2>/dev/null
  cp gpurun_out/${TAG}_prof/stats/bench_kernel_stats.csv $O/bench_cfg3_kernel_stats.csv 2>/dev/null; cp gpurun_out/${TAG}_prof/pmc_traffic.json $O/pmc_traffic_cfg3.json 2>/dev/null
  cp gpurun_out/${TAG}_prof/pmc_traffic_for_bench.json $O/ 2>/dev/null
  FQTK_BENCH_DEVICES=0,0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_line_devices00.json 2> $O/bench_devices00.err; cp gpurun_out/bench_detail.json $O/bench_detail_devices00.json 2>/dev/null
fi
if has matrix; then
  bash tools/bench_matrix.sh > $O/bench_matrix.jsonl 2> /dev/null
  bash tools/bench_cliff.sh > $O/cliff.jsonl 2> /dev/null
  echo "== bench_custom 384 24 1 2" > $O/bench_custom.txt; timeout 300 python tools/bench_custom.py 384 24 1 2 >> $O/bench_custom.txt 2>&1
fi
if has decoder; then
  python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
  for m in 8192 24576; do for a in "--level 1" "--level 6" "--const-qual"; do python tools/inflate_bench.py --members $m $a 2>/dev/null | tail -1 >> $O/inflate_kernel.jsonl; done; done
  MEMBERS=24576 bash tools/ab_inflate.sh "" "-DFQTK_INFLATE_LIT_BITS=10 -DFQTK_INFLATE_WAVES=5" "-DFQTK_INFLATE_WAVES=5" "-DFQTK_INFLATE_WAVES=7" > $O/ab_inflate.txt 2>&1
  bash tools/pmc_inflate.sh ${TAG}_pmc_inflate > $O/pmc_inflate_kernel.txt 2>&1
fi
if has pipeline; then
  for k in plain bgzf gz; do bash tools/profile_pipeline_kind.sh ${TAG}_pipe $k 64000000 > $O/pipeline_$k.txt 2>&1; cp gpurun_out/${TAG}_pipe/${k}_kernel_stats.csv $O/pipeline_${k}_inputs_kernel_stats.csv 2>/dev/null; done
fi
if has soak; then
  timeout 1500 python tools/soak_cli.py --iters ${SOAK_ITERS:-160} --seed 6 > $O/soak_cli.log 2>&1; tail -3 $O/soak_cli.log
fi
ls $O
