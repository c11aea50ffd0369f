O=$GRAFT_REPO_ROOT/gpurun_out/r02n; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
bash tools/bgzf_phases.sh > $O/bgzf_phases.txt 2>&1
for v in "--threads 16" "--threads 16 --extra=--gpu-bgzf" "--threads 8" "--threads 8 --extra=--gpu-bgzf" "--threads 32" "--threads 32 --extra=--gpu-bgzf"; do FQTK_TIMING=1 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block $v >> $O/scope_E.jsonl 2>> $O/scope_E.err; done
FQTK_TIMING=1 python tools/scope_bench.py --skip-b --templates 48000000 --repeat-block --threads 16 --extra=--gpu-bgzf >> $O/scope_E.jsonl 2>> $O/scope_E.err
for v in "" "--extra=--gpu-bgzf"; do python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz $v >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err; done
FQTK_ZLIB_INFLATE=1 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 8000000 --repeat-block --threads 16 --bgzf >> $O/scope_E.jsonl 2>> $O/scope_E.err
python tools/scope_bench.py --skip-b --templates 8000000 --repeat-block --threads 16 --bgzf --extra=--gpu-bgzf >> $O/scope_E.jsonl 2>> $O/scope_E.err
python tools/soak_cli.py --iters 40 --seed 21 > $O/soak_cli.log 2>&1; tail -1 $O/soak_cli.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
tail -3 $O/pytest_gpu.log
