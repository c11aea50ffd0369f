R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
python tools/bgzf_bench.py > $O/bgzf_kernel.json 2>/dev/null
python tools/bgzf_bench.py --const-qual > $O/bgzf_kernel_constq.json 2>/dev/null
bash tools/bgzf_phases.sh > $O/bgzf_phases.txt 2>&1
bash tools/pmc_bgzf.sh > $O/bgzf_pmc.txt 2>&1
for t in 16 32; do timeout 300 python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads $t >> $O/scope_E.jsonl 2>> $O/scope_E.err; done
timeout 300 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --gz >> $O/scope_E_gz.jsonl 2>> $O/scope_E.err
timeout 300 python tools/scope_bench.py --skip-b --templates 16000000 --repeat-block --threads 16 --bgzf >> $O/scope_E_bgzf_inputs.jsonl 2>> $O/scope_E.err
timeout 900 bash tools/profile_pipeline.sh r03b_pipe 16000000 > $O/profile_pipeline.log 2>&1
ls $O $R/gpurun_out/r03b_pipe | head -40
