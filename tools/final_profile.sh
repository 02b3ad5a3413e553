python bench.py > gpurun_out/bench_final_r3.json 2> gpurun_out/bench_final_r3.err
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round
rocprofv3 --kernel-trace --stats -d $O/stats_par --output-format csv -- python $R/bench.py --no-cpu-baseline --head-stream parallel > $O/stats_par.log 2>&1
cd $R; bash tools/prof_full_head.sh > gpurun_out/prof_full_head.log 2>&1
tail -c 400 gpurun_out/bench_final_r3.json
