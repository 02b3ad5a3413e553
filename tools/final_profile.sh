# The round's committed measurements in one gpurun call (run from the repo root on the GPU box):
#   bench line, rocprofv3 kernel stats + the two HBM PMC passes of the same command, the configs[4] line, full-head kernel stats.
# Afterwards, in the build container:  bash tools/collect_round_profiles.sh round6   (copies the summaries into profiles/)
# This is the LAST measurement command of a round: tests/test_host_logic.py::test_committed_roofline_records_match_the_kernel_source holds the tree
# to the source hash these passes record.
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_round gpurun_out/${ROUND:-round6}_pmc_hbm_traffic > gpurun_out/pmc_traffic.log 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_cfg5.json 2>/dev/null
bash tools/prof_full_head.sh > gpurun_out/prof_full_head.log 2>&1
find gpurun_out/prof_round -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
python tools/north_trace_split.py gpurun_out/prof_round/stats > gpurun_out/north_trace_split.txt 2>&1     # in-step against back-to-back launches of the roofline kernel, from the same trace
find gpurun_out/prof_round gpurun_out/prof_full -name "*kernel_trace.csv" -delete     # (the traces exceed what gpurun copies back)
tail -c 400 gpurun_out/bench_final.json
