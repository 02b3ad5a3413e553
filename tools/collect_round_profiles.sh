#!/bin/bash
# After `gpurun -- bash tools/final_profile.sh` returned: copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked).
#   bash tools/collect_round_profiles.sh round6
set -e
cd "$(dirname "$0")/.."
r=${1:?round tag, e.g. round6}
cp gpurun_out/${r}_pmc_hbm_traffic.json profiles/${r}_pmc_hbm_traffic.json
cp gpurun_out/${r}_pmc_hbm_traffic.txt profiles/${r}_pmc_hbm_traffic.txt
cp gpurun_out/kernel_stats.csv profiles/${r}_kernel_stats.csv
cp gpurun_out/prof_full/kernel_stats.csv profiles/${r}_kernel_stats_full_head.csv
cp gpurun_out/bench_final.json profiles/${r}_bench_line.json
cp gpurun_out/bench_cfg5.json profiles/${r}_bench_line_config5.json
cp gpurun_out/north_trace_split.txt profiles/${r}_north_trace_split.txt
ls -la profiles/${r}_*
