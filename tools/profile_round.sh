#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + the two HBM PMC passes of `python bench.py`.
# Output under gpurun_out/prof_round/; summaries are then copied to profiles/ by hand (tools/pmc_traffic.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python $R/bench.py --no-cpu-baseline --no-sequence > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-sequence --no-full-head > $O/$c.log 2>&1
done
tail -2 $O/stats.log
find $O -name "*kernel_stats.csv" | head -1 | xargs head -12
