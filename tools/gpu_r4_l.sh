#!/bin/bash
# kernel stats of small-batch steps (B = 1 x 1024): where does a latency-bound step go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4l; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "128 6" "512 12"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -- python $R/bench.py --dim $1 --depth $2 --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-side --no-parity > $O/prof_$1.log 2>&1
  cp $(ls $O/prof_$1/*/*kernel_stats.csv | head -1) $O/b1_d$1_kernel_stats.csv; rm -rf $O/prof_$1
  tail -1 $O/prof_$1.log | cut -c1-300
  head -22 $O/b1_d$1_kernel_stats.csv | cut -c1-150
done
