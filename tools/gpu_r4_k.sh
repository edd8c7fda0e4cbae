#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d128 -- python bench.py --dim 128 --depth 6 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $O/prof_d128.log 2>&1
cp $(ls $O/prof_d128/*/*kernel_stats.csv | head -1) $O/bench_d128_hybrid_kernel_stats.csv; rm -rf $O/prof_d128
tail -2 $O/prof_d128.log | cut -c1-300
