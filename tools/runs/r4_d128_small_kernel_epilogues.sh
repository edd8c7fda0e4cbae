#!/bin/bash
# 128x128 kernel: fast plane epilogues + the hybrid plan's half-product phase -- kernel tests, d128 goldens, d128 kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4j; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x > $O/t_kernels.txt 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_kernels.txt
timeout 1200 python -m pytest tests -q -m gpu --tb=short -k "d128 or golden or encoder or seanet" > $O/t_d128.txt 2>&1; echo "d128/golden/encoders rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_d128.txt
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --dim 128 --depth 6 --no-side --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/bench_d128.txt; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d128 -- python $R/bench.py --dim 128 --depth 6 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $O/prof_d128.log 2>&1
cp $(ls $O/prof_d128/*/*kernel_stats.csv | head -1) $O/bench_d128_hybrid_kernel_stats.csv; rm -rf $O/prof_d128
head -12 $O/bench_d128_hybrid_kernel_stats.csv | cut -c1-120
