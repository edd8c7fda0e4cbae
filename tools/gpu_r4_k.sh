#!/bin/bash
# skinny linear narrow launch shape: tests + training step before/after (the same box: NS2 lib is the new one; "before" numbers are profiles/r04_train_*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4k; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "skinny or time_embed" > $O/t_skinny.txt 2>&1; echo "skinny rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_skinny.txt
timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=short -x > $O/t_bwd.txt 2>&1; echo "backward+round4 rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_bwd.txt
python tools/bench_train.py --shapes d512 d128 --backends hip --iters 5 2>/dev/null | tee $O/train.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/tools/bench_train.py --shapes d512 --backends hip --iters 3 > $O/prof_train.log 2>&1
cp $(ls $O/prof_train/*/*kernel_stats.csv | head -1) $O/train_d512_kernel_stats.csv; rm -rf $O/prof_train
head -14 $O/train_d512_kernel_stats.csv | cut -c1-130
