#!/bin/bash
# small batches (split-K + 128x128 kernel dispatch): kernel tests incl. the split-K fixture variant, the model-level split-K test,
# ms per step at 1 ... 32 utterances, kernel stats of a 1 x 1024 step.  -> profiles/r04_small_batch_split_k.json,
# profiles/r04_bench_b1_hybrid_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short 2>&1 | tail -3
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "seanet or codec or split_k" 2>&1 | tail -3
python tools/exp_small_m_kernel.py 2>/dev/null | tee $O/small_batch.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-side --no-parity > $O/prof_b1.log 2>&1
cp $(ls $O/prof_b1/*/*kernel_stats.csv | head -1) $O/bench_b1_hybrid_kernel_stats.csv; rm -rf $O/prof_b1
grep -E "finish|gemm_kernel" $O/bench_b1_hybrid_kernel_stats.csv | cut -c1-140
