#!/bin/bash
# Round 5: weight gradients from the token-major planes (gemm2.hip TR, ns2_wgrad_rows): kernel test, every gradient of the model in
# both training arithmetics against the reference autograd, the warm training step and its kernel stats.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short -x -k "wgrad_from_row_planes" 2>&1 | tail -n 25 ) > $O/t_wgrad_rows.txt
( timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_round5_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 15 ) > $O/t_backward_round5.txt
timeout 900 python tools/bench_train.py --shapes d512,d128 --backends hip --train-precision exact,mixed --iters 5 --fused-adam --out $O/train.json > $O/train.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for tp in mixed exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$tp -- python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision $tp --iters 3 --fused-adam > $R/$O/prof_$tp.log 2>&1
  cp $(ls $R/$O/prof_$tp/*/*kernel_stats.csv | head -1) $R/$O/train_d512_${tp}_kernel_stats.csv; rm -rf $R/$O/prof_$tp
done
cd $R
cp gpurun_out/parity_r5.json $O/ 2>/dev/null
for f in t_wgrad_rows t_backward_round5; do echo "== $f"; tail -n 12 $O/$f.txt | cut -c1-300; done
grep -h ms_per_step $O/train.txt | cut -c1-260
head -n 14 $O/train_d512_mixed_kernel_stats.csv | cut -c1-160
