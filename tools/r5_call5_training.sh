#!/bin/bash
# Round 5, GPU call 5: the rewritten tplanes kernel (32-bit LDS words) and the wgrad split plan under the backward tests (both
# training arithmetics), then the warm training step and its kernel stats.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 12 ) > $O/t_backward.txt
( timeout 1200 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 12 ) > $O/t_round5.txt
timeout 900 python tools/bench_train.py --shapes d512,d128 --backends hip --train-precision exact,mixed --iters 5 --fused-adam --out $O/train.json > $O/train.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for tp in mixed exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train_$tp -- python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision $tp --iters 3 --fused-adam > $R/$O/prof_train_$tp.log 2>&1
  cp $(ls $R/$O/prof_train_$tp/*/*kernel_stats.csv | head -1) $R/$O/train_d512_${tp}_kernel_stats.csv; rm -rf $R/$O/prof_train_$tp
done
cd $R
cp gpurun_out/parity_r5.json $O/ 2>/dev/null
for f in t_backward t_round5; do echo "== $f"; tail -n 6 $O/$f.txt | cut -c1-260; done
grep -h ms_per_step $O/train.txt | cut -c1-230
head -n 12 $O/train_d512_mixed_kernel_stats.csv | cut -c1-150
