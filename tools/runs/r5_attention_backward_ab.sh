#!/bin/bash
# Round 5: attention backward on LDS-DMA double-buffered row-major tiles with LDS transpose reads (attn_bwd2_kernel) against the
# round-4 kernel on transposed copies (NS2_ATTN_BWD_V1=1): kernel test in both, every gradient vs the reference autograd, warm
# training step alternating, kernel stats of the new step.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -x -k "attention" 2>&1 | tail -n 12 ) > $O/t_attention_v2.txt
( NS2_ATTN_BWD_V1=1 timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -x -k "attention" 2>&1 | tail -n 4 ) > $O/t_attention_v1.txt
( timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_round5_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 8 ) > $O/t_backward_round5.txt
for rep in 1 2; do
  timeout 600 python tools/bench_train.py --shapes d512 --backends hip --train-precision mixed,exact --iters 5 --fused-adam > $O/train_v2_$rep.txt 2>&1
  NS2_ATTN_BWD_V1=1 timeout 600 python tools/bench_train.py --shapes d512 --backends hip --train-precision mixed,exact --iters 5 --fused-adam > $O/train_v1_$rep.txt 2>&1
done
timeout 600 python tools/bench_train.py --shapes d128 --backends hip --train-precision mixed,exact --iters 5 --fused-adam > $O/train_d128.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision mixed --iters 3 --fused-adam > $R/$O/prof.log 2>&1
cp $(ls $R/$O/prof/*/*kernel_stats.csv | head -1) $R/$O/train_d512_mixed_kernel_stats.csv; rm -rf $R/$O/prof
cd $R
for f in t_attention_v2 t_attention_v1 t_backward_round5; do echo "== $f"; tail -n 5 $O/$f.txt | cut -c1-200; done
for f in $O/train_v*.txt $O/train_d128.txt; do echo $f; grep -h ms_per_step $f | cut -c1-200; done
head -n 12 $O/train_d512_mixed_kernel_stats.csv | cut -c1-150
