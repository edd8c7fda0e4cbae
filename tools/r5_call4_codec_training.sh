#!/bin/bash
# Round 5, GPU call 4: (1) parity of the fused 32-channel residual block (kernel vs HF's block; the whole SEANet vs HF; codes), of the
# streaming attention-delta kernel and of the multi-tensor unscale; (2) codec timing with and without the fused block, alternating;
# its kernel timeline; (3) warm training step, exact vs mixed, after the delta rewrite (+ kernel stats of the mixed step).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short -k "resblock or mixed_attention or d64_L2 or d128_L6" 2>&1 | tail -n 12 ) > $O/t_round5.txt
( timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -k "attention or d64 or conditioned or ragged" 2>&1 | tail -n 8 ) > $O/t_backward.txt
( timeout 900 python -m pytest tests/test_parity_r2_gpu.py -q -m gpu --tb=short -k "seanet or codec or encodec" -s 2>&1 | tail -n 14 ) > $O/t_seanet.txt
for rep in 1 2 3; do
  timeout 300 python tools/run_codec.py --decode --iters 5 > $O/codec_new_$rep.txt 2>&1
  NS2_SEANET_NARROW_RESBLOCK=0 timeout 300 python tools/run_codec.py --decode --iters 5 > $O/codec_old_$rep.txt 2>&1
done
timeout 900 python tools/bench_train.py --shapes d512 --backends hip --train-precision exact,mixed --iters 5 --fused-adam --out $O/train_d512.json > $O/train_d512.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_codec -- python $R/tools/run_codec.py --decode --iters 3 > $R/$O/prof_codec.log 2>&1
cp $(ls $R/$O/prof_codec/*/*kernel_stats.csv | head -1) $R/$O/codec_kernel_stats.csv; rm -rf $R/$O/prof_codec
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train -- python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision mixed --iters 3 --fused-adam > $R/$O/prof_train.log 2>&1
cp $(ls $R/$O/prof_train/*/*kernel_stats.csv | head -1) $R/$O/train_d512_mixed_kernel_stats.csv; rm -rf $R/$O/prof_train
cd $R
for f in t_round5 t_backward t_seanet; do echo "== $f"; tail -n 6 $O/$f.txt | cut -c1-260; done
grep -h "^ok codes" $O/codec_*.txt
grep -h ms_per_step $O/train_d512.txt | cut -c1-230
head -n 14 $O/codec_kernel_stats.csv | cut -c1-150
