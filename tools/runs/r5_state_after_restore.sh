#!/bin/bash
# Round 5, GPU call 6 (first call after the container was re-created: the outputs of calls 4 / 5 were lost with it): backward + round-5
# tests on the committed tree (tplanes on 32-bit LDS words, wgrad split plan, streaming attention delta, fused 32-channel residual
# block), warm training step in both arithmetics, codec A/B, kernel stats of the hybrid step / the training steps / one encode.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5f; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_round5_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -n 15 ) > $O/t_backward_round5.txt
( timeout 600 python -m pytest tests/test_parity_r2_gpu.py -q -m gpu --tb=short -k "seanet or codec or encodec" 2>&1 | tail -n 8 ) > $O/t_seanet.txt
timeout 900 python tools/bench_train.py --shapes d512,d128 --backends hip --train-precision exact,mixed --iters 5 --fused-adam --out $O/train.json > $O/train.txt 2>&1
for rep in 1 2; do
  timeout 300 python tools/run_codec.py --decode --iters 5 > $O/codec_new_$rep.txt 2>&1
  NS2_SEANET_NARROW_RESBLOCK=0 timeout 300 python tools/run_codec.py --decode --iters 5 > $O/codec_old_$rep.txt 2>&1
done
timeout 600 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/bench_hybrid_short.json 2> $O/bench_hybrid_short.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
prof() { # name, command...
  local n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$n -- "$@" > $R/$O/prof_$n.log 2>&1
  cp $(ls $R/$O/prof_$n/*/*kernel_stats.csv | head -1) $R/$O/${n}_kernel_stats.csv; rm -rf $R/$O/prof_$n
}
prof bench_hybrid python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity
prof train_d512_mixed python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision mixed --iters 3 --fused-adam
prof train_d512_exact python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision exact --iters 3 --fused-adam
prof codec python $R/tools/run_codec.py --decode --iters 3
cd $R
cp gpurun_out/parity_r5.json $O/ 2>/dev/null
for f in t_backward_round5 t_seanet; do echo "== $f"; tail -n 8 $O/$f.txt | cut -c1-260; done
grep -h ms_per_step $O/train.txt | cut -c1-260
grep -h "^ok codes" $O/codec_*.txt
head -c 1500 $O/bench_hybrid_short.json; echo
head -n 12 $O/train_d512_mixed_kernel_stats.csv | cut -c1-160
head -n 14 $O/codec_kernel_stats.csv | cut -c1-160
