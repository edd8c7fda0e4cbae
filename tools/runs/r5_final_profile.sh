#!/bin/bash
# End-of-round-5 measurement on one GPU box: the default bench line, rocprofv3 kernel stats of the hybrid step / the dim-128 step / a warm
# training step in both arithmetics / one codec encode + decode.  Outputs under gpurun_out/final_r5/ (copied into profiles/ as r05_*).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_r5
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
prof() { n=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -- "$@" > $OUT/prof_$n.log 2>&1; cp $(ls $OUT/prof_$n/*/*kernel_stats.csv | head -1) $OUT/${n}_kernel_stats.csv; rm -rf $OUT/prof_$n; }
prof bench_hybrid python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity
prof bench_d128_hybrid python $R/bench.py --dim 128 --depth 6 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity
prof train_d512_mixed python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision mixed --iters 3 --fused-adam
prof train_d512_exact python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision exact --iters 3 --fused-adam
prof codec python $R/tools/run_codec.py --decode --iters 3
cd $R
python tools/time_config1.py > $OUT/time_config1.json 2> $OUT/time_config1.err
head -c 1500 $OUT/bench_default.json; echo
