#!/bin/bash
# End-of-round measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command per precision,
# PMC HBM traffic of the dominant kernel, kernel stats of a warm training step.  Outputs under gpurun_out/final_r4/ (copy what
# should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_r4
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for P in hybrid hybrid_ff mixed half exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity --precision $P > $OUT/prof_$P.log 2>&1
  cp $(ls $OUT/prof_$P/*/*kernel_stats.csv | head -1) $OUT/bench_${P}_kernel_stats.csv
  rm -rf $OUT/prof_$P
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d128 -- python $R/bench.py --dim 128 --depth 6 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $OUT/prof_d128.log 2>&1
cp $(ls $OUT/prof_d128/*/*kernel_stats.csv | head -1) $OUT/bench_d128_hybrid_kernel_stats.csv; rm -rf $OUT/prof_d128
for shp in d512 d128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$shp -- python $R/tools/bench_train.py --shapes $shp --backends hip --iters 3 > $OUT/prof_train_$shp.log 2>&1
  cp $(ls $OUT/prof_train_$shp/*/*kernel_stats.csv | head -1) $OUT/train_${shp}_kernel_stats.csv
  rm -rf $OUT/prof_train_$shp
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-side --no-parity > $OUT/prof_b1.log 2>&1
cp $(ls $OUT/prof_b1/*/*kernel_stats.csv | head -1) $OUT/bench_b1_hybrid_kernel_stats.csv; rm -rf $OUT/prof_b1
cd $R
python tools/time_config1.py > $OUT/time_config1.json 2> $OUT/time_config1.err
python tools/exp_small_m_kernel.py > $OUT/small_batch.json 2> /dev/null
for P in hybrid; do tools/pmc_bench.sh $P > $OUT/pmc_$P.log 2>&1; cp gpurun_out/pmc_traffic_$P.json $OUT/; done
rm -rf gpurun_out/pmc_bench_hybrid
head -c 3000 $OUT/bench_default.json; echo
