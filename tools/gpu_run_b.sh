#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
cd $R
python tools/bench_power.py > $OUT/power.txt 2>&1
for P in 4 2; do python tools/bench_gemm.py --prec $P --iters 30 > $OUT/gemm_p$P.txt 2>&1; done
cat $OUT/power.txt $OUT/gemm_p4.txt $OUT/gemm_p2.txt | grep -v amdgpu.ids
