#!/bin/bash
# round 3, pass A: diagnostics. Block timelines of the GEMM family (G2_BLKTRACE build), 128x128 kernel on the same shapes, baseline bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
cd $R
export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_g2_blk.so
timeout 300 python tools/trace_blocks.py --prec 4 --out $OUT/blocks_p4.json > $OUT/blocks_p4.txt 2>&1
timeout 200 python tools/trace_blocks.py --prec 2 --which ffconv,ffin,qkv --out $OUT/blocks_p2.json > $OUT/blocks_p2.txt 2>&1
timeout 200 python tools/trace_blocks.py --prec 5 --which wavenet --out $OUT/blocks_p5.json > $OUT/blocks_p5.txt 2>&1
timeout 200 python tools/trace_blocks.py --prec 3 --which ffconv,qkv,outproj --out $OUT/blocks_p3.json > $OUT/blocks_p3.txt 2>&1
unset NS2_LIB
cat $OUT/blocks_p4.txt $OUT/blocks_p2.txt $OUT/blocks_p5.txt $OUT/blocks_p3.txt
for K in 0 1; do
  timeout 200 python tools/bench_gemm.py --prec 4 --kernel $K --which all > $OUT/gemm_p4_k$K.txt 2>&1
  timeout 200 python tools/bench_gemm.py --prec 2 --kernel $K --which all > $OUT/gemm_p2_k$K.txt 2>&1
done
cat $OUT/gemm_p4_k0.txt $OUT/gemm_p4_k1.txt $OUT/gemm_p2_k0.txt $OUT/gemm_p2_k1.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $OUT/bench_base.json 2> $OUT/bench_base.err
cat $OUT/bench_base.json
