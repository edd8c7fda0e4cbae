#!/bin/bash
# Round-2 GPU pass N: stage-ring K loop of the single-product GEMM -- kernel tests, per-shape A/B against the two-stage build,
# bench of the hybrid and half modes with both builds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2n
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
OLD=$R/naturalspeech2_pytorch_amd/libns2hip_g2_noring.so
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -12 ) > $OUT/pytest_kernels.log 2>&1
tail -5 $OUT/pytest_kernels.log
for i in 1 2; do
  python tools/bench_gemm.py --precision 2 > $OUT/gemm_ring_$i.txt 2>&1
  NS2_LIB=$OLD python tools/bench_gemm.py --precision 2 > $OUT/gemm_noring_$i.txt 2>&1
done
grep -h "prec=2" $OUT/gemm_ring_1.txt | cut -c1-110; echo ---; grep -h "prec=2" $OUT/gemm_noring_1.txt | cut -c1-110; echo --- run 2; grep -h "prec=2" $OUT/gemm_ring_2.txt | cut -c1-110; echo ---; grep -h "prec=2" $OUT/gemm_noring_2.txt | cut -c1-110
for P in hybrid half; do
  python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_ring_$P.json 2>/dev/null
  NS2_LIB=$OLD python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_noring_$P.json 2>/dev/null
done
python - <<PY
import json
for n in ("ring_hybrid","noring_hybrid","ring_half","noring_half"):
    d=json.load(open("$OUT/bench_%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["parity"]["live_rel_err_vs_fp32_oracle"])
PY
