#!/bin/bash
# A/B of the persistent-tile gemm2 (libns2hip.so) against the previous kernel (libns2hip_old.so) on the same box + correctness.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -30 ) > $OUT/pytest.log 2>&1
for rep in 1 2; do
for L in new old; do
  if [ $L = old ]; then export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_old.so; else unset NS2_LIB; fi
  for P in 4 2; do python tools/bench_gemm.py --prec $P --iters 30 >> $OUT/gemm_${L}_p$P.txt 2>&1; done
done
done
unset NS2_LIB
for P in mixed half; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-side --precision $P > $OUT/bench_$P.json 2> $OUT/bench_$P.err; done
export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_old.so
for P in mixed half; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-side --precision $P > $OUT/bench_old_$P.json 2> $OUT/bench_old_$P.err; done
tail -8 $OUT/pytest.log
for L in new old; do for P in 4 2; do echo "== $L p$P"; grep -v amdgpu $OUT/gemm_${L}_p$P.txt; done; done
for f in bench_mixed bench_old_mixed bench_half bench_old_half; do python - <<PY
import json
try:
    j=json.loads([l for l in open("$OUT/$f.json") if l.startswith("{")][0]); print("$f", j["value"], j["ms_per_step"], j["roofline"]["achieved"])
except Exception as e: print("$f", "ERR", e); print(open("$OUT/$f.err").read()[-1500:])
PY
done
