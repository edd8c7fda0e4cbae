#!/bin/bash
# Round-2 GPU pass O: tap-shared conv path -- kernel tests, model goldens, bench A/B against the build without it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2o
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
OLD=$R/naturalspeech2_pytorch_amd/libns2hip_g2_noconv3.so
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "conv or linear" 2>&1 | tail -12 ) > $OUT/pytest_kernels.log 2>&1
tail -5 $OUT/pytest_kernels.log
if grep -q "failed" $OUT/pytest_kernels.log; then echo "KERNEL TESTS FAILED"; exit 0; fi
( time timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 ) > $OUT/pytest_model.log 2>&1
tail -3 $OUT/pytest_model.log
for i in 1 2; do
for P in hybrid half mixed; do
  python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_new_${P}_$i.json 2>/dev/null
  NS2_LIB=$OLD python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_old_${P}_$i.json 2>/dev/null
done
done
python - <<PY
import json
for i in (1,2):
  for P in ("hybrid","half","mixed"):
    for n in ("new","old"):
        d=json.load(open("$OUT/bench_%s_%s_%d.json"%(n,P,i))); print(i, P, n, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["parity"]["live_rel_err_vs_fp32_oracle"])
PY
