#!/bin/bash
# experiment A/B: $1 = tag of the comparison library libns2hip_g2_<tag>.so; kernel conv/linear tests on the default build, then bench pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2p
rm -rf $OUT; mkdir -p $OUT
cd $R
OLD=$R/naturalspeech2_pytorch_amd/libns2hip_g2_$1.so
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "${2:-conv or linear}" 2>&1 | tail -4 ) > $OUT/pytest_kernels.log 2>&1
tail -2 $OUT/pytest_kernels.log
if grep -q "failed" $OUT/pytest_kernels.log; then echo "KERNEL TESTS FAILED"; exit 0; fi
for i in 1 2; do
for P in ${3:-hybrid mixed}; do
  python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_new_${P}_$i.json 2>/dev/null
  NS2_LIB=$OLD python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline --precision $P > $OUT/bench_old_${P}_$i.json 2>/dev/null
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    d=json.load(open(f)); print(f.split("bench_")[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], list(d["parity"]["live_rel_err_vs_fp32_oracle"].values()))
PY
