#!/bin/bash
# Round-2 GPU pass M: refactored gemm2 (mode-tagged K loop, Wavenet phase-1 half product in the hybrid plan): every kernel
# test, model suites, the hybrid parity sweeps, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2m
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --ignore=tests/test_parity_r2_gpu.py 2>&1 | tail -15 ) > $OUT/pytest_old.log 2>&1
tail -4 $OUT/pytest_old.log
( time timeout 1800 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "golden or sweep or conditioned or ddim_trajectory or stress or bench" 2>&1 | tail -25 ) > $OUT/pytest_parity.log 2>&1
tail -6 $OUT/pytest_parity.log
python bench.py --steps 20 --warmup 3 --no-side --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], {k: d[k]["value"] for k in ("mixed_mode","half_mode","exact_mode")}, d["parity"]["live_rel_err_vs_fp32_oracle"])
p=json.load(open("gpurun_out/parity_r2.json"))
print({k: v["max"] for k, v in p.items() if k.startswith("sweep")}, {k: v for k, v in p.items() if k.startswith("conditioned")})
PY
