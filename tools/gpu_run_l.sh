#!/bin/bash
# experiment: wavenet dilated convs without correction terms (NS2_WN_HALF=1) -- speed and live parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2l
rm -rf $OUT; mkdir -p $OUT
cd $R
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/base$i.json 2>/dev/null
NS2_WN_HALF=1 python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/wnhalf$i.json 2>/dev/null
done
python - <<PY
import json
for n in ("base1","wnhalf1","base2","wnhalf2"):
    d=json.load(open("$OUT/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["parity"]["live_rel_err_vs_fp32_oracle"])
PY
