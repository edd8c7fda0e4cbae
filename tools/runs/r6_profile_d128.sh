#!/bin/bash
# Round 6: rocprofv3 kernel stats of the dim-128 / depth-6 step (BASELINE config 2, 32 x 1024), hybrid plan
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_d128
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python $R/bench.py --dim 128 --depth 6 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $OUT/bench.json 2> $OUT/bench.err < /dev/null
f=$(ls $OUT/p/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/d128_kernel_stats.csv
rm -rf $OUT/p
cd $R
python - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/prof_d128/d128_kernel_stats.csv")))
for r in rows[:22]:
    print(f"  {r['Name'][:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/12e6:7.3f} ms/step")
d = json.loads(open("gpurun_out/prof_d128/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
