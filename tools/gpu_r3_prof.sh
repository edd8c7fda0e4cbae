#!/bin/bash
# rocprofv3 kernel stats of the hybrid bench for a list of library builds: tools/gpu_r3_prof.sh <tag> [lib.so] ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  TAG=$1; LIB=$2; shift; shift
  if [ "$LIB" != "default" ]; then export NS2_LIB=$R/naturalspeech2_pytorch_amd/$LIB; else unset NS2_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity --precision ${PREC:-hybrid} > $OUT/prof_$TAG.log 2>&1
  cp $(ls $OUT/prof_$TAG/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
  rm -rf $OUT/prof_$TAG
  echo "== $TAG"; python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  total/step {float(r["TotalDurationNs"])/7/1e6:7.3f} ms  {float(r["Percentage"]):5.1f}%')
print("sum per step (7 steps)", tot/7/1e6, "ms")
PY
done
