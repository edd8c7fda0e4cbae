#!/bin/bash
# Round 6: rocprofv3 kernel stats of the hybrid step with and without the round-6 kernels (NS2_GEMM=4), same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { n=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -- "$@" > $OUT/prof_$n.log 2>&1; cp $(ls $OUT/prof_$n/*/*kernel_stats.csv | head -1) $OUT/${n}_kernel_stats.csv; rm -rf $OUT/prof_$n; }
prof bench_hybrid python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity
NS2_GEMM=4 prof bench_hybrid_round5_kernels python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity
cd $R
python - <<'PY'
import csv, os
for n in ("bench_hybrid", "bench_hybrid_round5_kernels"):
    rows = list(csv.DictReader(open(f"gpurun_out/prof_r6/{n}_kernel_stats.csv")))
    print("==", n)
    for r in rows[:14]:
        print(f"  {r['Name'][:90]:90s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/7e6:7.3f} ms/step")
PY
