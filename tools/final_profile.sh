#!/bin/bash
# End-of-round measurement on the GPU box: bench lines, rocprofv3 kernel stats of the same commands, PMC HBM traffic.
# Outputs under gpurun_out/final/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_half.json 2> $OUT/bench_half.err
python bench.py --steps 20 --warmup 3 --precision exact --no-cpu-baseline > $OUT/bench_exact.json 2> $OUT/bench_exact.err
python bench.py --steps 20 --warmup 3 --precision fast --no-cpu-baseline > $OUT/bench_fast.json 2> $OUT/bench_fast.err
python tools/bench_rvq.py > $OUT/bench_rvq.json 2> $OUT/bench_rvq.err
python tools/bench_power.py > $OUT/bench_power.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for P in half exact fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --precision $P > $OUT/prof_$P.log 2>&1
  cp $(ls $OUT/prof_$P/*/*kernel_stats.csv | head -1) $OUT/bench_${P}_kernel_stats.csv
done
cd $R
for P in half exact; do tools/pmc_bench.sh $P > $OUT/pmc_$P.log 2>&1; done
python - <<PY
import json
out = {"note": None, "hbm_bytes_per_launch_by_precision": {}, "by_precision": {}}
for p in ("half", "exact"):
    j = json.load(open("gpurun_out/pmc_traffic_%s.json" % p))
    out["note"] = j["note"]
    out["hbm_bytes_per_launch_by_precision"][p] = j.get("hbm_bytes_per_launch")
    out["by_precision"][p] = {"dominant_kernel": j.get("dominant_kernel"), "kernels": j["kernels"]}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/prof_exact $OUT/prof_fast $OUT/prof_half
head -c 900 $OUT/bench_half.json; echo; head -c 300 $OUT/bench_exact.json; echo; head -c 300 $OUT/bench_fast.json; echo; cat $OUT/bench_rvq.json | head -c 400; echo; cat $OUT/bench_power.txt | grep prec=
