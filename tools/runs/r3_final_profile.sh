#!/bin/bash
# End-of-round measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command per precision,
# PMC HBM traffic of the dominant kernel, codec kernel stats.  Outputs under gpurun_out/final_r3/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_r3
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for P in hybrid mixed half exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity --precision $P > $OUT/prof_$P.log 2>&1
  cp $(ls $OUT/prof_$P/*/*kernel_stats.csv | head -1) $OUT/bench_${P}_kernel_stats.csv
  rm -rf $OUT/prof_$P
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_codec -- python $R/tools/run_codec.py --batch 32 --precision exact --decode --iters 3 > $OUT/prof_codec.log 2>&1
cp $(ls $OUT/prof_codec/*/*kernel_stats.csv | head -1) $OUT/codec_kernel_stats.csv; rm -rf $OUT/prof_codec
cd $R
for P in hybrid mixed half; do tools/pmc_bench.sh $P > $OUT/pmc_$P.log 2>&1; done
python - <<PY
import json
out = {"note": None, "hbm_bytes_per_launch_by_precision": {}, "as_reported_by_precision": {}, "by_precision": {}}
for p in ("hybrid", "mixed", "half"):
    j = json.load(open("gpurun_out/pmc_traffic_%s.json" % p))
    out["note"] = j["note"]
    out["hbm_bytes_per_launch_by_precision"][p] = j.get("hbm_bytes_per_launch")
    out["as_reported_by_precision"][p] = j.get("hbm_bytes_per_launch_as_reported")
    out["by_precision"][p] = {"dominant_kernel": j.get("dominant_kernel"), "kernels": j["kernels"]}
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc_bench_hybrid gpurun_out/pmc_bench_mixed gpurun_out/pmc_bench_half
head -c 2500 $OUT/bench_default.json; echo; python -c "import json; j=json.load(open('$OUT/pmc_traffic.json')); print(j['hbm_bytes_per_launch_by_precision'], j['as_reported_by_precision'])"
tail -2 $OUT/prof_codec.log
