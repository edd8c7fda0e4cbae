R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3codec; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in exact mixed; do
python $R/tools/run_codec.py --batch 32 --precision $P --decode 2>&1 | grep -v amdgpu | tail -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/tools/run_codec.py --batch 32 --precision exact --iters 3 > $OUT/prof.log 2>&1
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/codec_kernel_stats.csv; rm -rf $OUT/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/codec_kernel_stats.csv")))
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/4/1e6:8.3f} ms/encode {float(r["Percentage"]):5.1f}%')
PY
