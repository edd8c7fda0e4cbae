R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3f; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | cut -c1-250
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $OUT/bench_side.json 2> $OUT/bench_side.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_new*.json")):
    d=json.load(open(f)); print(f.split("bench_")[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["parity"]["live_rel_err_vs_fp32_oracle"])
d=json.load(open("$OUT/bench_side.json"))
print(json.dumps(d.get("side"), indent=None)[:3000])
PY
bash tools/gpu_r3_prof.sh cur default 2>&1 | grep -v amdgpu.ids
