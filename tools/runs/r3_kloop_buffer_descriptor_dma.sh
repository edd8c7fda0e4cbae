R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT
cd $R
NEW=$R/naturalspeech2_pytorch_amd/libns2hip_g2_buf.so
for P in 4 2 3; do
  timeout 200 python tools/ab_epilogue.py --prec $P > $OUT/ab_old_p$P.txt 2>&1
  for rep in 1 2; do
    NS2_LIB=$NEW timeout 200 python tools/ab_epilogue.py --prec $P > $OUT/ab_new_p${P}_$rep.txt 2>&1
    echo "== prec $P rep $rep"; diff $OUT/ab_new_p${P}_$rep.txt $OUT/ab_old_p$P.txt && echo identical
  done
done
( NS2_LIB=$NEW timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -4 ) | cut -c1-200
for L in blk blkbuf; do
echo "== $L"
export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_g2_$L.so
timeout 300 python tools/trace_blocks.py --prec 4 --which qkv,ffin,outproj,ffout 2>&1 | grep -v amdgpu
timeout 200 python tools/trace_blocks.py --prec 2 --which ffin,qkv,ffout 2>&1 | grep -v amdgpu
done
unset NS2_LIB
for i in 1 2; do
NS2_LIB=$NEW timeout 300 python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/bench_old_$i.json 2> $OUT/bench_old_$i.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("bench_")[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["parity"]["live_rel_err_vs_fp32_oracle"])
    except Exception as e: print(f, "FAILED", e)
PY
