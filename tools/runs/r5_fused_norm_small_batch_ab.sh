cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=short -k "whole_row_epilogue or small_batches_split_k" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12 ) > $O/t.txt
cat $O/t.txt
for rep in 1 2 3; do
  for v in new old; do
    L=$PWD/naturalspeech2_pytorch_amd/libns2hip.so; [ $v = old ] && L=$PWD/tools/ab/libns2hip_nofusednorm.so
    NS2_LIB=$L timeout 300 python bench.py --dim 128 --depth 6 --batch 1 --steps 200 --warmup 10 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/d128_b1_${v}_$rep.json 2> /dev/null
    NS2_LIB=$L timeout 300 python bench.py --batch 1 --steps 100 --warmup 10 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/d512_b1_${v}_$rep.json 2> /dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["ms_per_step"], j["value"])
    except Exception as e: print(f, "ERR", e)
PY
