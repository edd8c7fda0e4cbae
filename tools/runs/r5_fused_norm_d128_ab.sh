#!/bin/bash
# Round 5: residual update + RMSNorm in ONE launch where the GEMM's workgroups own whole rows (dim = 128: gemm.hip EPI_F32 with nrm_*),
# against the separate rmsnorm_kernel (build variant tools/ab/libns2hip_nofusednorm.so): model tests at dim 128 / 64 first, then
# bench.py --dim 128 --depth 6 (config 2) and its batch-1 form, alternating on one box.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5m; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py -q -m gpu --tb=short -x -k "not seanet and not codec and not encodec and not trajectory_1000" 2>&1 | tail -n 6 ) > $O/t_model.txt
for rep in 1 2 3; do
  for v in new old; do
    L=""; [ $v = old ] && L=$PWD/tools/ab/libns2hip_nofusednorm.so
    NS2_LIB=${L:-$PWD/naturalspeech2_pytorch_amd/libns2hip.so} timeout 300 python bench.py --dim 128 --depth 6 --steps 50 --warmup 5 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/d128_b32_${v}_$rep.json 2> /dev/null
    NS2_LIB=${L:-$PWD/naturalspeech2_pytorch_amd/libns2hip.so} timeout 300 python bench.py --dim 128 --depth 6 --batch 1 --steps 100 --warmup 10 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/d128_b1_${v}_$rep.json 2> /dev/null
  done
done
tail -n 4 $O/t_model.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["ms_per_step"], j["value"])
    except Exception as e: print(f, "ERR", e)
PY
