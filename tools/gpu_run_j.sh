#!/bin/bash
# Round-2 GPU pass J: attention softmax fast path (A/B against the previous build, kernel + model tests), codec kernel profile.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2j
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
for i in 1 2; do
  python tools/bench_attention.py >> $OUT/attn_ab.log 2>&1
  NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_old.so python tools/bench_attention.py >> $OUT/attn_ab.log 2>&1
done
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "attention or attn" 2>&1 | tail -8 ) > $OUT/pytest_attn.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --ignore=tests/test_parity_r2_gpu.py --ignore=tests/test_kernels_gpu.py 2>&1 | tail -8 ) > $OUT/pytest_model.log 2>&1
python bench.py --steps 10 --warmup 3 --no-side --no-secondary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_codec -- python $R/tools/run_codec.py > $OUT/prof_codec.log 2>&1
cp $(ls $OUT/prof_codec/*/*kernel_stats.csv | head -1) $OUT/codec_kernel_stats.csv
rm -rf $OUT/prof_codec
cd $R
grep -v amdgpu $OUT/attn_ab.log; tail -3 $OUT/pytest_attn.log; tail -3 $OUT/pytest_model.log; cut -c1-330 $OUT/bench.json; echo; head -14 $OUT/codec_kernel_stats.csv | cut -c1-150
