R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-250
timeout 900 python -m pytest tests/test_parity_r2_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "lstm or seanet or encodec or golden or encoders" 2>&1 | tail -4 | cut -c1-250
python tools/run_codec.py --batch 32 --precision exact --decode 2>&1 | grep -v amdgpu | tail -1
python tools/run_codec.py --batch 8 --precision exact --decode 2>&1 | grep -v amdgpu | tail -1
