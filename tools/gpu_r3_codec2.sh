R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "lstm or seanet or encodec" 2>&1 | tail -8 | cut -c1-250
python tools/run_codec.py --batch 32 --precision exact --decode 2>&1 | grep -v amdgpu | tail -1
python tools/run_codec.py --batch 8 --precision exact --decode 2>&1 | grep -v amdgpu | tail -1
python tools/bench_attention.py 2>/dev/null | tail -1
