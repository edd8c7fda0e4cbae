#!/bin/bash
# Round-2 GPU pass K: persistent LSTM recurrence (tests under a hard timeout, then codec timing with both paths).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2k
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time timeout 300 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 120 -p no:cacheprovider -k "lstm or seanet" 2>&1 | tail -15 ) > $OUT/pytest_lstm.log 2>&1
tail -6 $OUT/pytest_lstm.log
if grep -q "failed\|Timeout\|rror" $OUT/pytest_lstm.log; then echo "LSTM TESTS FAILED: skipping the rest"; exit 0; fi
( time timeout 600 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "encodec or codec or composition" 2>&1 | tail -8 ) > $OUT/pytest_codec.log 2>&1
tail -4 $OUT/pytest_codec.log
timeout 600 python bench.py --steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-parity > $OUT/bench_side.json 2> $OUT/bench_side.err
python -c "import json; d=json.load(open('$OUT/bench_side.json')); print(json.dumps(d['side'].get('codec_seanet_rvq'))[:900])"
NS2_LSTM_PERSISTENT=0 timeout 600 python bench.py --steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-parity > $OUT/bench_side_step.json 2> $OUT/bench_side_step.err
python -c "import json; d=json.load(open('$OUT/bench_side_step.json')); print(json.dumps(d['side'].get('codec_seanet_rvq'))[:900])"
