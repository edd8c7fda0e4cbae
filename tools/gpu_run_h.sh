#!/bin/bash
# Round-2 GPU pass H: where the conditioned model's error enters; RVQ 16-wave kernel A/B against the 8-wave build; graph
# replay and inter-kernel gaps of the hybrid step; the codec side workload.  Outputs under gpurun_out/r2h/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2h
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python tools/cond_error_trace.py ) > $OUT/cond_trace.log 2>&1
for i in 1 2; do
  python tools/bench_rvq.py >> $OUT/rvq_ab.log 2>&1
  NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_old.so python tools/bench_rvq.py >> $OUT/rvq_ab.log 2>&1
done
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "rvq or encodec or codec" 2>&1 | tail -8 ) > $OUT/pytest_rvq.log 2>&1
python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $OUT/bench_nograph.json 2> $OUT/bench_nograph.err
python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --graph > $OUT/bench_graph.json 2> $OUT/bench_graph.err
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/bench_side.json 2> $OUT/bench_side.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $OUT/trace.log 2>&1
python $R/tools/gap_analysis.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) --last 440 > $OUT/gaps.txt 2>&1
rm -rf $OUT/trace
cd $R
cat $OUT/cond_trace.log | cut -c1-400; cat $OUT/rvq_ab.log; tail -3 $OUT/pytest_rvq.log; cut -c1-200 $OUT/bench_nograph.json; cut -c1-200 $OUT/bench_graph.json; cat $OUT/gaps.txt
python -c "import json; d=json.load(open('$OUT/bench_side.json')); print(json.dumps(d['side'].get('codec_seanet_rvq'))[:1500])"
