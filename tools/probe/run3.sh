mkdir -p gpurun_out/r6b
P=tools/probe/ffconv_probe
V="--variants 0,1,7,9 --iters 20 --rounds 5"
( timeout 120 $P $V --order 0 --verify 1; timeout 120 $P $V --order 1 --verify 1; timeout 120 $P $V --order 2 --verify 1; timeout 120 $P $V --order 0 --hot 1 --verify 0; timeout 120 $P $V --order 0 --verify 0 ) > gpurun_out/r6b/probe3.txt 2>&1
grep -v "rerun\|digest" gpurun_out/r6b/probe3.txt
python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old kernel in bench.py:', d['value'], d['ms_per_step'], d['roofline'])"
