#!/bin/bash
# per-dispatch timeline of one encode and one decode (kernel trace), plus the stats table
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3codec4; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/tools/run_codec.py --batch 32 --precision exact --iters 1 --decode > $OUT/prof.log 2>&1
tail -1 $OUT/prof.log
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/codec_kernel_stats.csv
cp $(ls $OUT/prof/*/*kernel_trace.csv | head -1) $OUT/trace.csv
rm -rf $OUT/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# keep the second half (timed encode + timed decode): find the last two rvq_encode/rvq_decode markers
names=[r["Kernel_Name"] for r in rows]
idx_enc=[i for i,n in enumerate(names) if "rvq_encode_kernel" in n]
start=idx_enc[-2]+1 if len(idx_enc)>=2 else 0
# after warm-up encode comes warm-up decode, then timed encode, timed decode: print from the dispatch after the 1st decode's end
idx_dec=[i for i,n in enumerate(names) if "rvq_decode_kernel" in n]
t0=None
out=[]
for r in rows[start:]:
    n=r["Kernel_Name"]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    short=n.split("(")[0].replace("void ns2::","").replace("ns2::","")[:60]
    g=r.get("Grid_Size_X") or r.get("Grid_Size") or ""
    out.append(f"{short:60s} grid {g:>10s} {d:10.1f} us")
open("$OUT/timeline.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:150]))
PY
