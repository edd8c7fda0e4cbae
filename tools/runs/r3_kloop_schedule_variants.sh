R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for L in blk var1 var2 var3 blk; do
echo "== $L"
export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_g2_$L.so
timeout 300 python tools/trace_blocks.py --prec 4 --which qkv,ffout 2>&1 | grep -v amdgpu | cut -c1-175
timeout 200 python tools/trace_blocks.py --prec 2 --which qkv,ffout 2>&1 | grep -v amdgpu | cut -c1-175
done
