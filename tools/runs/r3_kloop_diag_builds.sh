R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for L in blk diag1 diag2 diag3; do
echo "== $L"
export NS2_LIB=$R/naturalspeech2_pytorch_amd/libns2hip_g2_$L.so
timeout 300 python tools/trace_blocks.py --prec 4 --which qkv,ffout 2>&1 | grep -v amdgpu
timeout 200 python tools/trace_blocks.py --prec 2 --which qkv,ffout 2>&1 | grep -v amdgpu
done
