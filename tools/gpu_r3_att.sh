cd $GRAFT_REPO_ROOT
for NW in 4 8; do echo "default NW=$NW"; NS2_ATTN_NW=$NW python tools/bench_attention.py 2>/dev/null | tail -1; done
for NW in 4 8; do echo "at3 NW=$NW"; NS2_LIB=$GRAFT_REPO_ROOT/naturalspeech2_pytorch_amd/libns2hip_at3.so NS2_ATTN_NW=$NW python tools/bench_attention.py 2>/dev/null | tail -1; done
