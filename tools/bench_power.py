"""Is the FF-conv GEMM power/clock bound?  Same launch on random vs all-zero operands (identical instruction stream and
cycle count; fewer toggling bits -> less power -> higher sustained clock)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naturalspeech2_pytorch_amd import _lib, ops
_lib.check(_lib.load().ns2_debug_force_gemm(2))
M, f, N = 32768, 1365, 1024
g = torch.Generator().manual_seed(0)
def run(x, w, b, prec, tag):
    a = ops.split(x, ldo=ops.round_up(f, 32), precision=prec); pw = ops.PackedWeight(w, precision=prec)
    fn = lambda: ops.linear_split(pw, a, bias=b, conv_taps=3, dilation=1, seq_len=N, precision=prec)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"{tag:28s} prec={prec}: {us:8.1f} us   {2.0*M*f*3*f/us/1e6:7.1f} TFLOP/s algorithmic")
for prec in (4, 2, 3):
    run(torch.randn(M, f, generator=g).cuda(), (torch.randn(f, f, 3, generator=g) * 0.02).cuda(), torch.randn(f, generator=g).cuda(), prec, "random operands")
    run(torch.zeros(M, f).cuda(), torch.zeros(f, f, 3).cuda(), torch.zeros(f).cuda(), prec, "all-zero operands")
    run(torch.randn(M, f, generator=g).cuda(), torch.zeros(f, f, 3).cuda(), torch.zeros(f).cuda(), prec, "random A, zero W")
