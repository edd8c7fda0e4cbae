"""Experiment: does the row stride of the K = 512 operands matter (L2 channel aliasing of 2 KiB-strided rows)?
Times the QKV / out-proj shaped GEMMs with padded A / W row strides."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops, _lib
dev = torch.device("cuda:0")
_lib.check(_lib.load().ns2_debug_force_gemm(2))
M, d, N = 32768, 512, 1024
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
X = rnd(M, d)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for P in (4, 2, 3):
    for lda, kw in ((512, 512), (544, 512), (576, 512), (544, 544), (576, 576), (640, 640)):
        x = ops.split(X, ldo=lda, precision=P)
        Wt = torch.zeros(1536, kw, device=dev); Wt[:, :d] = rnd(1536, d, scale=0.04)
        wq = ops.PackedWeight(Wt, precision=P)
        if kw > lda: continue
        if kw != lda and kw != 512: continue
        xa = x
        if kw > 512 and lda == kw: pass
        us = timeit(lambda: ops.linear_qkv(wq, xa, seq_len=N, split_col=1024, precision=P))
        Wo = torch.zeros(512, kw, device=dev); Wo[:, :d] = rnd(512, d, scale=0.04)
        wo = ops.PackedWeight(Wo, precision=P); ro = rnd(M, d)
        us2 = timeit(lambda: ops.linear_f32(wo, xa, resid=ro, precision=P))
        print(f"prec {P} lda {lda} (A row {lda * (2 if P == 2 else 4)} B) wK {kw}: qkv {us:7.1f} us   outproj {us2:7.1f} us", flush=True)
