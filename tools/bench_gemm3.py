"""Round 6: the lean mixed linear kernel (csrc/gemm3_kernel.h) against gemm2_kernel<2, *> at the headline shapes, alternating in one process.
    python tools/bench_gemm3.py [--iters 20] [--rounds 5]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from naturalspeech2_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N, d, f = 32, 1024, 512, 1365
M = B * N
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
x512 = ops.split(rnd(M, d), precision=4)
xf = ops.split(rnd(M, f), precision=4)
wq = ops.PackedWeight(rnd(1536, d, scale=0.04), precision=4).tile_linear()
w1 = ops.PackedWeight(rnd(2 * f, d, scale=0.04), geglu=True, precision=4).tile_linear(); pb = ops.geglu_pack_bias(rnd(2 * f), f)
w2 = ops.PackedWeight(rnd(d, f, scale=0.03), precision=4).tile_linear(); b2 = rnd(d); r = rnd(M, d)
wo = ops.PackedWeight(rnd(d, d, scale=0.04), precision=4).tile_linear()
cases = {
    "qkv [M,512]x[512,1536]": (lambda: ops.linear_qkv(wq, x512, seq_len=N, split_col=1024, precision=4), 2.0 * M * d * 1536),
    "ffin+geglu [M,512]x[512,2730]": (lambda: ops.linear_geglu(w1, x512, pb, precision=4), 2.0 * M * d * 2 * f),
    "ffout+res [M,1365]x[1365,512]": (lambda: ops.linear_f32(w2, xf, bias=b2, resid=r, precision=4), 2.0 * M * f * d),
    "outproj+res [M,512]x[512,512]": (lambda: ops.linear_f32(wo, x512, resid=r, precision=4), 2.0 * M * d * d),
}
force = lambda k: _lib.check(_lib.load().ns2_debug_force_gemm(k))
KS = (5, 2, 1)
res = {(n, k): [] for n in cases for k in KS}
for rd in range(args.rounds):
    for n, (fn, fl) in cases.items():
        for k in KS:
            force(k)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[(n, k)].append(e0.elapsed_time(e1) / args.iters)
force(0)
for n, (fn, fl) in cases.items():
    t = {k: sorted(res[(n, k)])[len(res[(n, k)]) // 2] for k in KS}
    print(f"{n:34s} gemm3 {t[5]*1e3:7.1f} us ({fl/t[5]/1e9:6.1f} TF)   gemm2 {t[2]*1e3:7.1f} us ({fl/t[2]/1e9:6.1f} TF)   ratio {t[2]/t[5]:.3f}   128x128 kernel {t[1]*1e3:7.1f} us")
